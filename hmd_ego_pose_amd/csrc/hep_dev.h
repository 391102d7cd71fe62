// hep_dev.h - device-side helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  // ONE v_cvt_pk_bf16_f32 (RNE, NaN-preserving).  Two scalar casts + shift + or compiled to four instructions per pair -
  // on every bf16 store of every kernel.
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float swishf(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float apply_act(float x, int act) {
  return act == 1 ? swishf(x) : (act == 2 ? sigmoidf(x) : x);
}
// bf16 kernels use v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions): the
// result is rounded to bf16 (8 bits) right after.
// fp32 sessions: v_rcp_f32 + one Newton step (<= 1 ulp of 1 / d, 4 instructions) in place of the division sequence (round 4:
// 25.7k -> 26.5k frames/s; ADD against the reference unchanged at 0.003 mm).  The reference's own x * sigmoid(x) rounds twice
// (efficientnet/utils.py:57-59), so neither form is bit-identical to it; both sit ~1 ulp from the exact quotient.
__device__ __forceinline__ float rcp_newton(float d) {
  d = d > 0x1p126f ? 0x1p126f : d;              // e^-x overflowed: keep the step finite (the product is ~1e-37 either way); a NaN stays a NaN (fminf would return the bound)
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
template <bool FAST> __device__ __forceinline__ float swish_t(float x) {
  const float d = 1.0f + __expf(-x);
  return x * (FAST ? __builtin_amdgcn_rcpf(d) : rcp_newton(d));
}
template <bool FAST> __device__ __forceinline__ float sigmoid_t(float x) {
  const float d = 1.0f + __expf(-x);
  return FAST ? __builtin_amdgcn_rcpf(d) : rcp_newton(d);
}
// Two activations at a time with the multiplies and the add as packed instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two
// lanes of fp32 per issue slot) - bit-identical to the scalar forms above (same IEEE operations on the same operands), one issue
// slot fewer per value in bf16 sessions (7 -> 5.5 of 4 cycles; the two transcendentals stay 8 each), 2.5 fewer in fp32.  The chip's
// VALU pipes are >= 50 % busy with four batches in flight (profiles/r04/d_valu_mix.json), swish is ~a quarter of that.
template <bool FAST> __device__ __forceinline__ f32x2_t rcp2_t(f32x2_t d) {
  if (FAST) return (f32x2_t){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  d = (f32x2_t){d[0] > 0x1p126f ? 0x1p126f : d[0], d[1] > 0x1p126f ? 0x1p126f : d[1]};      // NaN-preserving clamp (see rcp_newton)
  const f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2_t e = __builtin_elementwise_fma(-d, r, (f32x2_t){1.0f, 1.0f});
  return __builtin_elementwise_fma(e, r, r);
}
template <bool FAST> __device__ __forceinline__ f32x2_t sigmoid_den2(f32x2_t x) {      // 1 + e^-x
  const f32x2_t t = x * (f32x2_t){-1.44269504088896340736f, -1.44269504088896340736f};
  return (f32x2_t){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + (f32x2_t){1.0f, 1.0f};
}
template <bool FAST, int N> __device__ __forceinline__ void swish_n(float* v) {
  static_assert(N % 2 == 0, "pairs");
#pragma unroll
  for (int c = 0; c < N; c += 2) {
    const f32x2_t x = {v[c], v[c + 1]};
    const f32x2_t y = x * rcp2_t<FAST>(sigmoid_den2<FAST>(x));
    v[c] = y[0]; v[c + 1] = y[1];
  }
}
template <bool FAST, int N> __device__ __forceinline__ void sigmoid_n(float* v) {
  static_assert(N % 2 == 0, "pairs");
#pragma unroll
  for (int c = 0; c < N; c += 2) {
    const f32x2_t y = rcp2_t<FAST>(sigmoid_den2<FAST>((f32x2_t){v[c], v[c + 1]}));
    v[c] = y[0]; v[c + 1] = y[1];
  }
}
template <bool FAST> __device__ __forceinline__ float apply_act_t(float x, int act) {
  return act == 1 ? swish_t<FAST>(x) : (act == 2 ? sigmoid_t<FAST>(x) : x);
}

// 8 bf16 activations (optionally times 8 squeeze-excite scales) -> 8 e4m3 bytes: x * inv_scale, saturated to +-448, RNE
__device__ __forceinline__ long cvt_fp8x8(u32x4 raw, const float* s8, float inv_scale) {
  float f[8];
#pragma unroll
  for (int q = 0; q < 4; q++) { f[2 * q] = __uint_as_float(raw[q] << 16); f[2 * q + 1] = __uint_as_float(raw[q] & 0xffff0000u); }
  if (s8) {
#pragma unroll
    for (int q = 0; q < 8; q++) f[q] *= s8[q];
  }
#pragma unroll
  for (int q = 0; q < 8; q++) f[q] = fminf(fmaxf(f[q] * inv_scale, -448.f), 448.f);
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

// 8 consecutive channels <-> 8 floats
template <bool BF16> struct Vec8;
template <> struct Vec8<true> {
  typedef bf16_t elem;
  static __device__ __forceinline__ void load(const void* base, int64_t idx, float v[8]) {
    u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(base) + idx);
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(void* base, int64_t idx, const float v[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(base) + idx) = r;
  }
  static __device__ __forceinline__ void store4(void* base, int64_t idx, const float v[4]) {
    u32x2 r; r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(base) + idx) = r;
  }
  static __device__ __forceinline__ void load4(const void* base, int64_t idx, float v[4]) {
    u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx);
    v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xffff0000u);
    v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xffff0000u);
  }
};
template <> struct Vec8<false> {
  typedef float elem;
  static __device__ __forceinline__ void load(const void* base, int64_t idx, float v[8]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
    f32x4 a = p[0], b = p[1];
#pragma unroll
    for (int i = 0; i < 4; i++) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(void* base, int64_t idx, const float v[8]) {
    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx);
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = v[i]; b[i] = v[4 + i]; }
    p[0] = a; p[1] = b;
  }
  static __device__ __forceinline__ void store4(void* base, int64_t idx, const float v[4]) {
    f32x4 a; a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = a;
  }
  static __device__ __forceinline__ void load4(const void* base, int64_t idx, float v[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  }
};

// XCD-aware block remap (guide T1): hardware deals consecutive block ids round-robin over the
// 8 XCDs; give each XCD a contiguous range of logical ids so neighbours share an L2.  Bijective
// for any grid size.
// two-dimensional grids: the hardware deals the LINEAR block id (x + y * gx) round-robin, so the remap works on that
__device__ __forceinline__ void xcd_remap2(int bx, int by, int gx, int gy, int* x, int* y);
// x / d with rcp = rcp_u32(d) (hep_internal.h): exact for x * d < 2^32
__device__ __forceinline__ int udiv_rcp(int x, uint32_t rcp) { const int q = (int)__umulhi((uint32_t)x, rcp); return rcp == 0u ? x : q; }   // (a select, not a branch)
// x / d for 0 <= x < 2^24 with inv ~ 1 / d (v_rcp_f32): float estimate, one correction step either way
__device__ __forceinline__ int udiv_f(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  const int rem = x - q * d;
  q += rem >= d ? 1 : 0; q -= rem < 0 ? 1 : 0;
  return q;
}
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
__device__ __forceinline__ void xcd_remap2(int bx, int by, int gx, int gy, int* x, int* y) {
  const int logical = xcd_remap(bx + by * gx, gx * gy);
  *y = logical / gx; *x = logical - *y * gx;
}

// Sanitizer build (make poison -> libhep_poison.so, selected with HEP_LIB): the LDS-heavy kernels fill their dynamic LDS with
// NaN words (0x7FC07FC0: a NaN as fp32 and as two bf16) before their first use.  A value read from a cell the kernel never
// wrote - a pad column, a row behind the staged ones - then reaches the outputs as NaN and fails the parity tests instead of
// depending on what an earlier launch left behind (GPU AddressSanitizer is not available on this pool; this caught the
// 0 x NaN of k_mbf.hip's k-step tail at phi 0 @ 128).
#ifdef HEP_POISON_LDS
__device__ __forceinline__ void hep_poison_lds(void* smem, size_t bytes) {
  unsigned* p = reinterpret_cast<unsigned*>(smem);
  for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) p[i] = 0x7FC07FC0u;
  __syncthreads();
}
#define HEP_POISON(smem, bytes) hep_poison_lds(smem, bytes)
#else
#define HEP_POISON(smem, bytes)

#endif

// One dword of every 64-byte line of a by-value kernel argument struct (<= 512 bytes) is requested at once and waited for: the
// compiler fetches argument fields next to their first use, in several dependent groups, and each group that starts a new line is a
// scalar-cache miss (~0.2 us) in the prologue of a latency-bound workgroup.  Behind this the later loads hit the scalar cache.
// (Inline asm: a plain C++ read of the same memory is folded into the field's own late load.)
template <int BYTES> __device__ __forceinline__ void kernarg_warm() {
  static_assert(BYTES >= 4 && BYTES <= 512, "eight lines");
  constexpr int L = BYTES - 4;
  typedef const __attribute__((address_space(4))) uint32_t* cptr;
  cptr kp = (cptr)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile("s_load_dword %0, %8, %9\n\ts_load_dword %1, %8, %10\n\ts_load_dword %2, %8, %11\n\ts_load_dword %3, %8, %12\n\t"
               "s_load_dword %4, %8, %13\n\ts_load_dword %5, %8, %14\n\ts_load_dword %6, %8, %15\n\ts_load_dword %7, %8, %16\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
               : "s"(kp), "i"(0), "i"(64 < L ? 64 : L), "i"(128 < L ? 128 : L), "i"(192 < L ? 192 : L), "i"(256 < L ? 256 : L), "i"(320 < L ? 320 : L), "i"(384 < L ? 384 : L), "i"(448 < L ? 448 : L)
               : "memory");
}
