// k_late.hip - an image-resident run of late MBConv blocks as ONE gfx950 launch (bf16 sessions).
//
//   per block:  expand 1x1 (+BN0, swish) -> depthwise k x k, TF-SAME (+BN1, swish) -> squeeze-excite (mean -> FC -> swish -> FC ->
//               sigmoid) -> project 1x1 (+BN2) (+ input)                                     (reference efficientnet/model.py:69-104)
//
// On the 8x8 maps one image is 64 pixels, so a whole block is local to a workgroup: ONE workgroup (1024 threads, one CU) per image
// runs the blocks back to back with `__syncthreads` seams only.  Launch-by-launch the same blocks are a front launch
// (k_mbf.hip: 18 workgroups per image that each stage the same input tile), a squeeze-excite launch and a project GEMM per block;
// with four batches in flight a launch costs the pipeline ~2.5 us + ~0.13 x its stand-alone duration whatever it computes (measured:
// tools/exp/late_pricing.sh, profiles/r05), so 12 launches of ~8 us that each fill a fraction of the chip become one launch of
// 16 workgroups.  What bounds the workgroup is its CU's vector ALU (depthwise taps + two swish per expanded element), not bytes:
// one CU pulls a fragment-ordered weight stream at ~90 GB/s (tools/wstream) and the five blocks need ~50.
//
// Layout of the work inside the workgroup (expanded channels in chunks of LATE_CC = 128):
//   waves 8-15 ("mm"): chunk c+1: expand conv on MFMA - the wave owns one n-tile (16 expanded channels) for all four m-tiles
//              (64 pixels); its weight fragments arrive straight from global memory in MFMA operand order (host-packed: one wave
//              instruction = 1 KB contiguous), the block input sits in LDS; bias + swish -> bf16 -> the halo'd tile E[(c+1) & 1];
//              plus: the depthwise weights of chunk c+1 -> LDS, and the squeeze-excite reduce FC of chunk c-1 (linear in the channel
//              sums: accumulated chunk by chunk, never a 221 KB matrix at the end)
//   waves 0-7  ("dw"): chunk c: depthwise taps from E[c & 1]: lane = 8 channels x 4 adjacent pixels of a row x one half of the tap
//              rows (lanes 0-31: rows 0 .. KH-1 + bias, lanes 32-63: the rest; exchanged with v_permlane32_swap and added first +
//              second - the summation order of k_mbf.hip's two half-workgroups, so the depthwise output is bit-identical to it);
//              bias + swish -> bf16 -> global scratch (L2) in [channel group][pixel] order, channel sums -> LDS
//   one barrier per chunk.  Then every wave: hidden vector, expand FC + sigmoid -> scale[Cexp] (LDS); the depthwise outputs come
//   back from L2, are multiplied by the scale, rounded to bf16 (the A operand of k_pw_impl.h's project GEMM) and parked as
//   As[64][Cexp] - the expanded tiles are dead by then and all of LDS is free; project conv on MFMA: wave = (n-tile group, K half),
//   weight fragments streamed once, activation fragments from LDS feeding 2-3 n-tiles each; the K halves meet in LDS (fixed
//   order); + bias (+ residual from global memory) -> bf16 -> global block output AND the next block's input tile in LDS.
//
// Rounding points are those of the launch-by-launch plan (expanded tile bf16, depthwise output bf16 with the channel sums taken
// from the fp32 values, A x scale rounded to bf16, fp32 accumulation); only fp32 summation orders differ (squeeze-excite sums,
// the split of K in the project conv).  Gate: the teacher-forced bf16 stage tests (tests/test_gpu_parity.py).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "hep_dev.h"
#include "hep_internal.h"

namespace {
constexpr int LATE_EP = LATE_CC + 8;            // pitch of an expanded-tile position in bf16 elements (272 B)
constexpr int LATE_NDW = 8;                     // waves 0..7 depthwise, 8..15 expand / staging / squeeze-excite
constexpr int LATE_MM_LANES = LATE_THREADS - LATE_NDW * 64;
}

// Optional per-wave timeline (make trace: -DHEP_LATE_TRACE): s_memrealtime (100 MHz) stamps, 64 slots per (workgroup, wave, block),
// read back with hep_dbg_late_trace() (tools/trace_late.py) - profiling builds only.
//   0 block start, 1 first chunk barrier, 2 chunk loop done, 3 scale ready, 4 As parked, 5 block done, 6 K loop done, 7 partials parked
//   8 + c: barrier behind chunk c;  32..: inside chunk 1 - mm: 32 top, 33 expand done, 34 squeeze-excite partial done, 35 weights parked;
//   dw: 40 top, 41 taps done, 42 outputs stored, 43 channel sums done
#ifdef HEP_LATE_TRACE
__device__ unsigned long long* g_late_trace = nullptr;
#define LSTAMP(i) do { if (g_late_trace && lane == 0) g_late_trace[(((size_t)blockIdx.x * 16 + wave) * LATE_MAX_BLOCKS + bi) * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LSTAMP(i)
#endif

__device__ __forceinline__ u32x4 late_ldg(const unsigned char* p) { return *reinterpret_cast<const u32x4*>(p); }
// Bytes handed from one workgroup of an image's group to the others (depthwise outputs, squeeze-excite partial sums, block outputs
// read back as residuals): every store write-through (sc0 sc1), every load past the non-coherent caches (sc0 sc1) - the hand-off then
// needs no release / acquire fence (MI355X_MICROARCH.md, "Valid forms"): the storing waves drain with vmcnt(0), one lane adds to the
// group's counter, the others poll it.  Raw buffer instructions: the compiler counts them in its own s_waitcnt bookkeeping.
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned late_v4;
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned late_v2;
#define LATE_COHERENT 17      /* aux bits of the raw buffer builtins on gfx942 / gfx950: sc0 | sc1 */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t late_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); }

// zero the halo positions of both expanded-tile buffers (the interior is rewritten by every chunk, the halo never)
template <int KS>
__device__ __forceinline__ void late_zero_halo(unsigned char* smem, const LateArgs& a) {
  constexpr int PW = KS + 7, PAD = (KS - 1) / 2, VPP = LATE_EP * 2 / 16, PER = PW * PW * VPP;
  for (int u = threadIdx.x; u < 2 * PER; u += LATE_NDW * 64) {      // (called by the depthwise waves: threads 0 .. 511)
    const int buf = u >= PER ? 1 : 0, rem = u - buf * PER, p = rem / VPP, v = rem - p * VPP;
    const int ty = p / PW, tx = p - ty * PW;
    if (ty < PAD || ty >= PAD + 8 || tx < PAD || tx >= PAD + 8)
      *reinterpret_cast<u32x4*>(smem + a.off_e + buf * a.e_stride + p * (LATE_EP * 2) + v * 16) = (u32x4){0u, 0u, 0u, 0u};
  }
}

// ---- project conv of one block: As[64][Cexp] (LDS, scaled) x Wp -> + bias (+ residual) -> global + next input tile ----
template <int NTW>
__device__ __forceinline__ void late_project(const LateArgs& a, const LateBlock& L, unsigned char* smem, int b, bool last_block, int bi, int gw) {
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));       // (opaque per call: lane-derived addresses are not hoisted to kernel entry and spilled there)
  const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), r = lane & 15, g = lane >> 4;
  const int NG = L.ng, KSP = L.Cexp >> 5, KSH = KSP >> 1, AP = L.Cexp + 8;
  const bool active = wave < 2 * NG;
  const bool coh = a.G > 1;                                  // the block outputs cross workgroups (written by workgroup 0 of the group, read as residuals by all)
  const bool timed_out = coh && reinterpret_cast<const int*>(smem + a.off_hid)[63] != 0;      // a meeting of the group timed out (late_block): NaN out
  const int kq = wave >= NG ? 1 : 0, ng = wave - kq * NG;
  const bf16_t* As = reinterpret_cast<const bf16_t*>(smem);
  f32x4 acc[NTW][4];
#pragma unroll
  for (int j = 0; j < NTW; j++)
#pragma unroll
    for (int mt = 0; mt < 4; mt++) acc[j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // residual and bias of the lanes that finish the tile (K half 0): requested before the K loop where the registers allow it
  // (two n-tiles per wave), behind it otherwise - a round trip to L2 in front of the partial-sum barrier cost 2 us per block
  u32x2 resv[NTW][4]; f32x4 biasv[NTW];
  auto load_tail = [&]() {
    if (active && kq == 0) {
#pragma unroll
      for (int j = 0; j < NTW; j++) {
        const int n = min((ng * NTW + j) * 16 + 4 * g, L.N - 4);
        biasv[j] = *reinterpret_cast<const f32x4*>(a.blob + L.off_bp + (size_t)((ng * NTW + j) * 16 + 4 * g) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
          if (!L.skip) resv[j][mt] = (u32x2){0u, 0u};
          else if (!coh) resv[j][mt] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(L.res) + ((size_t)b * 64 + mt * 16 + r) * L.N + n);
          else {     // (coherent loads are 16 bytes wide: the aligned pair of this lane's four channels, the lane keeps its half)
            const int n16 = min((ng * NTW + j) * 16 + 4 * (g & ~1), L.N - 8);
            const u32x4 rv = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(late_rsrc(L.res), (int)((((size_t)b * 64 + mt * 16 + r) * L.N + n16) * 2), 0, LATE_COHERENT));
            resv[j][mt] = (g & 1) ? (u32x2){rv[2], rv[3]} : (u32x2){rv[0], rv[1]};
          }
      }
    }
  };
  if constexpr (NTW <= 2) load_tail();
  if (active) {
    constexpr int R = 3;                                     // k-steps of weight fragments in flight
    const unsigned char* wp = a.blob + L.off_wp + ((size_t)(ng * KSP + kq * KSH) * NTW * 64 + lane) * 16;
    u32x4 wf[R][NTW];
#pragma unroll
    for (int s = 0; s < R; s++)
#pragma unroll
      for (int j = 0; j < NTW; j++) wf[s][j] = late_ldg(wp + (size_t)(min(s, KSH - 1) * NTW + j) * 1024);
    const bf16_t* arow = As + r * AP + kq * KSH * 32 + 8 * g;
    for (int ks = 0; ks < KSH; ks += R) {
#pragma unroll
      for (int s = 0; s < R; s++) {
        if (ks + s < KSH) {                                  // (wave-uniform)
          u32x4 x[4];
#pragma unroll
          for (int mt = 0; mt < 4; mt++) x[mt] = *reinterpret_cast<const u32x4*>(arow + mt * 16 * AP + (ks + s) * 32);
#pragma unroll
          for (int j = 0; j < NTW; j++)
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
              acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s][j]), __builtin_bit_cast(bf16x8, x[mt]), acc[j][mt], 0, 0, 0);
          const int kn = min(ks + s + R, KSH - 1);           // (past the slice: the last step again, never used)
#pragma unroll
          for (int j = 0; j < NTW; j++) wf[s][j] = late_ldg(wp + (size_t)(kn * NTW + j) * 1024);
        }
      }
    }
  }
  LSTAMP(6);
  if constexpr (NTW > 2) load_tail();
  __syncthreads();                                           // As is dead: its space takes the K-half partial sums
  f32x4* part = reinterpret_cast<f32x4*>(smem);
  if (active && kq == 1) {
#pragma unroll
    for (int j = 0; j < NTW; j++)
#pragma unroll
      for (int mt = 0; mt < 4; mt++) part[((ng * NTW + j) * 4 + mt) * 64 + lane] = acc[j][mt];
  }
  __syncthreads();
  LSTAMP(7);
  if (active && kq == 0) {
    bf16_t* Xn = reinterpret_cast<bf16_t*>(smem + a.off_x);
    const int XPn = L.N + 8;
#pragma unroll
    for (int j = 0; j < NTW; j++) {
      const int n = (ng * NTW + j) * 16 + 4 * g;
      if (n < L.N) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
          const f32x4 s = acc[j][mt] + part[((ng * NTW + j) * 4 + mt) * 64 + lane];
          const int m = mt * 16 + r;
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; q++) v[q] = timed_out ? __builtin_nanf("") : s[q] + biasv[j][q];
          if (L.skip) {
            const u32x2 rv = resv[j][mt];
            v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
            v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
          }
          if (!coh) Vec8<true>::store4(L.out, ((int64_t)b * 64 + m) * L.N + n, v);
          if (!last_block || coh) Vec8<true>::store4(Xn, (int64_t)m * XPn + n, v);
        }
      }
    }
  }
  __syncthreads();                                           // the next block's input tile is complete, the partial sums are consumed
  if (coh && gw == 0) {
    // groups: the block output leaves through the finished LDS tile - consecutive lanes store consecutive 16-byte vectors of the
    // row-major [64][N] tensor, so every 128-byte line is written whole by one store instruction (the hand-off rule; the other
    // members read it back as the next block's residual)
    const bf16_t* Xn = reinterpret_cast<const bf16_t*>(smem + a.off_x);
    const int XPn = L.N + 8, vpr = L.N >> 3;
    const float vpr_inv = __builtin_amdgcn_rcpf((float)vpr);
    for (int u = tid_; u < 64 * vpr; u += LATE_THREADS) {
      const int m = (int)(((float)u + 0.5f) * vpr_inv), v = u - m * vpr;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(late_v4, *reinterpret_cast<const u32x4*>(Xn + m * XPn + v * 8)), late_rsrc(L.out), (int)(((size_t)b * 64 * L.N) * 2) + u * 16, 0, LATE_COHERENT);
    }
  }
}

// ---- one block ----
template <int KS, int KSE>
__device__ __forceinline__ void late_block(const LateArgs& a, const LateBlock& L, unsigned char* smem, int b, bool last_block, int bi, int gw) {
  constexpr int PW = KS + 7, PAD = (KS - 1) / 2, KK = KS * KS, KH = (KS + 1) / 2, NXP = KS + 3;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));        // (opaque per block: the compiler hoisted the lane-derived addresses of every instantiation to kernel entry and spilled them)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
  const bool mm = wave >= LATE_NDW;
  const int ml = tid - LATE_NDW * 64, t = wave - LATE_NDW;      // mm lane index 0..511, n-tile of the chunk
  const int XP = L.Cin + 8, NC = L.nchunks, sqp = L.sqp;
  const int G = a.G, c_lo = gw * NC / G, c_hi = (gw + 1) * NC / G;      // this workgroup's chunks of the expanded channels
  const bool coh = G > 1;
  const bf16_t* Xs = reinterpret_cast<const bf16_t*>(smem + a.off_x);
  const unsigned char* blob = a.blob;
  // (groups: two copies by block parity - a workgroup that is a block ahead writes the copy nobody still reads)
  unsigned char* dimg = reinterpret_cast<unsigned char*>(a.dscratch) + (size_t)b * a.dstride + (a.G > 1 && (bi & 1) ? (size_t)(a.dstride >> 1) : 0);
  float hacc = 0.f, b1v = 0.f;                                  // mm lanes: reduce-FC partial of hidden unit sj over channel slice spart; its bias
  const int sj = ml >> 3, spart = ml & 7;
  const bool se_lane = mm && sj < sqp;

  // ---- mm helpers ----
  u32x4 wfr[KSE]; f32x4 bexp = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto load_w = [&](int c) {
    const unsigned char* p = blob + L.off_we + ((size_t)((c * 8 + t) * KSE) * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KSE; ks++) wfr[ks] = late_ldg(p + (size_t)ks * 1024);
    bexp = *reinterpret_cast<const f32x4*>(blob + L.off_be + (size_t)(c * LATE_CC + t * 16 + 4 * g) * 4);
  };
  constexpr int NWV = (KK * 32 + LATE_MM_LANES - 1) / LATE_MM_LANES;   // 16-byte vectors of depthwise weights per mm lane and chunk
  u32x4 wdv[NWV], bdv;
  auto load_dw = [&](int c) {
    const unsigned char* p = blob + L.off_wdw + (size_t)c * KK * LATE_CC * 4;
#pragma unroll
    for (int i = 0; i < NWV; i++) { const int v = ml + i * LATE_MM_LANES; wdv[i] = late_ldg(p + (size_t)(v < KK * 32 ? v : 0) * 16); }
    bdv = late_ldg(blob + L.off_bdw + (size_t)c * LATE_CC * 4 + (size_t)(ml < 32 ? ml : 0) * 16);
  };
  auto park_dw = [&](int c) {
    unsigned char* wd = smem + a.off_wdw + ((c - c_lo) & 1) * a.wdw_stride;
#pragma unroll
    for (int i = 0; i < NWV; i++) { const int v = ml + i * LATE_MM_LANES; if (v < KK * 32) *reinterpret_cast<u32x4*>(wd + v * 16) = wdv[i]; }
    if (ml < 32) *reinterpret_cast<u32x4*>(smem + a.off_bias + ((c - c_lo) & 1) * 512 + ml * 16) = bdv;
  };
  auto expand = [&](int c) {
    unsigned char* E = smem + a.off_e + ((c - c_lo) & 1) * a.e_stride;
    f32x4 acc[4] = {bexp, bexp, bexp, bexp};
#pragma unroll
    for (int ks = 0; ks < KSE; ks++) {
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        const u32x4 x = *reinterpret_cast<const u32x4*>(Xs + (mt * 16 + r) * XP + ks * 32 + 8 * g);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[ks]), __builtin_bit_cast(bf16x8, x), acc[mt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      const int m = mt * 16 + r, pos = ((m >> 3) + PAD) * PW + (m & 7) + PAD;
      float v[4] = {acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]};
      swish_n<true, 4>(v);
      Vec8<true>::store4(E, (int64_t)pos * LATE_EP + t * 16 + 4 * g, v);
    }
  };
  f32x4 w1v[4];
  auto load_w1 = [&](int c) {
    const unsigned char* p = blob + L.off_w1 + ((size_t)(c * sqp + (se_lane ? sj : 0)) * LATE_CC + spart * 16) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) w1v[i] = *reinterpret_cast<const f32x4*>(p + i * 16);
  };
  auto se_partial = [&](int c) {
    const f32x4* cs = reinterpret_cast<const f32x4*>(smem + a.off_csum + ((c - c_lo) & 1) * 512 + spart * 64);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { const f32x4 x = cs[i]; s = fmaf(w1v[i][0], x[0], s); s = fmaf(w1v[i][1], x[1], s); s = fmaf(w1v[i][2], x[2], s); s = fmaf(w1v[i][3], x[3], s); }
    if (se_lane) hacc += s;
  };

  // ---- dw: depthwise conv of chunk c from E[c & 1] ----
  auto dwconv = [&](int c) {
    const bf16_t* E = reinterpret_cast<const bf16_t*>(smem + a.off_e + ((c - c_lo) & 1) * a.e_stride);
    const float* wd = reinterpret_cast<const float*>(smem + a.off_wdw + ((c - c_lo) & 1) * a.wdw_stride);
    const float* bd = reinterpret_cast<const float*>(smem + a.off_bias + ((c - c_lo) & 1) * 512);
    const int yx = lane & 15, cgw = (lane >> 4) & 1, rg = lane >> 5;
    const int cg = wave * 2 + cgw, y = yx >> 1, xh = yx & 1;
    f32x2_t acc[4][4];                                         // [pixel of the strip][channel pair]: every multiply-add is a v_pk_fma_f32
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bd + cg * 8), b1 = *reinterpret_cast<const f32x4*>(bd + cg * 8 + 4);
      const f32x2_t z = {0.f, 0.f};
#pragma unroll
      for (int px = 0; px < 4; px++) {
        acc[px][0] = rg ? z : (f32x2_t){b0[0], b0[1]}; acc[px][1] = rg ? z : (f32x2_t){b0[2], b0[3]};
        acc[px][2] = rg ? z : (f32x2_t){b1[0], b1[1]}; acc[px][3] = rg ? z : (f32x2_t){b1[2], b1[3]};
      }
    }
    const int ky0 = rg ? KH : 0, nrows = rg ? KS - KH : KH;
#pragma unroll 1
    for (int it = 0; it < KH; it++) {
      if (it < nrows) {
        const int ky = ky0 + it;
        const bf16_t* erow = E + ((y + ky) * PW + xh * 4) * LATE_EP + cg * 8;
        // the row's KS weight vectors stay in registers; the KS + 3 input positions are unpacked one at a time (a bf16 pair word IS
        // a channel pair: shift / mask -> the two halves of a packed operand) and feed the (up to four) pixels whose window holds
        // them - per accumulator the taps still arrive in kx order (k_mbf.hip's order)
        f32x2_t w[KS][4];
#pragma unroll
        for (int kx = 0; kx < KS; kx++) {
          const f32x4* wp = reinterpret_cast<const f32x4*>(wd + (ky * KS + kx) * LATE_CC + cg * 8);
          const f32x4 w0 = wp[0], w1 = wp[1];
          w[kx][0] = (f32x2_t){w0[0], w0[1]}; w[kx][1] = (f32x2_t){w0[2], w0[3]}; w[kx][2] = (f32x2_t){w1[0], w1[1]}; w[kx][3] = (f32x2_t){w1[2], w1[3]};
        }
#pragma unroll
        for (int j = 0; j < NXP; j++) {
          const u32x4 raw = *reinterpret_cast<const u32x4*>(erow + j * LATE_EP);
          f32x2_t ev[4];
#pragma unroll
          for (int q = 0; q < 4; q++) ev[q] = (f32x2_t){__uint_as_float(raw[q] << 16), __uint_as_float(raw[q] & 0xffff0000u)};
#pragma unroll
          for (int px = 0; px < 4; px++) {
            const int kx = j - px;
            if (kx >= 0 && kx < KS) {
#pragma unroll
              for (int q = 0; q < 4; q++) acc[px][q] = __builtin_elementwise_fma(ev[q], w[kx][q], acc[px][q]);
            }
          }
        }
      }
    }
    if (c == c_lo + 1) LSTAMP(41);
    // the two tap-row halves meet: lanes 0-31 finish pixels 0,1 of the strip, lanes 32-63 pixels 2,3 (first + second)
    float cs8[8];
#pragma unroll
    for (int p = 0; p < 2; p++) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[p][q][h]), __float_as_uint(acc[2 + p][q][h]), false, false);
          v[2 * q + h] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
      swish_n<true, 8>(v);
#pragma unroll
      for (int ch = 0; ch < 8; ch++) cs8[ch] = p ? cs8[ch] + v[ch] : v[ch];
      const int du = p * 32 + y * 4 + xh * 2 + rg;          // unit of the channel group's 64: [pixel-of-pair p][row y][strip xh][half rg] - for one p the 32
                                                            // lanes of a channel group write 512 contiguous bytes: whole 128-byte lines per store instruction
      if (!coh) Vec8<true>::store(dimg, ((int64_t)(c * 16 + cg) * 64 + du) * 8, v);
      else {
        u32x4 pk;
#pragma unroll
        for (int q = 0; q < 4; q++) pk[q] = pack_bf16x2(v[2 * q], v[2 * q + 1]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(late_v4, pk), late_rsrc(dimg), ((c * 16 + cg) * 64 + du) * 16, 0, LATE_COHERENT);
      }
    }
    if (c == c_lo + 1) LSTAMP(42);
    // channel sums over the 32 lanes of this channel group (16 strips x 2 halves): DPP adds inside the 16-lane row (quad, quad pair,
    // half row, row), then the two halves of the wave (fixed order, no LDS round trips: five ds_bpermute per value cost 1 us per chunk)
#pragma unroll
    for (int ch = 0; ch < 8; ch++) {
      float s = cs8[ch];
      s += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s), 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
      s += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s), 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
      s += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s), 0x141, 0xf, 0xf, false));    // row_half_mirror
      s += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s), 0x140, 0xf, 0xf, false));    // row_mirror
      const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
      cs8[ch] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (yx == 0 && rg == 0) {
      f32x4* d = reinterpret_cast<f32x4*>(smem + a.off_csum + ((c - c_lo) & 1) * 512 + cg * 32);
      d[0] = (f32x4){cs8[0], cs8[1], cs8[2], cs8[3]}; d[1] = (f32x4){cs8[4], cs8[5], cs8[6], cs8[7]};
    }
  };

  // ---- chunk loop: the two roles run their own loops (one barrier per chunk each - a role's registers are not live in the other's code) ----
  LSTAMP(0);
  if (mm) {
    // Every load of an iteration is consumed in the NEXT one (the compiler waits with vmcnt(0) behind the loop edge: a load
    // requested at the top of an iteration and the fragments requested a chunk ago are then waited for together - measured: the
    // expand phase took 4.5 us of a 4.9 us chunk that way and the depthwise waves idled 2 us per chunk at the barrier)
    load_w(c_lo); load_dw(c_lo);
    expand(c_lo); park_dw(c_lo);
    if (c_lo + 1 < c_hi) { load_w(c_lo + 1); load_dw(c_lo + 1); }
    __syncthreads();
    LSTAMP(1);
    for (int c = c_lo; c < c_hi; c++) {
      if (c == c_lo + 1) LSTAMP(32);
      if (c + 1 < c_hi) expand(c + 1);
      if (c == c_lo + 1) LSTAMP(33);
      if (c > c_lo) se_partial(c - 1);                           // (its reduce-FC slice was requested in the previous iteration)
      if (c == c_lo + 1) LSTAMP(34);
      if (c + 1 < c_hi) park_dw(c + 1);
      if (c + 2 < c_hi) { load_w(c + 2); load_dw(c + 2); }
      load_w1(c);
      if (c == c_hi - 1) b1v = *reinterpret_cast<const float*>(blob + L.off_b1 + (size_t)(se_lane ? sj : 0) * 4);
      if (c == c_lo + 1) LSTAMP(35);
      __syncthreads();
      LSTAMP(8 + c - c_lo);
    }
  } else {
    late_zero_halo<KS>(smem, a);                                // (the depthwise waves: idle until the first chunk is expanded)
    __syncthreads();
    LSTAMP(1);
    for (int c = c_lo; c < c_hi; c++) {
      if (c == c_lo + 1) LSTAMP(40);
      dwconv(c);
      if (c == c_lo + 1) LSTAMP(43);
      __syncthreads();
      LSTAMP(8 + c - c_lo);
    }
  }
  LSTAMP(2);

  // ---- squeeze-excite: hidden vector, scale; As = depthwise output x scale ----
  float* hid_s = reinterpret_cast<float*>(smem + a.off_hid);
  float* scale_s = reinterpret_cast<float*>(smem + a.off_scale);
  // Nothing below depends on the hidden vector until the expand FC: this thread's rows of that FC (two of Cexp <= 2048) and its
  // share of the depthwise outputs (Cexp / 128 16-byte units, back from L2) are requested first and land under the two barriers
  constexpr int W2V = 6;                                       // 16-byte vectors per row (sqp <= 48)
  constexpr int NDU = 9;                                       // depthwise-output units per thread (Cexp <= 1152)
  const int nv2 = sqp >> 3;
  u32x4 w2a[W2V], w2b[W2V], dreg[NDU];
  const int k0 = tid, k1 = tid + LATE_THREADS;
  {
    const unsigned char* p0 = blob + L.off_w2 + (size_t)min(k0, L.Cexp - 1) * sqp * 2;
    const unsigned char* p1 = blob + L.off_w2 + (size_t)min(k1, L.Cexp - 1) * sqp * 2;
#pragma unroll
    for (int v = 0; v < W2V; v++) { w2a[v] = late_ldg(p0 + min(v, nv2 - 1) * 16); w2b[v] = late_ldg(p1 + min(v, nv2 - 1) * 16); }
  }
  const float b2a = *reinterpret_cast<const float*>(blob + L.off_b2 + (size_t)min(k0, L.Cexp - 1) * 4);
  const float b2b = *reinterpret_cast<const float*>(blob + L.off_b2 + (size_t)min(k1, L.Cexp - 1) * 4);
  if (!coh) {
#pragma unroll
    for (int i = 0; i < NDU; i++) dreg[i] = late_ldg(dimg + (size_t)(tid + LATE_THREADS * min(i, NC - 1)) * 16);
  }
  float hsum = 0.f;
  if (mm) {
    se_partial(c_hi - 1);                                      // (its reduce-FC slice and bias were requested inside the last chunk)
    float sv = hacc;
    sv += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(sv), 0xB1, 0xf, 0xf, false));    // the eight channel slices of a hidden unit: lanes 8 j .. 8 j + 7
    sv += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(sv), 0x4E, 0xf, 0xf, false));
    sv += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(sv), 0x141, 0xf, 0xf, false));
    hsum = sv;
  }
  if (coh) {
    // ---- the seam of the block: this workgroup's third of the depthwise outputs and of the reduce-FC sums is on its way to memory
    //      (write-through stores); drain, arrive at the group's counter, wait for the others ----
    float* hp = a.hpart + ((size_t)(b * 2 + (bi & 1)) * G) * 64;
    // (hand-off rules of MI355X_MICROARCH.md "Valid forms": every 128-byte line written whole by ONE store instruction of one wave,
    //  loads of 4 or 16 bytes - the sums go through LDS so that one wave stores the 64 floats)
    if (se_lane && spart == 0) hid_s[sj] = hsum;
    __syncthreads();
    if (wave == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lane < sqp ? hid_s[lane] : 0.f), late_rsrc(hp), (gw * 64 + lane) * 4, 0, LATE_COHERENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the hand-off does not depend on where the members run: the launch places a group on workgroup ids that are equal mod 8 = one
    //  XCD and its L2 under round-robin dispatch, for speed; HEP_LATE_XCD=1 places it on consecutive ids = three XCDs and gives the
    //  same bits - tested)
    int* timed_out = reinterpret_cast<int*>(smem + a.off_hid) + 63;           // (workgroup-uniform flag, behind the hidden units)
    if (tid == 0) {
      // one counter per block and image (words 0 .. nblk - 1 of the image's 64-byte line), each used once per launch and ZEROED BY
      // THE KERNEL ITSELF: whoever has passed meeting bi knows that every member has left meeting bi - 1, so member 0 clears that
      // counter; the last block's counter is cleared at the next launch's first meeting.  The line lives in an allocation of its
      // own (Session::d_sync, zeroed once), never in the activation arena.
      // (A hipMemsetAsync in front of the launch did this first: as a memset node of the captured graph it left garbage in the
      //  upper half of the counter word on every REPLAY - eager launches were fine - every meeting fell through and the groups read
      //  each other's bytes before they were written: right on the first launch, 10-30 % off on later ones.)
      unsigned* line = a.counters + (size_t)b * 16;
      unsigned* cnt = line + bi;
      if (bi == 0) *timed_out = 0;
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (bounded: ~1 s of polling - two launches sharing one session's counters, a misuse, must end in NaN outputs, never in a hung GPU)
      int spins = 0;
      while (__hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
      if (spins >= (1 << 20)) *timed_out = 1;
      if (gw == 0) __hip_atomic_exchange(line + (bi == 0 ? a.nblk - 1 : bi - 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    LSTAMP(44);
#pragma unroll
    for (int i = 0; i < NDU; i++)
      dreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(late_rsrc(dimg), (tid + LATE_THREADS * min(i, NC - 1)) * 16, 0, LATE_COHERENT));
    if (se_lane && spart == 0) {
      hsum = 0.f;
      for (int q = 0; q < G; q++) hsum += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(late_rsrc(hp), (q * 64 + sj) * 4, 0, LATE_COHERENT));     // (fixed order: every workgroup of the group gets the same bits)
    }
  }
  if (se_lane && spart == 0) hid_s[sj] = sj < L.sq ? swishf(fmaf(hsum, L.inv_hw, b1v)) : 0.f;
  __syncthreads();
  {
    auto row_scale = [&](const u32x4 (&w)[W2V], float be) {
      float e0 = 0.f, e1 = 0.f;
#pragma unroll
      for (int v = 0; v < W2V; v++) {
        if (v < nv2) {
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(hid_s + v * 8), h1 = *reinterpret_cast<const f32x4*>(hid_s + v * 8 + 4);
          e0 = fmaf(__uint_as_float(w[v][0] << 16), h0[0], e0); e1 = fmaf(__uint_as_float(w[v][0] & 0xffff0000u), h0[1], e1);
          e0 = fmaf(__uint_as_float(w[v][1] << 16), h0[2], e0); e1 = fmaf(__uint_as_float(w[v][1] & 0xffff0000u), h0[3], e1);
          e0 = fmaf(__uint_as_float(w[v][2] << 16), h1[0], e0); e1 = fmaf(__uint_as_float(w[v][2] & 0xffff0000u), h1[1], e1);
          e0 = fmaf(__uint_as_float(w[v][3] << 16), h1[2], e0); e1 = fmaf(__uint_as_float(w[v][3] & 0xffff0000u), h1[3], e1);
        }
      }
      return sigmoidf((e0 + e1) + be);
    };
    if (k0 < L.Cexp) scale_s[k0] = row_scale(w2a, b2a);
    if (k1 < L.Cexp) scale_s[k1] = row_scale(w2b, b2b);
  }
  __syncthreads();
  LSTAMP(3);
  {
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);
    const int AP = L.Cexp + 8, du = tid & 63;
    const int px = ((du >> 2) & 7) * 8 + ((du >> 1) & 1) * 4 + (du & 1) * 2 + (du >> 5);        // the depthwise waves' unit order (dwconv)
#pragma unroll
    for (int i = 0; i < NDU; i++) {
      if (i < NC) {
        const int cgi = wave + 16 * i;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale_s + cgi * 8), s1 = *reinterpret_cast<const f32x4*>(scale_s + cgi * 8 + 4);
        const float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        u32x4 raw = dreg[i];
#pragma unroll
        for (int q = 0; q < 4; q++)
          raw[q] = pack_bf16x2(__uint_as_float(raw[q] << 16) * sc[2 * q], __uint_as_float(raw[q] & 0xffff0000u) * sc[2 * q + 1]);
        *reinterpret_cast<u32x4*>(As + (int64_t)px * AP + cgi * 8) = raw;
      }
    }
  }
  __syncthreads();
  LSTAMP(4);
  if (L.ntw == 2) late_project<2>(a, L, smem, b, last_block, bi, gw);
  else late_project<3>(a, L, smem, b, last_block, bi, gw);
  LSTAMP(5);
}

__global__ __launch_bounds__(LATE_THREADS) void late_kernel(LateArgs a_by_value) {
  // The block table is indexed at run time: read through the by-value parameter the compiler copies all of it to scratch (94 stores
  // at kernel entry, a scratch load per field).  The kernel-argument segment itself is addressable constant memory: scalar loads.
  const LateArgs& a = *(const LateArgs*)(const __attribute__((address_space(4))) LateArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)a_by_value;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, a.lds_bytes);
  // workgroup -> (image, member of the image's group): the G workgroups of an image get linear ids that are equal mod 8 - the
  // hardware deals consecutive ids round-robin over the eight XCDs, so a group shares an XCD and its L2 (speed only)
  int b, gw;
  {
    const int L_ = blockIdx.x, G = a.G;
    if (G > 1 && !a.cross_xcd) { const int q = L_ >> 3; gw = q % G; b = (L_ & 7) + 8 * (q / G); if (b >= a.B) return; }      // (the grid is padded to whole octets of images)
    else { b = L_ / G; gw = L_ - b * G; }
  }
  {   // the first block's input tile: [64][Cin] -> LDS rows of Cin + 8
    const int Cin = a.blk[0].Cin, XP = Cin + 8, vpr = Cin >> 3;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * 64 * Cin * 2;
    bf16_t* Xs = reinterpret_cast<bf16_t*>(smem + a.off_x);
    for (int u = threadIdx.x; u < 64 * vpr; u += LATE_THREADS) {
      const int m = u / vpr, v = u - m * vpr;
      *reinterpret_cast<u32x4*>(Xs + m * XP + v * 8) = late_ldg(src + (size_t)u * 16);
    }
  }
  __syncthreads();
  for (int i = 0; i < a.nblk; i++) {
    const LateBlock& L = a.blk[i];
    const bool last = i + 1 == a.nblk;
    if (L.k == 5) late_block<5, 6>(a, L, smem, b, last, i, gw);
    else late_block<3, 6>(a, L, smem, b, last, i, gw);
  }
}

#ifdef HEP_LATE_TRACE
extern "C" int hep_dbg_late_trace(unsigned long long* host, int max_words, int enable) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)64 * 16 * LATE_MAX_BLOCKS * 64;      // up to 64 workgroups
  if (!buf) { if (hipMalloc((void**)&buf, cap * 8) != hipSuccess) return -1; }
  if (enable) hipMemset(buf, 0, cap * 8);
  unsigned long long* p = enable ? buf : nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_late_trace), &p, sizeof p);
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)std::min<size_t>(max_words, cap) * 8, hipMemcpyDeviceToHost); }
  return (int)cap;
}
#endif

// ---- host side ----
int late_block_supported(int Cin, int Cexp, int N, int k, int stride, int H, int W, int sq) {
  // (KSE = Cin / 32 is a template parameter: 192 input channels - EfficientNet-B0's last stage - is the instantiation that exists)
  return stride == 1 && H == 8 && W == 8 && Cin == 192 && Cexp % LATE_CC == 0 && Cexp <= 2 * LATE_THREADS && (Cexp / 32) % 2 == 0 && N % 8 == 0 && N >= 16 &&
         (k == 3 || k == 5) && sq >= 1 && sq <= 48 && Cexp <= 9 * LATE_CC && (N + 15) / 16 <= 21;
}

int late_layout(LateArgs* a) {
  int kmax = 3, cin_max = 0, cexp_max = 0, part_max = 0;
  for (int i = 0; i < a->nblk; i++) {
    LateBlock& L = a->blk[i];
    kmax = std::max(kmax, L.k); cin_max = std::max(cin_max, L.Cin); cexp_max = std::max(cexp_max, L.Cexp);
    cin_max = std::max(cin_max, L.N);                             // the project conv leaves the next input tile (groups: every block's output tile) where this one was
    const int nt = (L.N + 15) / 16;
    L.ntw = nt <= 16 ? 2 : 3; L.ng = (nt + L.ntw - 1) / L.ntw;
    if (2 * L.ng > LATE_THREADS / 64) return 0;
    part_max = std::max(part_max, L.ng * L.ntw * 4 * 1024);
  }
  const int pw = kmax + 7;
  a->off_e = 0; a->e_stride = pw * pw * LATE_EP * 2;
  a->off_wdw = a->off_e + 2 * a->e_stride; a->wdw_stride = kmax * kmax * LATE_CC * 4;
  a->off_bias = a->off_wdw + 2 * a->wdw_stride;
  a->off_csum = a->off_bias + 2 * 512;
  a->off_x = a->off_csum + 2 * 512;
  if (a->off_x < part_max) a->off_x = (part_max + 15) & ~15;      // the K-half partial sums of the project conv start at 0
  const int chunk_end = a->off_x + 64 * (cin_max + 8) * 2;
  const int as_bytes = 64 * (cexp_max + 8) * 2;
  a->off_scale = (std::max(chunk_end, as_bytes) + 15) & ~15;
  a->off_hid = a->off_scale + cexp_max * 4;
  a->lds_bytes = a->off_hid + 64 * 4;
  return a->lds_bytes <= 160 * 1024 ? a->lds_bytes : 0;
}

int late_prepare(void) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(late_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : -1;
}

void launch_late(const LateArgs& a, hipStream_t s) {
  const int nwg = a.G > 1 && !a.cross_xcd ? ((a.B + 7) / 8) * 8 * a.G : a.B * a.G;
  hipLaunchKernelGGL(late_kernel, dim3(nwg), dim3(LATE_THREADS), (size_t)a.lds_bytes, s, a);
}
