// k_sep.hip - fused SeparableConvBlock for the BiFPN nodes and the five head towers on gfx950:
//
//   [ weighted fusion of 2-3 maps (nearest x2 up / zero-padded 3x3/2 max-pool gathers) + swish ]
//     -> depthwise 3x3 SAME (no bias) -> pointwise 1x1 (+bias, BN folded) -> [swish | sigmoid]
//
// replaces, per BiFPN node, `swish(w0*a + w1*up(b) [+ w2*pool(c)])` + `SeparableConvBlock`
// (reference efficientdet/model.py:212-264, 42-52) and, per head layer,
// `conv(feat); bn(feat); swish(feat)` / the header conv + permute/view/cat
// (efficientdet/model.py:361-417; hmdegopose/model.py:55-90,127-156,191-228).
//
// One workgroup (8 waves) = one 8x8 output tile of one image of one "segment" (a node, or one
// (head, level, column-chunk) triple); a single launch covers every segment of a layer, e.g. all
// 5 heads x 5 levels of tower layer i.  These maps are tiny (32x32 .. 2x2 per image), so the
// kernel is latency-bound: the design goal is the shortest dependent chain per workgroup.
//   phase 0  issue the first pointwise-weight fragments (registers) and copy the depthwise
//            weights + bias to LDS - nothing below depends on them until phase 3
//   phase 1  fused (+swish) 10x10 halo of the depthwise input -> LDS fp32, zero outside the map
//   phase 2  depthwise 3x3 from LDS -> MFMA operand tile [64 pixels][C] in LDS
//   phase 3  1x1 conv as the transposed MFMA product of k_pw.hip (W fragment = A operand,
//            pixels = B operand); (m-tile, n-tile) pairs are dealt round-robin to the 8 waves and
//            the weight fragments run 4 deep ahead of the MFMAs in a register ring.
// Head outputs are written directly at their anchor offset in the [B, N_anchors, K] result (no
// permute / cat pass).
#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

#define SEP_THREADS 512
#define SEP_WAVES 8

__host__ __device__ static inline int sep_cp(int C) { return C + 4; }                 // halo row pitch (floats)
__host__ __device__ static inline int sep_ca(int C, int bf16) { return bf16 ? C + 8 : C + 4; }
__host__ __device__ static inline size_t sep_off_atile(int C) { return (size_t)100 * sep_cp(C) * 4; }
__host__ __device__ static inline size_t sep_off_wdw(int C, int bf16) { return sep_off_atile(C) + (size_t)64 * sep_ca(C, bf16) * (bf16 ? 2 : 4); }
__host__ __device__ static inline size_t sep_off_bias(int C, int bf16) { return sep_off_wdw(C, bf16) + (size_t)9 * C * 4; }

size_t sep_lds_bytes(int C, int bf16) { return sep_off_bias(C, bf16) + (size_t)SEP_MAX_TILES_N * 16 * 4; }

template <bool BF16>
__device__ __forceinline__ void gather_src(const SepSeg& sg, int i, int b, int y, int x, int c0, float v[8]) {
  typedef Vec8<BF16> V;
  const int C = sg.C, sh = sg.sh[i], sw = sg.sw[i];
  const int64_t img = (int64_t)b * sh * sw * C;
  if (sg.kind[i] == SRC_SAME) {
    V::load(sg.src[i], img + ((int64_t)y * sw + x) * C + c0, v);
  } else if (sg.kind[i] == SRC_UP) {
    V::load(sg.src[i], img + ((int64_t)(y >> 1) * sw + (x >> 1)) * C + c0, v);
  } else {   // SRC_DOWN: 3x3/2 max-pool, zero padding takes part in the max
    // one row of the window (3 loads in flight) at a time keeps the register footprint small
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      float t[3][8];
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int iy = 2 * y - sg.pool_pad[i] + ky, ix = 2 * x - sg.pool_pad[i] + kx;
#pragma unroll
        for (int c = 0; c < 8; c++) t[kx][c] = 0.f;
        if (iy >= 0 && iy < sh && ix >= 0 && ix < sw) V::load(sg.src[i], img + ((int64_t)iy * sw + ix) * C + c0, t[kx]);
      }
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float m = fmaxf(fmaxf(t[0][c], t[1][c]), t[2][c]);
        v[c] = ky == 0 ? m : fmaxf(v[c], m);
      }
    }
  }
}

template <bool BF16>
__global__ __launch_bounds__(SEP_THREADS, 4) void sep_kernel(SepArgs a) {
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
  constexpr int KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- locate the segment and tile ----
  // one table lookup, then the whole descriptor by VALUE: it lands in scalar registers once, instead
  // of being re-read from global memory (a dependent round trip) at every use after a barrier
  const SepSeg sg = a.segs[a.tile_seg[blockIdx.x]];
  const int C = sg.C, CG = C >> 3, h = sg.h, w = sg.w;
  const int t = blockIdx.x - sg.tile_begin;
  const int b = blockIdx.y;                 // grid = (tiles of one image over all segments, batch)
  const int y0 = (t / sg.tiles_x) * 8, x0 = (t % sg.tiles_x) * 8;
  const int CP = sep_cp(C), CA = sep_ca(C, BF16);
  float* halo = reinterpret_cast<float*>(smem);
  T* atile = reinterpret_cast<T*>(smem + sep_off_atile(C));
  float* wdw_s = reinterpret_cast<float*>(smem + sep_off_wdw(C, BF16));
  float* bias_s = reinterpret_cast<float*>(smem + sep_off_bias(C, BF16));

  // ---- phase 0: weight prefetch (independent of the activations) ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int rows_valid = min(8, h - y0);
  const int mtv = (rows_valid + 1) >> 1;                // m-tiles (2 tile rows = 16 pixels) with any valid pixel
  const int ksteps = (C + KSTEP - 1) / KSTEP;
  const int nitems = mtv * sg.tilesN * ksteps;          // (pair, kstep) items; pair = nt * mtv + mt
  const T* W = reinterpret_cast<const T*>(sg.wpw);
  auto wload = [&](int it) -> raw_t {                   // it = this wave's it-th (pair, kstep)
    raw_t v = {};
    const int pair = wave + SEP_WAVES * (it / ksteps), ks = it % ksteps;
    const int k = ks * KSTEP + KLANE * g;
    if (pair < mtv * sg.tilesN && k < C) v = *reinterpret_cast<const raw_t*>(W + (int64_t)((pair / mtv) * 16 + r) * C + k);
    return v;
  };
  raw_t wring[4];
#pragma unroll
  for (int q = 0; q < 4; q++) wring[q] = wload(q);
  for (int i = threadIdx.x; i < 9 * C; i += SEP_THREADS) wdw_s[i] = sg.wdw[i];
  for (int i = threadIdx.x; i < sg.tilesN * 16; i += SEP_THREADS) bias_s[i] = sg.bias[i];

  // ---- phase 1: fused (+swish) 10x10 halo of the depthwise input, zero outside the image ----
  for (int item = threadIdx.x; item < 100 * CG; item += SEP_THREADS) {
    const int pos = item / CG, cg = item % CG;
    const int y = y0 + pos / 10 - 1, x = x0 + pos % 10 - 1;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (y >= 0 && y < h && x >= 0 && x < w) {
#pragma unroll
      for (int i = 0; i < HEP_MAX_SRC; i++) {
        if (i >= sg.nsrc) break;
        float s[8];
        gather_src<BF16>(sg, i, b, y, x, cg * 8, s);
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = fmaf(sg.fw[i], s[c], v[c]);
      }
      if (sg.pre_act) {
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = swishf(v[c]);
      }
    }
    f32x4* hp = reinterpret_cast<f32x4*>(halo + pos * CP + cg * 8);
    hp[0] = (f32x4){v[0], v[1], v[2], v[3]};
    hp[1] = (f32x4){v[4], v[5], v[6], v[7]};
  }
  __syncthreads();

  // ---- phase 2: depthwise 3x3 -> operand tile [64 pixels][C] ----
  for (int item = threadIdx.x; item < 64 * CG; item += SEP_THREADS) {
    const int p = item / CG, cg = item % CG;
    const int py = p >> 3, px = p & 7;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const f32x4* hp = reinterpret_cast<const f32x4*>(halo + ((py + ky) * 10 + px + kx) * CP + cg * 8);
        const f32x4* wp = reinterpret_cast<const f32x4*>(wdw_s + (ky * 3 + kx) * C + cg * 8);
        const f32x4 h0 = hp[0], h1 = hp[1], w0 = wp[0], w1 = wp[1];
#pragma unroll
        for (int c = 0; c < 4; c++) { acc[c] = fmaf(h0[c], w0[c], acc[c]); acc[4 + c] = fmaf(h1[c], w1[c], acc[4 + c]); }
      }
    V::store(atile, (int64_t)p * CA + cg * 8, acc);
  }
  __syncthreads();

  // ---- phase 3: pointwise conv, D[n, pixel] = W[n,:] . tile[pixel,:] ----
  const int my_pairs = (mtv * sg.tilesN - wave + SEP_WAVES - 1) / SEP_WAVES;    // pairs wave, wave+8, ...
  if (my_pairs <= 0) return;
  const int my_items = my_pairs * ksteps;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto step = [&](int it, raw_t wfrag) {
    const int pair = wave + SEP_WAVES * (it / ksteps), ks = it % ksteps;
    const int mt = pair % mtv, nt = pair / mtv;
    const int m = mt * 16 + r;
    const int k = ks * KSTEP + KLANE * g;
    raw_t xa = {};
    if (k < C) xa = *reinterpret_cast<const raw_t*>(atile + (int64_t)m * CA + k);
    if constexpr (BF16) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[q], xa[q], acc, 0, 0, 0);
    }
    if (ks != ksteps - 1) return;
    // ---- epilogue of this (m-tile, n-tile) pair ----
    const int y = y0 + (m >> 3), x = x0 + (m & 7);
    const int n = nt * 16 + 4 * g;
    if (y < h && x < w && n < sg.N) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + n);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = apply_act(acc[q] + bias[q], sg.act);
      const int64_t obase = (int64_t)b * sg.out_bstride + sg.out_off + ((int64_t)y * w + x) * sg.out_rowstride;
      if (sg.out_f32) {
        float* o = reinterpret_cast<float*>(sg.out) + obase;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int nn = n + q + sg.n_base;
          if (n + q < sg.N) o[(nn / sg.col_kin) * sg.col_kout + nn % sg.col_kin + sg.col_off] = v[q];
        }
      } else {
        V::store4(sg.out, obase + n, v);     // N is a multiple of 8 for every non-header layer
      }
    }
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  for (int it = 0; it < my_items; it += 4) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (it + q < my_items) {
        const raw_t wf = wring[q];
        wring[q] = wload(it + q + 4);       // refill the slot 4 items ahead (zero past the end)
        step(it + q, wf);
      }
    }
  }
  (void)nitems;
}

int sep_prepare(void) {
  hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(sep_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(sep_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : -1;
}

void launch_sep(const SepArgs& a, hipStream_t s) {
  dim3 grid(a.total_tiles, a.B);
  if (a.bf16) hipLaunchKernelGGL(sep_kernel<true>, grid, dim3(SEP_THREADS), a.lds_bytes, s, a);
  else hipLaunchKernelGGL(sep_kernel<false>, grid, dim3(SEP_THREADS), a.lds_bytes, s, a);
}
