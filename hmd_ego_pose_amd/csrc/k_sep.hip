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
// One workgroup = one 8x8 output tile of one image of one "segment" (a node, or one
// (head, level) pair); a single launch covers every segment of a layer, e.g. all 5 heads x 5
// levels of tower layer i.  The 10x10 halo of the fused map lives in LDS as fp32, the depthwise
// result goes to LDS as the MFMA operand tile [64 pixels][C], and the 1x1 conv is the same
// transposed MFMA product as k_pw.hip (W fragment = A operand, pixels = B operand) so each lane
// ends with 4 consecutive output channels of one pixel.  Head outputs are written directly at
// their anchor offset in the [B, N_anchors, K] result (no permute / cat pass).
#include "hep_dev.h"
#include "hep_internal.h"

__host__ __device__ static inline int sep_cp(int C) { return C + 4; }                 // halo row pitch (floats)
__host__ __device__ static inline int sep_ca(int C, int bf16) { return bf16 ? C + 8 : C + 4; }

size_t sep_lds_bytes(int C, int bf16) {
  return (size_t)100 * sep_cp(C) * 4 + (size_t)64 * sep_ca(C, bf16) * (bf16 ? 2 : 4);
}

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
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int iy = 2 * y - sg.pool_pad[i] + ky, ix = 2 * x - sg.pool_pad[i] + kx;
        float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (iy >= 0 && iy < sh && ix >= 0 && ix < sw) V::load(sg.src[i], img + ((int64_t)iy * sw + ix) * C + c0, t);
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = first ? t[c] : fmaxf(v[c], t[c]);
        first = false;
      }
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void sep_kernel(SepArgs a) {
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- locate the segment and tile ----
  int si = 0;
  for (int i = 1; i < a.nseg; i++) if ((int)blockIdx.x >= a.segs[i].tile_begin) si = i;
  const SepSeg& sg = a.segs[si];
  const int C = sg.C, CG = C >> 3, h = sg.h, w = sg.w;
  const int local = blockIdx.x - sg.tile_begin;
  const int b = blockIdx.y, t = local;     // grid = (tiles of one image over all segments, batch)
  const int y0 = (t / sg.tiles_x) * 8, x0 = (t % sg.tiles_x) * 8;
  const int CP = sep_cp(C), CA = sep_ca(C, BF16);
  float* halo = reinterpret_cast<float*>(smem);
  T* atile = reinterpret_cast<T*>(smem + (size_t)100 * CP * 4);

  // ---- phase 1: fused (+swish) 10x10 halo of the depthwise input, zero outside the image ----
  for (int item = threadIdx.x; item < 100 * CG; item += 256) {
    const int pos = item / CG, cg = item % CG;
    const int y = y0 + pos / 10 - 1, x = x0 + pos % 10 - 1;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (y >= 0 && y < h && x >= 0 && x < w) {
      for (int i = 0; i < sg.nsrc; i++) {
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
  for (int item = threadIdx.x; item < 64 * CG; item += 256) {
    const int p = item / CG, cg = item % CG;
    const int py = p >> 3, px = p & 7;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const f32x4* hp = reinterpret_cast<const f32x4*>(halo + ((py + ky) * 10 + px + kx) * CP + cg * 8);
        const f32x4* wp = reinterpret_cast<const f32x4*>(sg.wdw + (ky * 3 + kx) * C + cg * 8);
        const f32x4 h0 = hp[0], h1 = hp[1], w0 = wp[0], w1 = wp[1];
#pragma unroll
        for (int c = 0; c < 4; c++) { acc[c] = fmaf(h0[c], w0[c], acc[c]); acc[4 + c] = fmaf(h1[c], w1[c], acc[4 + c]); }
      }
    V::store(atile, (int64_t)p * CA + cg * 8, acc);
  }
  __syncthreads();

  // ---- phase 3: pointwise conv, D[n, pixel] = W[n,:] . tile[pixel,:] ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  if (y0 + 2 * wave >= h) return;                      // this wave's two tile rows are outside the map
  const int m = wave * 16 + r;
  const int y = y0 + (m >> 3), x = x0 + (m & 7);
  const bool pix_ok = y < h && x < w;
  const T* W = reinterpret_cast<const T*>(sg.wpw);
  const int64_t obase = (int64_t)b * sg.out_bstride + sg.out_off + ((int64_t)y * w + x) * sg.out_rowstride;
  for (int nt = 0; nt < sg.tilesN; nt++) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16) {
      for (int kk = 0; kk < C; kk += 32) {
        const int k = kk + 8 * g;
        u32x4 wa = (u32x4){0, 0, 0, 0}, xa = (u32x4){0, 0, 0, 0};
        if (k < C) {
          wa = *reinterpret_cast<const u32x4*>(W + (int64_t)(nt * 16 + r) * C + k);
          xa = *reinterpret_cast<const u32x4*>(atile + (int64_t)m * CA + k);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
      }
    } else {
      for (int kk = 0; kk < C; kk += 16) {
        const int k = kk + 4 * g;
        f32x4 wa = (f32x4){0.f, 0.f, 0.f, 0.f}, xa = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (k < C) {
          wa = *reinterpret_cast<const f32x4*>(W + (int64_t)(nt * 16 + r) * C + k);
          xa = *reinterpret_cast<const f32x4*>(atile + (int64_t)m * CA + k);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q], xa[q], acc, 0, 0, 0);
      }
    }
    const int n = nt * 16 + 4 * g;
    if (!pix_ok || n >= sg.N) continue;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(sg.bias + n);
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = apply_act(acc[q] + bias[q], sg.act);
    if (sg.out_f32) {
      float* o = reinterpret_cast<float*>(sg.out) + obase;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int nn = n + q;
        if (nn < sg.N) o[(nn / sg.col_kin) * sg.col_kout + nn % sg.col_kin + sg.col_off] = v[q];
      }
    } else {
      V::store4(sg.out, obase + n, v);     // N is a multiple of 8 for every non-header layer
    }
  }
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
  if (a.bf16) hipLaunchKernelGGL(sep_kernel<true>, grid, dim3(256), a.lds_bytes, s, a);
  else hipLaunchKernelGGL(sep_kernel<false>, grid, dim3(256), a.lds_bytes, s, a);
}
