// k_head.hip - a whole head (D tower layers + header[s]) of one (net, pyramid level) for one
// TSxTS tile of anchor cells in ONE gfx950 workgroup:
//
//   for i in 0..D-1:  feat = swish(bn_list[level][i](conv_list[i](feat)))      (SeparableConvBlock)
//   out = header(feat)  ->  [B, N_anchors, K] at the level's anchor offset
//
// replaces Regressor / Classifier / RotationNet / TranslationNet / HandNet.forward (reference
// efficientdet/model.py:361-417; hmdegopose/model.py:55-90,127-156,191-228): D+1 launches of
// the per-layer kernel (k_sep.hip) and 3 round trips of every tower activation through memory
// become one launch whose intermediate maps never leave LDS.
//
// Halo recompute: a tile of TS x TS outputs needs its input on (TS + 2(D+1))^2 cells; layer i is
// evaluated on the region that still matters ((TS + 2(D-i))^2 cells) and cells outside the
// image are forced to ZERO after every layer, because every SeparableConv zero-pads ITS OWN
// input at the image border (TF-SAME).  The redundant arithmetic (about 2.7x on the 64->64 tower
// layers at TS=8, D=3) is cheap next to D extra kernel boundaries on a latency-bound path.
//
// Per layer: depthwise 3x3 from the LDS map -> MFMA operand tile -> 1x1 conv as the transposed
// MFMA product of k_pw.hip (the whole 64x64 weight sits in 8 fragment registers per lane) ->
// +bias, swish, mask -> the other LDS map.  Header: same, with the weight fragments streamed
// through a 4-deep register ring, results staged in LDS and written as coalesced rows.
#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

#define HEAD_THREADS 1024
#define HEAD_WAVES 16

template <bool BF16>
__global__ __launch_bounds__(HEAD_THREADS) void head_kernel(HeadArgs a) {
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
  constexpr int KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4, PAD = BF16 ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ HeadSeg sg;
  {
    const int si = a.tile_seg[blockIdx.x];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.segs + si);
    for (int i = threadIdx.x; i < (int)(sizeof(HeadSeg) / 4); i += HEAD_THREADS) reinterpret_cast<uint32_t*>(&sg)[i] = src[i];
    __syncthreads();
  }
  const int C = a.C, CG = C >> 3, CP = C + PAD, D = a.depth, TS = a.ts;
  const int h = sg.h, w = sg.w;
  const int t = blockIdx.x - sg.tile_begin, b = blockIdx.y;
  const int y0 = (t / sg.tiles_x) * TS, x0 = (t % sg.tiles_x) * TS;
  const int R = TS + 2 * (D + 1);                                  // side of the input region
  T* buf0 = reinterpret_cast<T*>(smem);                             // [R*R][CP]
  T* buf1 = reinterpret_cast<T*>(smem + a.off_buf1);                // [R*R][CP]
  T* atile = reinterpret_cast<T*>(smem + a.off_atile);              // [16*ceil((R-2)^2/16)][CP]
  float* wdw_s = reinterpret_cast<float*>(smem + a.off_wdw);        // [9][C] of the current layer
  float* bias_s = reinterpret_cast<float*>(smem + a.off_bias);      // [max(C, header chunk)]
  float* otile = reinterpret_cast<float*>(smem);                    // header output tile [TS*TS][chunk] over buf0
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int ksteps = (C + KSTEP - 1) / KSTEP;

  // ---- input region of the level's feature map -> buf0 (zero outside the image) ----
  {
    const T* feat = reinterpret_cast<const T*>(sg.feat) + (int64_t)b * h * w * C;
    const int oy = y0 - (D + 1), ox = x0 - (D + 1);
    for (int item = threadIdx.x; item < R * R * CG; item += HEAD_THREADS) {
      const int pos = item / CG, cg = item % CG;
      const int y = oy + pos / R, x = ox + pos % R;
      raw_t v0 = {}, v1 = {};
      if (y >= 0 && y < h && x >= 0 && x < w) {
        const T* src = feat + ((int64_t)y * w + x) * C + cg * 8;
        v0 = *reinterpret_cast<const raw_t*>(src);
        if (!BF16) v1 = *reinterpret_cast<const raw_t*>(src + 4);
      }
      raw_t* d = reinterpret_cast<raw_t*>(buf0 + (int64_t)pos * CP + cg * 8);
      d[0] = v0;
      if (!BF16) d[1] = v1;
    }
  }

  // one separable layer on a square region: in [rin x rin] -> out [(rin-2) x (rin-2)], N == C
  T* bin = buf0; T* bout = buf1;
  for (int L = 0; L < D; L++) {
    const int rin = R - 2 * L, rout = rin - 2, npx = rout * rout;
    const int halo = D - L;                                       // output region = tile + halo on each side
    // this layer's pointwise weight: all fragments of the C x C matrix this wave can need
    const T* W = reinterpret_cast<const T*>(sg.wpw[L]);
    // depthwise weights + bias of the layer -> LDS (previous layer's readers are past the barrier below)
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * C; i += HEAD_THREADS) wdw_s[i] = sg.wdw[L][i];
    for (int i = threadIdx.x; i < C; i += HEAD_THREADS) bias_s[i] = sg.bias[L][i];
    __syncthreads();
    // depthwise 3x3 -> atile
    for (int item = threadIdx.x; item < npx * CG; item += HEAD_THREADS) {
      const int p = item / CG, cg = item % CG;
      const int py = p / rout, px = p % rout;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          float hv[8];
          V::load(bin, (int64_t)((py + ky) * rin + px + kx) * CP + cg * 8, hv);
          const f32x4* wp = reinterpret_cast<const f32x4*>(wdw_s + (ky * 3 + kx) * C + cg * 8);
          const f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
          for (int c = 0; c < 4; c++) { acc[c] = fmaf(hv[c], w0[c], acc[c]); acc[4 + c] = fmaf(hv[4 + c], w1[c], acc[4 + c]); }
        }
      V::store(atile, (int64_t)p * CP + cg * 8, acc);
    }
    __syncthreads();
    // pointwise C -> C, + bias, swish, zero outside the image -> bout
    const int mtiles = (npx + 15) / 16, ntiles = C >> 4, npairs = mtiles * ntiles;
    for (int pair = wave; pair < npairs; pair += HEAD_WAVES) {
      const int mt = pair % mtiles, nt = pair / mtiles;
      const int m = mt * 16 + r;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < ksteps; ks++) {
        const int k = ks * KSTEP + KLANE * g;
        raw_t wf = {}, xa = {};
        if (k < C) {
          wf = *reinterpret_cast<const raw_t*>(W + (int64_t)(nt * 16 + r) * C + k);
          if (m < npx) xa = *reinterpret_cast<const raw_t*>(atile + (int64_t)m * CP + k);
        }
        if constexpr (BF16) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xa[q], acc, 0, 0, 0);
        }
      }
      if (m < npx) {
        const int y = y0 - halo + m / rout, x = x0 - halo + m % rout;
        const bool inside = y >= 0 && y < h && x >= 0 && x < w;
        const int n = nt * 16 + 4 * g;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + n);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = inside ? swish_t<BF16>(acc[q] + bias[q]) : 0.f;
        V::store4(bout, (int64_t)m * CP + n, v);
      }
    }
    T* tmp = bin; bin = bout; bout = tmp;
  }

  // ---- header(s): input bin [(TS+2)^2], output TS x TS cells x N columns ----
  const int rin = TS + 2, npx = TS * TS;
  const int rows_valid = min(TS, h - y0), cols_valid = min(TS, w - x0);
  for (int hd = 0; hd < sg.nheaders; hd++) {
    const HeadOut& ho = sg.hdr[hd];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * C; i += HEAD_THREADS) wdw_s[i] = ho.wdw[i];
    __syncthreads();
    for (int item = threadIdx.x; item < npx * CG; item += HEAD_THREADS) {
      const int p = item / CG, cg = item % CG;
      const int py = p / TS, px = p % TS;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          float hv[8];
          V::load(bin, (int64_t)((py + ky) * rin + px + kx) * CP + cg * 8, hv);
          const f32x4* wp = reinterpret_cast<const f32x4*>(wdw_s + (ky * 3 + kx) * C + cg * 8);
          const f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
          for (int c = 0; c < 4; c++) { acc[c] = fmaf(hv[c], w0[c], acc[c]); acc[4 + c] = fmaf(hv[4 + c], w1[c], acc[4 + c]); }
        }
      V::store(atile, (int64_t)p * CP + cg * 8, acc);
    }
    // column chunks: MFMA -> fp32 tile in LDS (over the tower maps, dead by now except `bin`,
    // which lives in the OTHER buffer when D is odd... so the tile goes over `bout`) -> coalesced rows
    float* ot = reinterpret_cast<float*>(bout);
    const T* W = reinterpret_cast<const T*>(ho.wpw);
    const int mtiles = (npx + 15) / 16;
    for (int n0 = 0; n0 < ho.N; n0 += a.chunk) {
      const int nc = min(a.chunk, ho.N - n0), ntiles = (nc + 15) / 16, npairs = mtiles * ntiles;
      __syncthreads();                                     // atile ready / previous chunk's tile fully copied out
      for (int i = threadIdx.x; i < ntiles * 16; i += HEAD_THREADS) bias_s[i] = ho.bias[n0 + i];
      __syncthreads();
      for (int pair = wave; pair < npairs; pair += HEAD_WAVES) {
        const int mt = pair % mtiles, nt = pair / mtiles;
        const int m = mt * 16 + r;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < ksteps; ks++) {
          const int k = ks * KSTEP + KLANE * g;
          raw_t wf = {}, xa = {};
          if (k < C) {
            wf = *reinterpret_cast<const raw_t*>(W + (int64_t)(n0 + nt * 16 + r) * C + k);
            if (m < npx) xa = *reinterpret_cast<const raw_t*>(atile + (int64_t)m * CP + k);
          }
          if constexpr (BF16) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xa[q], acc, 0, 0, 0);
          }
        }
        const int n = nt * 16 + 4 * g;
        if (m < npx && n < nc) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
          for (int q = 0; q < 4; q++) if (n + q < nc) ot[(int64_t)m * a.chunk + n + q] = apply_act_t<BF16>(acc[q] + bias[q], ho.act);
        }
      }
      __syncthreads();
      // copy-out: cell p owns 9*K consecutive floats of [B, N_anchors, K]; this chunk's columns
      float* o = ho.out + (int64_t)b * ho.out_bstride + sg.out_cell0 * ho.out_rowstride;
      const int nvalid = rows_valid * cols_valid;
      for (int pp = wave; pp < nvalid; pp += HEAD_WAVES) {
        const int py = pp / cols_valid, px = pp % cols_valid;
        const int m = py * TS + px;
        float* orow = o + ((int64_t)(y0 + py) * w + x0 + px) * ho.out_rowstride;
        for (int c = lane; c < nc; c += 64) {
          const int nn = n0 + c;
          orow[(nn / ho.col_kin) * ho.col_kout + nn % ho.col_kin + ho.col_off] = ot[(int64_t)m * a.chunk + c];
        }
      }
    }
  }
  (void)otile;
}

void head_lds_layout(int C, int depth, int ts, int bf16, int chunk, HeadArgs* a) {
  const size_t es = bf16 ? 2 : 4, pad = bf16 ? 8 : 4;
  const size_t R = ts + 2 * (depth + 1);
  size_t buf = R * R * (C + pad) * es;
  buf = std::max(buf, (size_t)ts * ts * chunk * 4);                 // header output tile lives in a tower buffer
  buf = (buf + 15) & ~(size_t)15;
  const size_t rows = (((R - 2) * (R - 2) + 15) / 16) * 16;
  a->off_buf1 = buf;
  a->off_atile = 2 * buf;
  a->off_wdw = a->off_atile + ((rows * (C + pad) * es + 15) & ~(size_t)15);
  a->off_bias = a->off_wdw + (size_t)9 * C * 4;
  a->lds_bytes = a->off_bias + (size_t)std::max(C, ((chunk + 15) / 16) * 16) * 4;
}

int head_prepare(void) {
  const void* fns[2] = {reinterpret_cast<const void*>(head_kernel<true>), reinterpret_cast<const void*>(head_kernel<false>)};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;
  return 0;
}

void launch_head(const HeadArgs& a, hipStream_t s) {
  dim3 grid(a.total_tiles, a.B);
  if (a.bf16) hipLaunchKernelGGL(head_kernel<true>, grid, dim3(HEAD_THREADS), a.lds_bytes, s, a);
  else hipLaunchKernelGGL(head_kernel<false>, grid, dim3(HEAD_THREADS), a.lds_bytes, s, a);
}
