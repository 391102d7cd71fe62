// k_sepstream.hip - the head layers' SeparableConvBlock as a STREAMING kernel on gfx950.
//
// Same arithmetic as k_sep.hip mode 1 (dw3x3 from an LDS halo -> MFMA 1x1 -> +bias, act -> LDS
// tile -> coalesced rows), for launches made of many independent single-source segments (a tower
// layer or the headers of all five heads: ~1800-4000 tiles of 8x8 cells).  There every workgroup
// was one tile = one chain of ~6 dependent global round trips with only two workgroups per CU to
// hide them.  Here a workgroup owns a RUN of consecutive tiles and software-pipelines them: the
// input vectors of tile i+1 (and its descriptor) are in flight while tile i goes through
// depthwise / MFMA / copy-out, and depthwise weights + bias are reloaded only when the run crosses
// into another segment.  ~256 workgroups per launch instead of ~2000.
#include <stdlib.h>

#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

#define SST_THREADS 512
#define SST_WAVES 8
#define SST_NP 4          // prefetched 8-channel vectors per lane (100 halo cells x C/8 <= 4 * 512)

template <bool BF16>
__global__ __launch_bounds__(SST_THREADS, 2) void sep_stream_kernel(SepArgs a) {
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
  constexpr int KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4, PAD = BF16 ? 8 : 4;
  constexpr int TS = 8, HS = 10;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ SepSeg seg_s[2];
  const int b = blockIdx.y;
  const int tpb = (a.total_tiles + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * tpb, hi = min(lo + tpb, a.total_tiles);
  if (lo >= hi) return;
  const int C = a.C, CG = C >> 3, CH = C + PAD;
  T* halo = reinterpret_cast<T*>(smem);
  T* atile = reinterpret_cast<T*>(smem + a.off_atile);
  float* wdw_s = reinterpret_cast<float*>(smem + a.off_wdw);
  float* bias_s = reinterpret_cast<float*>(smem + a.off_bias);
  float* otile_f = reinterpret_cast<float*>(smem);
  T* otile_t = reinterpret_cast<T*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int ksteps = (C + KSTEP - 1) / KSTEP;
  const int nitems = HS * HS * CG;

  auto load_seg = [&](int tile, int slot) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.segs + a.tile_seg[tile]);
    if (threadIdx.x < sizeof(SepSeg) / 4) reinterpret_cast<uint32_t*>(&seg_s[slot])[threadIdx.x] = src[threadIdx.x];
  };
  // the (single, SRC_SAME) input vectors of a tile's halo, zero outside the image; fp32 keeps its
  // second half in p1
  raw_t p0[SST_NP], p1[SST_NP];
  auto prefetch = [&](int tile, const SepSeg& sg) {
    const int t = tile - sg.tile_begin;
    const int y0 = (t / sg.tiles_x) * TS, x0 = (t % sg.tiles_x) * TS;
    const T* src = reinterpret_cast<const T*>(sg.src[0]) + (int64_t)b * sg.h * sg.w * C;
#pragma unroll
    for (int j = 0; j < SST_NP; j++) {
      const int item = threadIdx.x + j * SST_THREADS;
      p0[j] = raw_t{}; p1[j] = raw_t{};
      if (item < nitems) {
        const int pos = item / CG, cg = item % CG;
        const int y = y0 + pos / HS - 1, x = x0 + pos % HS - 1;
        if (y >= 0 && y < sg.h && x >= 0 && x < sg.w) {
          const T* q = src + ((int64_t)y * sg.w + x) * C + cg * 8;
          p0[j] = *reinterpret_cast<const raw_t*>(q);
          if (!BF16) p1[j] = *reinterpret_cast<const raw_t*>(q + 4);
        }
      }
    }
  };

  load_seg(lo, 0);
  __syncthreads();
  prefetch(lo, seg_s[0]);
  int loaded_seg = -1;

  for (int tile = lo; tile < hi; tile++) {
    const int slot = (tile - lo) & 1;
    const SepSeg& sg = seg_s[slot];
    const int h = sg.h, w = sg.w;
    const int t = tile - sg.tile_begin;
    const int y0 = (t / sg.tiles_x) * TS, x0 = (t % sg.tiles_x) * TS;
    const int rows_valid = min(TS, h - y0), cols_valid = min(TS, w - x0);
    const int mtv = (rows_valid + 1) >> 1;
    const int npairs = mtv * sg.tilesN;
    const int Nc = sg.N;
    const T* W = reinterpret_cast<const T*>(sg.wpw);
    // next tile's descriptor; this segment's depthwise weights + bias when the run enters it
    if (tile + 1 < hi) load_seg(tile + 1, slot ^ 1);
    const int this_seg = a.tile_seg[tile];
    if (this_seg != loaded_seg) {
      for (int i = threadIdx.x; i < 9 * C; i += SST_THREADS) wdw_s[i] = sg.wdw[i];
      for (int i = threadIdx.x; i < sg.tilesN * 16; i += SST_THREADS) bias_s[i] = sg.bias[i];
      loaded_seg = this_seg;
    }
    // first pointwise-weight fragments of this tile (used in phase 3)
    auto wload = [&](int it) -> raw_t {
      raw_t v = {};
      const int pair = wave + SST_WAVES * (it / ksteps), ks = it % ksteps;
      const int k = ks * KSTEP + KLANE * g;
      if (pair < npairs && k < C) v = *reinterpret_cast<const raw_t*>(W + (int64_t)((pair / mtv) * 16 + r) * C + k);
      return v;
    };
    raw_t wring[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wring[q] = wload(q);
    // phase 1: the prefetched vectors -> halo
#pragma unroll
    for (int j = 0; j < SST_NP; j++) {
      const int item = threadIdx.x + j * SST_THREADS;
      if (item < nitems) {
        raw_t* d = reinterpret_cast<raw_t*>(halo + (int64_t)(item / CG) * CH + (item % CG) * 8);
        d[0] = p0[j];
        if (!BF16) d[1] = p1[j];
      }
    }
    __syncthreads();
    if (tile + 1 < hi) prefetch(tile + 1, seg_s[slot ^ 1]);      // in flight during phases 2-4

    // phase 2: depthwise 3x3 -> operand tile
    for (int item = threadIdx.x; item < TS * TS * CG; item += SST_THREADS) {
      const int p = item / CG, cg = item % CG;
      const int py = p / TS, px = p % TS;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          float hv[8];
          V::load(halo, (int64_t)((py + ky) * HS + px + kx) * CH + cg * 8, hv);
          const f32x4* wp = reinterpret_cast<const f32x4*>(wdw_s + (ky * 3 + kx) * C + cg * 8);
          const f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
          for (int c = 0; c < 4; c++) { acc[c] = fmaf(hv[c], w0[c], acc[c]); acc[4 + c] = fmaf(hv[4 + c], w1[c], acc[4 + c]); }
        }
      V::store(atile, (int64_t)p * CH + cg * 8, acc);
    }
    __syncthreads();

    // phase 3: pointwise conv -> LDS output tile (over the dead halo)
    const int my_pairs = npairs > wave ? (npairs - wave + SST_WAVES - 1) / SST_WAVES : 0;
    const int my_items = my_pairs * ksteps;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto step = [&](int it, raw_t wfrag) {
      const int pair = wave + SST_WAVES * (it / ksteps), ks = it % ksteps;
      const int mt = pair % mtv, nt = pair / mtv;
      const int m = mt * 16 + r;
      const int k = ks * KSTEP + KLANE * g;
      raw_t xa = {};
      if (k < C) xa = *reinterpret_cast<const raw_t*>(atile + (int64_t)m * CH + k);
      if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[q], xa[q], acc, 0, 0, 0);
      }
      if (ks != ksteps - 1) return;
      const int n = nt * 16 + 4 * g;
      if (n < Nc) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + n);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = apply_act_t<BF16>(acc[q] + bias[q], sg.act);
        if (sg.out_f32) {
#pragma unroll
          for (int q = 0; q < 4; q++) if (n + q < Nc) otile_f[(int64_t)m * Nc + n + q] = v[q];
        } else {
          V::store4(otile_t, (int64_t)m * Nc + n, v);
        }
      }
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    for (int it = 0; it < my_items; it += 4) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (it + q < my_items) {
          const raw_t wf = wring[q];
          wring[q] = wload(it + q + 4);
          step(it + q, wf);
        }
      }
    }
    __syncthreads();

    // phase 4: coalesced copy-out
    if (sg.out_f32) {
      float* o = reinterpret_cast<float*>(sg.out) + (int64_t)b * sg.out_bstride + sg.out_off;
      const int npx = rows_valid * cols_valid;
      for (int pp = wave; pp < npx; pp += SST_WAVES) {
        const int py = pp / cols_valid, px = pp % cols_valid;
        float* orow = o + ((int64_t)(y0 + py) * w + x0 + px) * sg.out_rowstride;
        for (int c = lane; c < Nc; c += 64) {
          const int nn = c + sg.n_base;
          orow[(nn / sg.col_kin) * sg.col_kout + nn % sg.col_kin + sg.col_off] = otile_f[(int64_t)(py * TS + px) * Nc + c];
        }
      }
    } else {
      T* o = reinterpret_cast<T*>(sg.out) + (int64_t)b * sg.out_bstride + sg.out_off;
      const int vpp = Nc >> 3;
      for (int idx = threadIdx.x; idx < rows_valid * cols_valid * vpp; idx += SST_THREADS) {
        const int pix = idx / vpp, cv = idx % vpp;
        const int py = pix / cols_valid, px = pix % cols_valid;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(otile_t) + ((int64_t)(py * TS + px) * Nc + cv * 8) * sizeof(T);
        T* dst = o + ((int64_t)(y0 + py) * w + x0 + px) * sg.out_rowstride + cv * 8;
        *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
        if constexpr (!BF16) *reinterpret_cast<u32x4*>(dst + 4) = *reinterpret_cast<const u32x4*>(src + 16);
      }
    }
    __syncthreads();      // the output tile (= halo) and the operand tile are free for the next tile
  }
}

int sep_stream_prepare(void) {
  const void* fns[2] = {reinterpret_cast<const void*>(sep_stream_kernel<true>), reinterpret_cast<const void*>(sep_stream_kernel<false>)};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;
  return 0;
}

// usable when every segment is single-source SRC_SAME without fusion/pre-activation and 8x8 tiles
void launch_sep_stream(const SepArgs& a, hipStream_t s) {
  const int nblk = a.stream_blocks;
  dim3 grid(nblk, a.B);
  if (a.bf16) hipLaunchKernelGGL(sep_stream_kernel<true>, grid, dim3(SST_THREADS), a.lds_bytes, s, a);
  else hipLaunchKernelGGL(sep_stream_kernel<false>, grid, dim3(SST_THREADS), a.lds_bytes, s, a);
}
