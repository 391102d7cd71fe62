// k_sep.hip - fused SeparableConvBlock of the BiFPN nodes on gfx950 (and, with HEP_TOWER=0, of the head
// layers - by default those run on k_tower.hip):
//
//   [ weighted fusion of 2-3 maps (nearest x2 up / zero-padded 3x3/2 max-pool gathers) + swish ]
//     -> depthwise 3x3 SAME (no bias) -> pointwise 1x1 (+bias, BN folded) -> [swish | sigmoid]
//
// replaces, per BiFPN node, `swish(w0*a + w1*up(b) [+ w2*pool(c)])` + `SeparableConvBlock`
// (reference efficientdet/model.py:212-264, 42-52) and, per head layer,
// `conv(feat); bn(feat); swish(feat)` / the header conv + permute/view/cat
// (efficientdet/model.py:361-417; hmdegopose/model.py:55-90,127-156,191-228).
//
// One workgroup = one 8x8 output tile (4x4 where the fp32 tile of a wide layer would not fit in LDS) of one image of one "segment" (a node, or one (head, level,
// column-chunk) triple).  These maps are tiny (32x32 .. 2x2 per image), so the kernel is latency-bound:
// the design goal is the shortest dependent chain per workgroup and full-line stores.
//   phase 0  depthwise weights + bias -> LDS
//   phase 1  fused (+swish) 10x10 halo of the depthwise input -> LDS (dtype), zero outside the
//            map; all gather loads of an item are issued before the first is consumed
//   phase 2  depthwise 3x3 from LDS -> MFMA operand tile [64 pixels][C] in LDS
//   phase 3  1x1 conv as the transposed MFMA product of k_pw.hip (W fragment = A operand,
//            pixels = B operand); (m-tile, n-tile) pairs are dealt round-robin to the waves;
//            results go to an LDS output tile (over the dead halo)
//   phase 4  the output tile leaves as full coalesced rows.  Head outputs land directly at their
//            anchor offset in the [B, N_anchors, K] result (no permute / cat pass).
// All index arithmetic is shifts, masks and compile-time divisors (see the note in the kernel).
#include <stdlib.h>

#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

// profiling build (make trace): per-wave s_memrealtime stamps at the phase boundaries of the single-node launches on maps of side
// g_sep_trace_hw, read back with hep_dbg_sep_trace() (tools/trace_sep.py); compiled out of the product library
#ifdef HEP_TOWER_TRACE
__device__ unsigned long long* g_sep_trace = nullptr;
__device__ int g_sep_trace_hw = 0;
#define SSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); sstamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SSTAMP_NOWAIT(i) do { sstamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SSTAMP(i)
#define SSTAMP_NOWAIT(i)
#endif

// Launch modes:
//   0  one segment, descriptor in the kernel arguments            (BiFPN node on a 16x16 / 32x32 level)
//   1  many independent segments, tile -> segment table            (a tower layer / the headers of all heads)
//   2  a CHAIN of dependent single-tile segments run back to back by one workgroup per image
//      (consecutive BiFPN nodes on levels <= 8x8: p5_out -> p6_out -> p7_out -> p6_up' -> p5_up'):
//      the dependency is carried by a workgroup barrier instead of 4 kernel boundaries
// Modes 0/2 (<= 256 workgroups, pure latency): 16 waves per workgroup.  Mode 1 (~2000 workgroups,
// throughput): 8 waves so two workgroups share a CU.
// fp32 sessions: 8 waves in every mode - with 16 the 128-register budget spilled (20-100 bytes of scratch per lane, and scratch memory
// that the runtime keeps per queue: 2 MB per session created and destroyed)
#define SEP_THREADS_OF(mode, bf16) ((mode) != 1 && (bf16) ? 1024 : 512)
#ifndef SEP_M1_WAVES
#define SEP_M1_WAVES 4
#endif

// 8 channels as loaded (bf16: one 16-byte vector = 4 VGPRs; fp32: two)
template <bool BF16> struct Raw8;
template <> struct Raw8<true> {
  u32x4 v;
  __device__ __forceinline__ void zero() { v = (u32x4){0, 0, 0, 0}; }
  __device__ __forceinline__ void load(const void* base, int64_t idx) { v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(base) + idx); }
  __device__ __forceinline__ float get(int c) const { return (c & 1) ? __uint_as_float(v[c >> 1] & 0xffff0000u) : __uint_as_float(v[c >> 1] << 16); }
};
template <> struct Raw8<false> {
  f32x4 a, b;
  __device__ __forceinline__ void zero() { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }
  __device__ __forceinline__ void load(const void* base, int64_t idx) {
    const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx); a = p[0]; b = p[1];
  }
  __device__ __forceinline__ float get(int c) const { return c < 4 ? a[c] : b[c - 4]; }
};

// fused value of 8 channels at (y, x): sum_i fw_i * gather_i, optional swish.  Every load of the
// item is issued before the first one is consumed (one memory round trip per item in bf16; the
// fp32 parity mode walks the max-pool window row by row to stay inside the register budget).
// All loads are UNCONDITIONAL on in-range addresses (a load under a branch costs its own round trip): SAME and UP
// sources differ by a shift, a DOWN source is also read at (y, x) (in range: it is the larger map) and ignored,
// window taps outside the map read a clamped address and are replaced by the zero padding.  Offsets inside an image
// are 32-bit; the per-image base is uniform.
template <bool BF16>
__device__ __forceinline__ void gather_fuse(const SepSeg& sg, int b, int y, int x, int c0, float v[8]) {
  const int C = sg.C;
  constexpr int ROWS = BF16 ? 3 : 1;        // window rows in flight at once
  typedef typename std::conditional<BF16, bf16_t, float>::type T;
  Raw8<BF16> same[HEP_MAX_SRC];
  Raw8<BF16> win[ROWS * 3];
  // (the descriptor's arrays are only ever indexed by unrolled constants: a run-time index would
  //  push the whole by-value struct out of scalar registers into scratch memory)
  int down = -1, dsh = 1, dsw = 1, dpad = 0; const T* dsrc = reinterpret_cast<const T*>(sg.src[0]);
#pragma unroll
  for (int i = 0; i < HEP_MAX_SRC; i++) {
    const bool used = i < sg.nsrc;                                   // (uniform)
    const int sh = used ? sg.sh[i] : sg.sh[0], sw = used ? sg.sw[i] : sg.sw[0];
    const T* src = reinterpret_cast<const T*>(used ? sg.src[i] : sg.src[0]) + (int64_t)b * sh * sw * C;
    const int kind = used ? sg.kind[i] : SRC_SAME, sft = kind == SRC_UP ? 1 : 0;
    same[i].load(src, (uint32_t)(((y >> sft) * sw + (x >> sft)) * C + c0));
    if (used && kind == SRC_DOWN) { down = i; dsh = sh; dsw = sw; dpad = sg.pool_pad[i]; dsrc = src; }   // (at most one per node)
  }
  float pooled[8];
  if (down >= 0) {                                                  // (uniform)
#pragma unroll
    for (int r0 = 0; r0 < 3; r0 += ROWS) {
      bool in[ROWS * 3];
#pragma unroll
      for (int q = 0; q < ROWS * 3; q++) {
        const int iy = 2 * y - dpad + r0 + q / 3, ix = 2 * x - dpad + q % 3;
        in[q] = iy >= 0 && iy < dsh && ix >= 0 && ix < dsw;
        win[q].load(dsrc, (uint32_t)((min(max(iy, 0), dsh - 1) * dsw + min(max(ix, 0), dsw - 1)) * C + c0));
      }
#pragma unroll
      for (int c = 0; c < 8; c++) {
        float m = in[0] ? win[0].get(c) : 0.f;
#pragma unroll
        for (int q = 1; q < ROWS * 3; q++) m = fmaxf(m, in[q] ? win[q].get(c) : 0.f);
        pooled[c] = r0 == 0 ? m : fmaxf(pooled[c], m);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; c++) v[c] = 0.f;
#pragma unroll
  for (int i = 0; i < HEP_MAX_SRC; i++) {
    if (i >= sg.nsrc) break;
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = fmaf(sg.fw[i], i == down ? pooled[c] : same[i].get(c), v[c]);
  }
  if (sg.pre_act) {
    swish_n<BF16, 8>(v);
  }
}

// WL: the node's pointwise weights are staged in LDS (a.off_wpw; bf16 nodes wider than 64 channels, modes 0 / 2)
// W8: eight waves instead of sixteen for a single bf16 node of width <= 64 (round 6).  A 1024-thread workgroup at 112 VGPRs is alone on
// its CU and spends 2.9 of its 5.4 us waiting for the fused gather; two 512-thread workgroups - of this launch or of another stream's -
// share the CU.  Alone the launches get 0.3-1.0 us slower (6.1 / 6.8 / 7.4 -> 6.4 / 7.3 / 8.4 us), with four batches in flight
// 53.66k -> 55.60k frames/s (+3.6 %), one batch -0.8 %, bit-identical.  (Width 160 with staged weights: 5598 -> 5376 at phi 3: stays at sixteen.)
template <bool BF16, int MODE, bool WL = false, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : SEP_THREADS_OF(MODE, BF16), MODE == 1 ? SEP_M1_WAVES : (BF16 ? 4 : 2)) void sep_kernel(SepArgs a) {
  static_assert(!W8 || (BF16 && MODE == 0 && !WL), "eight-wave form: single bf16 nodes without staged weights");
  constexpr int SEP_THREADS = W8 ? 512 : SEP_THREADS_OF(MODE, BF16), SEP_WAVES = SEP_THREADS / 64;
  constexpr bool SINGLE = MODE == 0;
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
  constexpr int KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4, PAD = BF16 ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, a.lds_bytes);
#ifdef HEP_TOWER_TRACE
  unsigned long long sstamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  SSTAMP_NOWAIT(0);
#endif
  // the segment descriptor is never re-read from global memory: single-segment launches (BiFPN
  // nodes) take it from the kernel arguments (scalar loads), multi-segment launches (heads) copy
  // theirs into LDS once
  __shared__ SepSeg seg_s;
  for (int chain_i = 0; chain_i < (MODE == 2 ? a.nseg : 1); chain_i++) {
  if (!SINGLE) {
    const int si = MODE == 2 ? chain_i : a.tile_seg[blockIdx.x];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.segs + si);
    if (threadIdx.x < sizeof(SepSeg) / 4) reinterpret_cast<uint32_t*>(&seg_s)[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
  }
  const SepSeg& sg = SINGLE ? a.seg0 : seg_s;
  const int C = sg.C, CG = C >> 3, h = sg.h, w = sg.w, TS = sg.ts, HS = TS + 2;
  const int t = MODE == 2 ? 0 : blockIdx.x - sg.tile_begin;
  const int b = blockIdx.y;                 // grid = (tiles of one image over all segments, batch)
  const int tile_y = udiv_rcp(t, sg.tiles_x_rcp);
  const int y0 = tile_y * TS, x0 = (t - tile_y * sg.tiles_x) * TS;
  const int CH = C + PAD;                   // halo and operand-tile row pitch (elements)
  T* halo = reinterpret_cast<T*>(smem);
  T* atile = reinterpret_cast<T*>(smem + a.off_atile);
  float* wdw_s = reinterpret_cast<float*>(smem + a.off_wdw);
  float* bias_s = reinterpret_cast<float*>(smem + a.off_bias);

  // (index arithmetic below: channel groups and tile sides are split with shifts and masks, the halo
  //  side with a compile-time divisor per tile size - a run-time integer division costs ~30 VALU
  //  instructions and this kernel used to issue half a dozen of them per work item)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int rows_valid = min(TS, h - y0), cols_valid = min(TS, w - x0);
  const int mtv = TS == 16 ? rows_valid : (TS == 8 ? (rows_valid + 1) >> 1 : 1);   // 16-pixel m-tiles holding a valid pixel (TS 4: the whole tile)
  const int ksteps = (C + KSTEP - 1) / KSTEP;
  const T* W = reinterpret_cast<const T*>(sg.wpw);
  const int cgsh = CG <= 1 ? 0 : 32 - __builtin_clz(CG - 1);
  const int tssh = TS == 16 ? 4 : (TS == 8 ? 3 : 2);
  const int mtsh = mtv <= 1 ? 0 : 32 - __builtin_clz(mtv - 1);
  // ---- phase 0: everything that does not depend on the activations is REQUESTED here and parked in LDS once the
  //      gathers are in flight: depthwise weights, bias (16-byte vectors; were two dword loops = two round trips in
  //      front of the gathers) and the weight fragments of this wave's first (m-tile, n-tile) pair of phase 3 (were
  //      a global round trip between the depthwise conv and the MFMAs) ----
  const int nwv = (9 * C) >> 2, nbv = sg.tilesN * 4;
  const f32x4* wdw_g = reinterpret_cast<const f32x4*>(sg.wdw);
  const f32x4 wv0 = wdw_g[min((int)threadIdx.x, nwv - 1)];
  const f32x4 bv = reinterpret_cast<const f32x4*>(sg.bias)[min((int)threadIdx.x, nbv - 1)];
  constexpr int WPRE = 2;                                 // k-steps of prefetched weight fragments (all of them for C = 64 in bf16)
  raw_t wpre[WPRE];
  if constexpr (!WL) {
    const T* wrow = W + (int64_t)(min(wave >> mtsh, sg.tilesN - 1) * 16 + r) * C;
#pragma unroll
    for (int q = 0; q < WPRE; q++) {
      wpre[q] = *reinterpret_cast<const raw_t*>(wrow + min(q * KSTEP + KLANE * g, C - KLANE));   // (k >= C meets a zero activation fragment)
    }
  }
  auto park = [&]() {
    f32x4* wd = reinterpret_cast<f32x4*>(wdw_s);
    if ((int)threadIdx.x < nwv) wd[threadIdx.x] = wv0;
    for (int i = threadIdx.x + SEP_THREADS; i < nwv; i += SEP_THREADS) wd[i] = wdw_g[i];      // (C > 455 / 227 only)
    if ((int)threadIdx.x < nbv) reinterpret_cast<f32x4*>(bias_s)[threadIdx.x] = bv;
  };

  // ---- phase 1: fused (+swish) halo of the depthwise input, zero outside the image ----
  // (thread -> (position, channel group): an exact split by CG through a float reciprocal - with the power-of-two split a 160-channel
  //  node used 20 of every 32 lanes: four sweeps over its 100 halo positions instead of two)
  //  (single nodes only: in the multi-segment and chain instantiations the extra live values spilled)
  //  (measured and left out: two halo items in flight per lane for the two-source top-down nodes - one global round trip instead of
  //   two at widths above 64 - changed no launch by more than 0.5 us: sixteen waves per CU already overlap the sweeps)
  constexpr bool EXACT = MODE == 0;
  const float cg_rinv = __builtin_amdgcn_rcpf((float)CG);
  const int t_pos = EXACT ? udiv_f((int)threadIdx.x, CG, cg_rinv) : (int)(threadIdx.x >> cgsh);
  const int t_cg = EXACT ? (int)threadIdx.x - t_pos * CG : (int)(threadIdx.x & ((1 << cgsh) - 1));
  const int t_stride = EXACT ? udiv_f(SEP_THREADS, CG, cg_rinv) : (SEP_THREADS >> cgsh);
  {
    const int cg = t_pos < t_stride ? t_cg : CG;                      // (the last SEP_THREADS % CG threads idle)
    if (cg < CG)
      for (int pos = t_pos; pos < HS * HS; pos += t_stride) {
        const int hy = TS == 16 ? pos / 18 : (TS == 8 ? pos / 10 : pos / 6), hx = pos - hy * HS;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (y >= 0 && y < h && x >= 0 && x < w) gather_fuse<BF16>(sg, b, y, x, cg * 8, v);
        V::store(halo, (int64_t)pos * CH + cg * 8, v);
      }
  }
  SSTAMP(1);                            // gather done (halo stored)
  park();
  __syncthreads();
  SSTAMP_NOWAIT(2);
  // staged pointwise weights (a.off_wpw): requested here, under the depthwise phase, parked before the MFMA phase
  T* wpw_s = reinterpret_cast<T*>(smem + a.off_wpw);
  constexpr int NWST = WL ? 4 : 1;
  static_assert(!WL || SEP_THREADS == 1024, "sep_lds_layout admits up to 4 x 1024 staged weight vectors");
  raw_t wst[NWST];
  const int wvecs = sg.tilesN * 16 * CG;                                          // 16-byte vectors of [16 * tilesN rows][C] (rows >= C are the pack's zero padding)
  if constexpr (WL) {
#pragma unroll
    for (int j = 0; j < NWST; j++) wst[j] = reinterpret_cast<const raw_t*>(W)[min((int)threadIdx.x + j * SEP_THREADS, wvecs - 1)];
  }

  // ---- phase 2: depthwise 3x3 -> operand tile [TS*TS pixels][C] ----
  {
    const int cg = t_pos < t_stride ? t_cg : CG;
    if (cg < CG)
      for (int p = t_pos; p < TS * TS; p += t_stride) {
        const int py = p >> tssh, px = p & (TS - 1);
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
  }
  SSTAMP(3);                            // depthwise done
  if constexpr (WL) {
    const float cg_inv = __builtin_amdgcn_rcpf((float)CG);
#pragma unroll
    for (int j = 0; j < NWST; j++) {
      const int i = threadIdx.x + j * SEP_THREADS, row = (int)(((float)i + 0.5f) * cg_inv), v = i - row * CG;    // i < 4096, CG <= 20
      if (i < wvecs) *reinterpret_cast<raw_t*>(wpw_s + row * CH + v * 8) = wst[j];
    }
  }
  __syncthreads();      // halo is dead from here on: its LDS becomes the output tile
  SSTAMP_NOWAIT(4);                     // staged weights parked, barrier passed

  // ---- phase 3: pointwise conv, D[n, pixel] = W[n,:] . tile[pixel,:] -> LDS output tile ----
  const int Nc = sg.N;                                    // columns of this segment
  float* otile_f = reinterpret_cast<float*>(smem);        // [TS*TS][Nc] fp32 (head outputs)
  T* otile_t = reinterpret_cast<T*>(smem);                // [TS*TS][Nc] dtype (maps)
  // (m-tile, n-tile) pairs are dealt round-robin to the waves; pair -> (nt, mt) is a shift and a mask
  for (int pair = wave; pair < (sg.tilesN << mtsh); pair += SEP_WAVES) {
    const int nt = pair >> mtsh, mt = pair & ((1 << mtsh) - 1);
    if (mt >= mtv) continue;
    const int m = mt * 16 + r;
    const T* arow = atile + (int64_t)m * CH + KLANE * g;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The weight fragments of a pair come straight from global memory (L2): up to WGRP k-steps are requested together,
    // unconditionally on clamped addresses (k >= C meets a zero activation fragment).  Two at a time - the prefetch
    // depth of the prologue - made a 160-wide layer (5 k-steps) three dependent round trips per pair: at phi 3 @ 512 the
    // nodes on the 64x64 level spent about half of their 13 us per workgroup there.
    // (narrow layers - 2 k-steps at width 64 - keep groups of two: clamped extra loads are not free)
    auto kloop = [&](auto grp) {
      constexpr int WGRP = decltype(grp)::value;
      for (int ks0 = 0; ks0 < ksteps; ks0 += WGRP) {        // (a trailing partial group multiplies zeros)
        raw_t wfr[WGRP];
#pragma unroll
        for (int q = 0; q < WGRP; q++) {
          if constexpr (WL) wfr[q] = *reinterpret_cast<const raw_t*>(wpw_s + (nt * 16 + r) * CH + min((ks0 + q) * KSTEP + KLANE * g, C - KLANE));
          else if (!WL && q < WPRE && pair == wave && ks0 == 0) wfr[q] = wpre[q];       // (uniform) requested at kernel start
          else wfr[q] = *reinterpret_cast<const raw_t*>(W + (int64_t)(nt * 16 + r) * C + min((ks0 + q) * KSTEP + KLANE * g, C - KLANE));
        }
#pragma unroll
        for (int q = 0; q < WGRP; q++) {
          const int ks = ks0 + q;
          if (ks < ksteps) {                                  // (uniform)
            raw_t xa = {};
            if (ks * KSTEP + KLANE * g < C) xa = *reinterpret_cast<const raw_t*>(arow + ks * KSTEP);
            if constexpr (BF16) {
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[q]), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
            } else {
#pragma unroll
              for (int qq = 0; qq < 4; qq++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[q][qq], xa[qq], acc, 0, 0, 0);
            }
          }
        }
      }
    };
    if constexpr (BF16) {
      if (WL || ksteps > 2) kloop(std::integral_constant<int, 6>()); else kloop(std::integral_constant<int, 2>());
    } else {
      // fp32 sessions: two k-steps at a time as before (wider groups spill in this kernel's fp32 instantiations)
      const T* wrow = W + (int64_t)(nt * 16 + r) * C + KLANE * g;
      for (int ks0 = 0; ks0 < ksteps; ks0 += WPRE) {        // (a trailing partial group multiplies zeros)
#pragma unroll
        for (int q = 0; q < WPRE; q++) {
          const int ks = ks0 + q;
          raw_t wf = {}, xa = {};
          if (ks * KSTEP + KLANE * g < C) {
            xa = *reinterpret_cast<const raw_t*>(arow + ks * KSTEP);
            if (pair == wave && ks0 == 0) wf = wpre[q];       // (uniform) requested at kernel start
            else wf = *reinterpret_cast<const raw_t*>(wrow + ks * KSTEP);
          }
#pragma unroll
          for (int qq = 0; qq < 4; qq++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[qq], xa[qq], acc, 0, 0, 0);
        }
      }
    }
    const int n = nt * 16 + 4 * g;          // lane: 4 consecutive columns of pixel m
    if (n < Nc) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + n);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = acc[q] + bias[q];
      if (sg.act == ACT_SWISH) swish_n<BF16, 4>(v);                 // (uniform branches: BiFPN nodes have no activation behind the conv - the
      else if (sg.act == ACT_SIGMOID) sigmoid_n<BF16, 4>(v);        //  branch-free select computed both transcendental forms for nothing)
      if (sg.out_f32) {
#pragma unroll
        for (int q = 0; q < 4; q++) if (n + q < Nc) otile_f[(int64_t)m * Nc + n + q] = v[q];
      } else {
        V::store4(otile_t, (int64_t)m * Nc + n, v);     // Nc is a multiple of 8 for every map-producing layer
      }
    }
  }
  SSTAMP(5);                            // MFMA pairs done
  __syncthreads();
  SSTAMP_NOWAIT(6);

  // ---- phase 4: coalesced copy-out ----
  if (sg.out_f32) {
    // head result [B, N_anchors, K]: pixel p owns 9*K consecutive floats; this segment's Nc columns
    float* o = reinterpret_cast<float*>(sg.out) + (int64_t)b * sg.out_bstride + sg.out_off;
    const int npx = rows_valid * cols_valid;
    for (int pp = wave; pp < npx; pp += SEP_WAVES) {
      const int py = pp / cols_valid, px = pp % cols_valid;
      const int m = py * TS + px;
      float* orow = o + ((int64_t)(y0 + py) * w + x0 + px) * sg.out_rowstride;
      for (int c = lane; c < Nc; c += 64) {
        const int nn = c + sg.n_base;
        orow[(nn / sg.col_kin) * sg.col_kout + nn % sg.col_kin + sg.col_off] = otile_f[(int64_t)m * Nc + c];
      }
    }
  } else {
    // NHWC map (possibly a column chunk [out_off, out_off+Nc) of a wider map): consecutive lanes
    // write consecutive 16-byte vectors of a pixel, then the next pixel of the tile row
    T* o = reinterpret_cast<T*>(sg.out) + (int64_t)b * sg.out_bstride + sg.out_off;
    const int vpp = Nc >> 3;
    const int vsh = vpp <= 1 ? 0 : 32 - __builtin_clz(vpp - 1);
    const int cv = threadIdx.x & ((1 << vsh) - 1);
    if (cv < vpp)
      for (int pix = threadIdx.x >> vsh; pix < TS * TS; pix += SEP_THREADS >> vsh) {
        const int py = pix >> tssh, px = pix & (TS - 1);
        if (py >= rows_valid || px >= cols_valid) continue;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(otile_t) + ((int64_t)pix * Nc + cv * 8) * sizeof(T);
        T* dst = o + ((int64_t)(y0 + py) * w + x0 + px) * sg.out_rowstride + cv * 8;
        *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
        if constexpr (!BF16) *reinterpret_cast<u32x4*>(dst + 4) = *reinterpret_cast<const u32x4*>(src + 16);
      }
  }
  // chain mode: this node's stores must have landed (and every LDS reader be done) before the next
  // node of the chain gathers them - same workgroup, same CU, so a barrier after vmcnt(0) suffices
  if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#ifdef HEP_TOWER_TRACE
  if (MODE == 0) {
    SSTAMP(7);                          // stores acknowledged
    if (g_sep_trace && (threadIdx.x & 63) == 0 && sg.h == g_sep_trace_hw) {
      unsigned long long* o = g_sep_trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * SEP_WAVES + (threadIdx.x >> 6)) * 8;
      for (int i = 0; i < 8; i++) o[i] = sstamps[i];
    }
  }
#endif
  }   // chain loop
}

#ifdef HEP_TOWER_TRACE
// profiling build: stamps of the LAST single-node launch on a map of side `hw`, [workgroups * waves][8]; hw = 0 turns the stamps off
extern "C" int hep_dbg_sep_trace(unsigned long long* host, int max_waves, int hw) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)1 << 20;
  if (!buf) { if (hipMalloc((void**)&buf, cap * 8) != hipSuccess) return -1; }
  if (!host) hipMemset(buf, 0, cap * 8);
  unsigned long long* p = hw ? buf : nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_sep_trace), &p, sizeof p);
  hipMemcpyToSymbol(HIP_SYMBOL(g_sep_trace_hw), &hw, sizeof hw);
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)max_waves * 64, hipMemcpyDeviceToHost); }
  return (int)(cap / 8);
}
#endif

void sep_lds_layout(int C, int bf16, int ts, int max_cols_f32, int max_cols_map, SepArgs* a, int stage_w) {
  const size_t es = bf16 ? 2 : 4, pad = bf16 ? 8 : 4;
  const size_t hs = ts + 2, px = (size_t)ts * ts;
  size_t region = hs * hs * (C + pad) * es;                              // halo ...
  region = std::max(region, px * (size_t)max_cols_f32 * 4);              // ... or the fp32 output tile
  region = std::max(region, px * (size_t)max_cols_map * es);             // ... or the dtype output tile
  region = (region + 15) & ~(size_t)15;
  a->off_atile = region;
  a->off_wdw = a->off_atile + px * (C + pad) * es;
  a->off_bias = a->off_wdw + (size_t)9 * C * 4;
  a->lds_bytes = a->off_bias + (size_t)SEP_MAX_TILES_MAP * 16 * 4;
  // BiFPN nodes wider than 64 channels (bf16, 8x8 tiles): the pointwise weights [C][C + pad] next to the tiles when they fit -
  // requested behind the gather barrier, parked before the MFMA phase.  (At width 64 every wave's only (m-tile, n-tile)
  // pair has its fragments prefetched at kernel start; at width 160 a wave runs 2.5 pairs, each one a round trip to L2
  // that nothing overlapped: 15-28 us per node at phi 3 @ 512 for 1.5-24 MB.)
  a->off_wpw = 0;
  const size_t wrows = (size_t)((C + 15) / 16) * 16, wbytes = wrows * (C + pad) * es;      // whole n-tiles: widths like 88 end in a half-used one
  if (stage_w && bf16 && C > 64 && (ts == 8 || ts == 4) && wrows * (C / 8) <= 4 * 1024 && a->lds_bytes + wbytes <= 159 * 1024) { a->off_wpw = (a->lds_bytes + 15) & ~(size_t)15; a->lds_bytes = a->off_wpw + wbytes; }
}

int sep_prepare(void) {
  const void* fns[9] = {reinterpret_cast<const void*>(sep_kernel<true, 0, false, true>),
                        reinterpret_cast<const void*>(sep_kernel<true, 0>), reinterpret_cast<const void*>(sep_kernel<true, 1>),
                        reinterpret_cast<const void*>(sep_kernel<true, 2>), reinterpret_cast<const void*>(sep_kernel<false, 0>),
                        reinterpret_cast<const void*>(sep_kernel<false, 1>), reinterpret_cast<const void*>(sep_kernel<false, 2>),
                        reinterpret_cast<const void*>(sep_kernel<true, 0, true>), reinterpret_cast<const void*>(sep_kernel<true, 2, true>)};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;
  return 0;
}

// single bf16 node of width <= 64 without staged weights: the eight-wave instantiation (launch_sep and hep_kernel_symbol agree through this)
int sep_w8(const SepArgs& a) { return a.bf16 && !a.chain && a.nseg == 1 && !a.off_wpw && a.C <= 64; }

void launch_sep(const SepArgs& a_, hipStream_t s) {
  const SepArgs& a = a_;
  const int mode = a.chain ? 2 : (a.nseg == 1 ? 0 : 1);
  dim3 grid(mode == 2 ? 1 : a.total_tiles, a.B);
  const dim3 block(SEP_THREADS_OF(mode, a.bf16));
  if (a.bf16) {
    // (staged weights: every segment of the launch is a map-to-map node of the full width - the planner only sets off_wpw then)
    if (mode == 0 && a.off_wpw) hipLaunchKernelGGL((sep_kernel<true, 0, true>), grid, block, a.lds_bytes, s, a);
    else if (mode == 2 && a.off_wpw) hipLaunchKernelGGL((sep_kernel<true, 2, true>), grid, block, a.lds_bytes, s, a);
    else if (mode == 0 && sep_w8(a)) hipLaunchKernelGGL((sep_kernel<true, 0, false, true>), grid, dim3(512), a.lds_bytes, s, a);
    else if (mode == 0) hipLaunchKernelGGL((sep_kernel<true, 0>), grid, block, a.lds_bytes, s, a);
    else if (mode == 1) hipLaunchKernelGGL((sep_kernel<true, 1>), grid, block, a.lds_bytes, s, a);
    else hipLaunchKernelGGL((sep_kernel<true, 2>), grid, block, a.lds_bytes, s, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL((sep_kernel<false, 0>), grid, block, a.lds_bytes, s, a);
    else if (mode == 1) hipLaunchKernelGGL((sep_kernel<false, 1>), grid, block, a.lds_bytes, s, a);
    else hipLaunchKernelGGL((sep_kernel<false, 2>), grid, block, a.lds_bytes, s, a);
  }
}
