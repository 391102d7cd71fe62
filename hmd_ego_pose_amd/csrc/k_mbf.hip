// k_mbf.hip - the front half of an MBConv block as ONE gfx950 kernel:
//
//   expand 1x1 (+BN0, swish)  ->  depthwise k x k stride s, TF-SAME (+BN1, swish)  ->  SE partial sums
//
// replaces `_expand_conv,_bn0,_swish,_depthwise_conv,_bn1,_swish` and the spatial half of
// `adaptive_avg_pool2d` (reference efficientnet/model.py:76-89).  The 6x expanded tensor - the
// largest activation of every block - never exists in HBM: it is produced by MFMA straight into
// LDS for one (8x8 output tile + halo) x (chunk of CC expanded channels) and consumed from LDS by
// the depthwise taps ("depthwise staged in LDS").  Every input element is fetched from global
// memory once per channel chunk, with 16-byte loads of contiguous channel runs.
//
//   phase A  input tile (with halo, zero outside the image) -> LDS [PIN pixels][K]
//   phase B  expand: D[n, pixel] = We[n,:] . tile[pixel,:] (the transposed MFMA product of
//            k_pw.hip; weight fragments run 4 deep ahead in a register ring) -> +bias, swish ->
//            LDS [PIN][CC]; only the tile pixels inside the image are expanded, the others are the
//            depthwise conv's padding of the *activated* map (reference utils_extra.py:33-44): zeros
//   phase C  depthwise taps from LDS (weights in LDS) -> +bias, swish -> global, 16 bytes per lane;
//            per-lane channel sums for squeeze-excite
//   phase D  channel sums of the tile (fixed order) -> their partial products with the squeeze-excite
//            reduce weights -> hpart[b][workgroup][j]   (no atomics; the project GEMM finishes the SE)
// Blocks without an expand conv (first block of the net) skip phase B: the input tile IS the
// depthwise input.
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"
#include "se_finish.h"

// Tile side TS: 8 (512 threads, two workgroups per CU) for the stride-2 layers and the 8x8 maps, 16 (1024
// threads) for the stride-1 layers on 16x16 / 32x32 maps: there an 8x8 tile re-expands its k x k halo (2.25x the
// pixels for k = 5) and every one of the 4x more workgroups pays the fixed staging / drain latency - the 16x16
// tile is the whole 16x16 map (no halo work at all) and a quarter of the 32x32 one.

// Optional per-wave timeline (make EXTRA=-DHEP_MBF_TRACE): s_memrealtime (100 MHz) stamps at the phase
// boundaries of the LAST mbf launch, read back with hep_dbg_mbf_trace() - profiling builds only.
#ifdef HEP_MBF_TRACE
__device__ unsigned long long* g_mbf_trace = nullptr;
#ifdef HEP_MBF_TRACE_A      // sub-steps of phase A instead of the phase boundaries: 1 kernargs in, 2 loads issued, 3 first data, 4 all parked, 5 barrier
#define MSTAMP(i) do { if ((i) == 0 || (i) == 6) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define XSTAMP(i) do { stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MSTAMP(i) do { stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define XSTAMP(i)
#endif
#else
#define MSTAMP(i)
#define XSTAMP(i)
#endif


// F8: e4m3 operands in the expand MFMA (fp8 sessions).  A template parameter: as a run-time flag both operand paths sat in
// the expand loop of every bf16 session and kept it from unrolling (340 instructions per 32-pixel x 16-channel item).
// MP: the expand conv runs in several passes over slices of its K input channels (a.kp channels per pass): per pass the slice of
// the input tile and of the weight chunk is staged in LDS and every wave adds its products to accumulators that stay in
// registers.  fp32 sessions: a whole-K input tile next to the expanded tile does not fit LDS for the 16x16 tile (blocks 9, 10
// fell back to 8x8 tiles = 704 workgroups = three rounds) or leaves one workgroup per CU where 288 want to be resident (the 8x8
// maps: two rounds).  With K in slices the fp32 fronts get the workgroup counts of the bf16 plan.
// MP = 1: the slices are fetched pass by pass (the next one under the running pass's MFMAs: a pass is then about one memory round
// trip, ~2 us); MP = 2: the WHOLE input tile and weight chunk are requested at kernel start into registers (up to six 8-channel
// vectors per lane: one round trip for the launch, as in the single-pass form) and only parked slice by slice - chosen whenever
// the vectors fit (mbf_mp_fits).
template <bool BF16, int KS, int S, int TS, bool F8, int MP>
__global__ __launch_bounds__(TS == 16 ? 1024 : 512) void mbf_kernel(MbfArgs a) {
  constexpr int MBF_THREADS = TS == 16 ? 1024 : 512, MBF_WAVES = MBF_THREADS / 64;
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
  constexpr int KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4, PAD = BF16 ? 8 : 4;
  constexpr int PW = (TS - 1) * S + KS;              // input tile side
  constexpr int PIN = PW * PW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, a.lds_bytes);
#ifdef HEP_MBF_TRACE
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MSTAMP(0);
#endif
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: whatever is derived from it alone runs on the scalar unit)
  const int r = lane & 15, g = lane >> 4;
  // The prologue below runs in every wave of a latency-bound workgroup: its instruction count is kernel time
  // (measured with the phase-A stamps: 1250 instructions = 2.6 us before the last staging load was issued).  So every
  // uniform quotient arrives as a host-made reciprocal, the shifts come from clz, addresses are 32-bit offsets from a
  // uniform base and the staging loads are unconditional on a clamped address.
  const int chunks = a.chunks;
  // XCD-aware order: the workgroups of one tile (its channel chunks) all stage the same input pixels; as consecutive
  // LOGICAL blocks they share an XCD and its L2 instead of fetching the tile into up to eight of them
  const int logical = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
  const int b = udiv_rcp(logical, a.gx_rcp), bxl = logical - b * (int)gridDim.x;
  const int tile = udiv_rcp(bxl, a.chunks_rcp), chunk = bxl - tile * chunks;
  const int tile_y = udiv_rcp(tile, a.tiles_x_rcp);
  const int oy0 = tile_y * TS, ox0 = (tile - tile_y * a.tiles_x) * TS;
  const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;     // input coords of tile pixel (0,0)
  const int c0 = chunk * a.CC;                                     // first expanded channel of this block
  const int cc = min(a.CC, a.Cexp - c0);                           // channels of this block (multiple of 8)
  const int K = a.Cin;
  const int KP = (MP ? a.kp : K) + PAD, EP = a.CC + PAD;           // LDS row pitches (elements)
  T* a_s = reinterpret_cast<T*>(smem);                             // [PIN][KP]   input tile
  T* e_s = reinterpret_cast<T*>(smem + a.off_e);                   // [PIN][EP]   expanded (activated) tile
  T* w_s = reinterpret_cast<T*>(smem + a.off_we);                  // [CC][KP]    expand-weight chunk
  float* wdw_s = reinterpret_cast<float*>(smem + a.off_w);         // [KS*KS][CC]
  float* be_s = wdw_s + KS * KS * a.CC;                            // [CC] expand bias
  float* bdw_s = be_s + a.CC;                                      // [CC] depthwise bias
  const int K16 = (K + 15) & ~15, KP8 = K16 + 16;                  // fp8 weight rows: bytes in memory / in LDS
  unsigned char* w8_s = smem + a.off_we;

  // ---- phase A: everything this workgroup needs, global -> LDS, all loads issued in batches of 8
  //      before the first LDS store (one memory round trip per batch): depthwise weights + biases,
  //      the expand-weight chunk [cc][K], and the input tile (zero outside the image) ----
  const int ntiles = (cc + 15) / 16;
  const int ksteps = (K + KSTEP - 1) / KSTEP;
  const int r0 = max(0, -iy0), r1 = min(PW, a.H - iy0), q0 = max(0, -ix0), q1 = min(PW, a.W - ix0);   // inside rectangle
  const int wi = q1 - q0, n_in = wi * (r1 - r0);
  // inside pixel m -> (row, column) of the rectangle: m < 2^10 and wi <= 20, so (m + 0.5) / wi stays 0.5 / wi away from
  // an integer and the 1-ulp reciprocal cannot move the truncation
  const float wi_inv = __builtin_amdgcn_rcpf((float)wi);
  auto row_of = [&](int m) { return (int)(((float)m + 0.5f) * wi_inv); };
  // depthwise weights + biases: 16-byte vectors; their loads are issued here, the input-tile / expand-weight
  // loads right behind them, and only then are they parked in LDS - one memory round trip for all of
  // phase A (cc and c0 are multiples of 8)
  constexpr int NDW = (KS * KS * 128 / 4 + MBF_THREADS - 1) / MBF_THREADS;       // CC <= 128
  const int cv = cc >> 2;                                                        // float4 vectors per tap
  const int cvsh = cv <= 2 ? 1 : 32 - __builtin_clz(cv - 1);
  f32x4 wv[NDW];
#ifdef HEP_MBF_TRACE_A
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  XSTAMP(1);
  const float* wdw_g = a.wdw + c0;
#pragma unroll
  for (int j = 0; j < NDW; j++) {
    const int i = threadIdx.x + j * MBF_THREADS, tap = i >> cvsh, c4 = i & ((1 << cvsh) - 1);
    const bool ok = tap < KS * KS && c4 < cv;
    wv[j] = *reinterpret_cast<const f32x4*>(wdw_g + (ok ? (uint32_t)(tap * a.Cexp + c4 * 4) : 0u));
  }
  const int bi = threadIdx.x;                        // threads [0, cv): depthwise bias, [64, 64 + cv): expand bias
  const bool b_dw = bi < cv, b_ex = a.has_expand && bi >= 64 && bi - 64 < cv;
  const f32x4 bv = *reinterpret_cast<const f32x4*>((b_ex ? a.be : a.bdw) + c0 + (b_dw || b_ex ? (uint32_t)(bi & 63) * 4 : 0u));
  auto park_weights = [&]() {
#pragma unroll
    for (int j = 0; j < NDW; j++) {
      const int i = threadIdx.x + j * MBF_THREADS, tap = i >> cvsh, c4 = i & ((1 << cvsh) - 1);
      if (tap < KS * KS && c4 < cv) *reinterpret_cast<f32x4*>(wdw_s + tap * a.CC + c4 * 4) = wv[j];
    }
    if (b_dw) *reinterpret_cast<f32x4*>(bdw_s + bi * 4) = bv;
    else if (b_ex) *reinterpret_cast<f32x4*>(be_s + (bi - 64) * 4) = bv;
  };
  {
    constexpr int NB = BF16 ? 8 : 4;                               // 16-byte vectors (bf16) / 32-byte pairs (fp32) in flight per lane
    constexpr int VB = 8 * (int)sizeof(T);                         // bytes of one 8-channel vector
    const int vecs = a.has_expand ? K >> 3 : cc >> 3;              // 8-channel vectors per staged pixel
    const unsigned char* in_b = reinterpret_cast<const unsigned char*>(a.in) + (int64_t)b * a.H * a.W * K * (int)sizeof(T);
    // thread -> (row, vector): vectors rounded up to a power of two so the split is a shift and a mask
    const int vsh = vecs <= 1 ? 0 : 32 - __builtin_clz(vecs - 1);
    const int v = threadIdx.x & ((1 << vsh) - 1), row0 = threadIdx.x >> vsh, rstride = MBF_THREADS >> vsh;
    if constexpr (MP != 0) {
      // ---- multi-pass expand (see the template comment): pass p = input channels [p * kp, (p + 1) * kp) ----
      const int Kp = a.kp, npass = a.npass, n_wrows = ntiles * 16;
      const int pvecs = Kp >> 3;                                     // 8-channel vectors per row and pass
      const int pvsh = pvecs <= 1 ? 0 : 32 - __builtin_clz(pvecs - 1);
      const int pv = threadIdx.x & ((1 << pvsh) - 1), prow0 = threadIdx.x >> pvsh, prs = MBF_THREADS >> pvsh;
      constexpr int NIW = TS == 16 ? 1 : 2, NIP = TS == 16 ? 2 : 3;   // row batches per lane and pass (host-checked: mbf_mp_fits)
      const unsigned char* w_b = reinterpret_cast<const unsigned char*>(a.we) + (int64_t)c0 * K * (int)sizeof(T);
      const int pix0 = (iy0 + r0) * a.W + ix0 + q0;
      constexpr bool RES = MP == 2;
      // RES: item i = tid + j * threads of the (weight rows + inside pixels) x (K / 8 vectors) grid; everything in flight at once
      constexpr int NJ = RES ? 6 : 1;
      raw_t rq0[NJ], rq1[NJ]; int qdst[NJ], qsl[NJ];                   // vectors, LDS element offset (>= 0: in a_s, < 0: -(1 + offset) in w_s), slice
      if constexpr (RES) {
        const int vk_ = K >> 3;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int i = threadIdx.x + j * MBF_THREADS, row = udiv_rcp(i, a.vk_rcp), vec = i - row * vk_;
          const bool ok = row < n_wrows + n_in, is_w = row < n_wrows;
          const int m = min(max(row - n_wrows, 0), n_in - 1), ri = row_of(m);
          const uint32_t off_w = (uint32_t)(min(row, n_wrows - 1) * K + vec * 8) * (uint32_t)sizeof(T);
          const uint32_t off_p = (uint32_t)((pix0 + ri * a.W + (m - ri * wi)) * K + vec * 8) * (uint32_t)sizeof(T);
          const unsigned char* src = (is_w ? w_b : in_b) + (ok ? (is_w ? off_w : off_p) : 0u);
          rq0[j] = *reinterpret_cast<const raw_t*>(src); if (!BF16) rq1[j] = *reinterpret_cast<const raw_t*>(src + 16);
          const int sl = udiv_rcp(vec, a.kpv_rcp), col = vec * 8 - sl * Kp;
          qsl[j] = ok ? sl : -1;
          qdst[j] = is_w ? -(1 + row * KP + col) : m * KP + col;
        }
      }
      raw_t w0[NIW], w1[NIW], p0[NIP], p1[NIP];
      auto issue = [&](int p) {
        if constexpr (RES) return;
        const int kb = p * Kp + pv * 8;
        const bool vk = pv < pvecs && kb < K;                       // (K is a multiple of 8: a vector is inside or outside)
#pragma unroll
        for (int j = 0; j < NIW; j++) {
          const int row = j * prs + prow0;
          const unsigned char* src = w_b + (vk && row < n_wrows ? (uint32_t)(row * K + kb) * (uint32_t)sizeof(T) : 0u);
          w0[j] = *reinterpret_cast<const raw_t*>(src); if (!BF16) w1[j] = *reinterpret_cast<const raw_t*>(src + 16);
        }
#pragma unroll
        for (int j = 0; j < NIP; j++) {
          const int mm = j * prs + prow0, m = min(mm, n_in - 1), ri = row_of(m);
          const uint32_t off = (uint32_t)((pix0 + ri * a.W + (m - ri * wi)) * K + kb) * (uint32_t)sizeof(T);
          const unsigned char* src = in_b + (vk && mm < n_in ? off : 0u);
          p0[j] = *reinterpret_cast<const raw_t*>(src); if (!BF16) p1[j] = *reinterpret_cast<const raw_t*>(src + 16);
        }
      };
      auto park = [&](int p) {
        if constexpr (RES) {
#pragma unroll
          for (int j = 0; j < NJ; j++)
            if (qsl[j] == p) {
              raw_t* d = qdst[j] < 0 ? reinterpret_cast<raw_t*>(w_s + (-1 - qdst[j])) : reinterpret_cast<raw_t*>(a_s + qdst[j]);
              d[0] = rq0[j]; if (!BF16) d[1] = rq1[j];
            }
          return;
        }
        const bool vk = pv < pvecs && p * Kp + pv * 8 < K;          // weights beyond K are parked as ZEROS (the activation columns there
#pragma unroll                                                        //  keep the previous pass's finite values: finite x 0)
        for (int j = 0; j < NIW; j++) {
          const int row = j * prs + prow0;
          if (pv < pvecs && row < n_wrows) {
            raw_t* d = reinterpret_cast<raw_t*>(w_s + row * KP + pv * 8);
            d[0] = vk ? w0[j] : raw_t{}; if (!BF16) d[1] = vk ? w1[j] : raw_t{};
          }
        }
#pragma unroll
        for (int j = 0; j < NIP; j++) {
          const int m = j * prs + prow0;
          if (vk && m < n_in) {
            raw_t* d = reinterpret_cast<raw_t*>(a_s + m * KP + pv * 8);
            d[0] = p0[j]; if (!BF16) d[1] = p1[j];
          }
        }
      };
      issue(0);
      {
        const int nv = PIN * EP * (int)sizeof(T) / 16;               // the whole [PIN][EP] tile: zeros = the depthwise conv's padding
        for (int i = threadIdx.x; i < nv; i += MBF_THREADS) reinterpret_cast<u32x4*>(e_s)[i] = (u32x4){0, 0, 0, 0};
      }
      XSTAMP(2); park_weights(); XSTAMP(3);
      park(0);
      MSTAMP(1); XSTAMP(4);
      __syncthreads();
      MSTAMP(2); XSTAMP(5);
      const int ntsh = ntiles <= 1 ? 0 : 32 - __builtin_clz(ntiles - 1);
      const int nt = wave & ((1 << ntsh) - 1), grp = wave >> ntsh, ngrp = MBF_WAVES >> ntsh;
      const int mtiles = (n_in + 15) >> 4, kspp = Kp / KSTEP;
      constexpr int MAXMT = TS == 16 ? 7 : 5, KSPMAX = 3;            // m-tiles per wave, k-steps per pass (host-checked)
      f32x4 acc[MAXMT];
#pragma unroll
      for (int i = 0; i < MAXMT; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int p = 0; p < npass; p++) {
        if (p + 1 < npass) issue(p + 1);                             // the next slice travels under this pass's MFMAs
        if (nt < ntiles) {
          raw_t wfr[KSPMAX];
          const T* wrow = w_s + (nt * 16 + r) * KP + KLANE * g;
#pragma unroll
          for (int ks = 0; ks < KSPMAX; ks++) {
            wfr[ks] = *reinterpret_cast<const raw_t*>(wrow + (ks < kspp ? ks * KSTEP : 0));
            if (p * Kp + ks * KSTEP + KLANE * g >= K) wfr[ks] = raw_t{};     // k >= K (last slice): the weight is the zero, the activation column holds finite leftovers
          }
#pragma unroll
          for (int i = 0; i < MAXMT; i++) {
            const int mt = grp + i * ngrp;
            if (mt < mtiles) {                                       // (wave-uniform)
              const T* arow = a_s + min(mt * 16 + r, n_in - 1) * KP + KLANE * g;
#pragma unroll
              for (int ks = 0; ks < KSPMAX; ks++) {
                if (ks < kspp) {                                     // (uniform)
                  const raw_t xa = *reinterpret_cast<const raw_t*>(arow + ks * KSTEP);
                  if constexpr (BF16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[ks]), __builtin_bit_cast(bf16x8, xa), acc[i], 0, 0, 0);
                  else {
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[ks][q], xa[q], acc[i], 0, 0, 0);
                  }
                }
              }
            }
          }
        }
        __syncthreads();                                             // every wave has read this slice
        if (p + 1 < npass) { park(p + 1); __syncthreads(); }
      }
      // bias + swish -> the expanded tile (only pixels inside the image; the rest stays zero)
      {
        const int n = nt * 16 + 4 * g;
        if (nt < ntiles && n < cc) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(be_s + n);
#pragma unroll
          for (int i = 0; i < MAXMT; i++) {
            const int m = (grp + i * ngrp) * 16 + r;
            if (m < n_in) {
              const int ri = row_of(m), pp_ = (r0 + ri) * PW + q0 + (m - ri * wi);
              float vv[4];
#pragma unroll
              for (int q = 0; q < 4; q++) vv[q] = acc[i][q] + bias[q];
              swish_n<BF16, 4>(vv);
              V::store4(e_s, (int64_t)pp_ * EP + n, vv);
            }
          }
        }
      }
      MSTAMP(3);
      __syncthreads();
      MSTAMP(4);
    } else if (a.has_expand) {
      // rows = the expand-weight rows c0 .. c0 + 16*ntiles, then the tile pixels INSIDE the image (compact rows
      // m = ri * wi + ci of the rectangle [r0,r1) x [q0,q1) of the PW x PW tile): only those are staged and expanded;
      // the rest of the expanded tile is the zero padding of the depthwise conv and is written as zeros right here
      const int n_wrows = ntiles * 16;
      // fp8 sessions: the expand weights are e4m3, rows padded to K16 = ceil16(K) bytes in memory and K16 + 16 in LDS
      const int wrow_b = F8 ? K16 : K * (int)sizeof(T), vecs_w = F8 ? K16 >> 4 : vecs, wv_b = F8 ? 16 : VB;
      const unsigned char* w_b = reinterpret_cast<const unsigned char*>(a.we) + (int64_t)c0 * wrow_b;
      const int pix0 = (iy0 + r0) * a.W + ix0 + q0;                // first inside pixel of the tile in the image
      const int nv = PIN * EP * (int)sizeof(T) / 16;               // the whole [PIN][EP] tile in 16-byte vectors
      for (int i = threadIdx.x; i < nv; i += MBF_THREADS) reinterpret_cast<u32x4*>(e_s)[i] = (u32x4){0, 0, 0, 0};
      // weight rows and pixel rows are separate batches (a per-row select between the two address forms compiled to
      // two branches per load); the first batch of each is in flight before anything is parked
      constexpr int NH = NB / 2;
      const bool vok_w = v < vecs_w, vok = v < vecs;
      const int wstep = rstride * wrow_b;
      raw_t w0[NH], w1[NH], p0[NH], p1[NH];
      auto issue_w = [&](int base) {
        uint32_t off = (uint32_t)((base + row0) * wrow_b + v * wv_b);
#pragma unroll
        for (int j = 0; j < NH; j++, off += wstep) {
          const unsigned char* src = w_b + (vok_w && base + j * rstride + row0 < n_wrows ? off : 0u);
          w0[j] = *reinterpret_cast<const raw_t*>(src); if (!BF16) w1[j] = *reinterpret_cast<const raw_t*>(src + 16);
        }
      };
      auto store_w = [&](int base) {
#pragma unroll
        for (int j = 0; j < NH; j++) {
          const int row = base + j * rstride + row0;
          if (vok_w && row < n_wrows) {
            raw_t* d = F8 ? reinterpret_cast<raw_t*>(w8_s + row * KP8 + v * 16) : reinterpret_cast<raw_t*>(w_s + row * KP + v * 8);
            d[0] = w0[j]; if (!BF16) d[1] = w1[j];
          }
        }
      };
      auto issue_p = [&](int base) {
#pragma unroll
        for (int j = 0; j < NH; j++) {
          const int mm = base + j * rstride + row0, m = min(mm, n_in - 1), ri = row_of(m);
          const uint32_t off = (uint32_t)((pix0 + ri * a.W + (m - ri * wi)) * K) * (uint32_t)sizeof(T) + (uint32_t)(v * VB);
          const unsigned char* src = in_b + (vok && mm < n_in ? off : 0u);
          p0[j] = *reinterpret_cast<const raw_t*>(src); if (!BF16) p1[j] = *reinterpret_cast<const raw_t*>(src + 16);
        }
      };
      auto store_p = [&](int base) {
#pragma unroll
        for (int j = 0; j < NH; j++) {
          const int m = base + j * rstride + row0;
          if (vok && m < n_in) {
            raw_t* d = reinterpret_cast<raw_t*>(a_s + m * KP + v * 8);
            d[0] = p0[j]; if (!BF16) d[1] = p1[j];
          }
        }
      };
      issue_w(0); issue_p(0);
      XSTAMP(2); park_weights(); XSTAMP(3);                       // their loads were issued first: this waits for them only
      store_w(0); store_p(0);
      for (int base = rstride * NH; base < n_wrows; base += rstride * NH) { issue_w(base); store_w(base); }
      for (int base = rstride * NH; base < n_in; base += rstride * NH) { issue_p(base); store_p(base); }
    } else {
      // depthwise alone: the PIN tile pixels of this block's channels, zero outside the image
      for (int base = 0; base < PIN; base += rstride * NB) {
        raw_t x0[NB], x1[NB];
        bool inside[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
          const int p = base + j * rstride + row0, ty = p / PW, gy = iy0 + ty, gx = ix0 + p - ty * PW;
          inside[j] = p < PIN && v < vecs && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          const uint32_t off = (uint32_t)((gy * a.W + gx) * K + c0) * (uint32_t)sizeof(T) + (uint32_t)(v * VB);
          const unsigned char* src = in_b + (inside[j] ? off : 0u);
          x0[j] = *reinterpret_cast<const raw_t*>(src); if (!BF16) x1[j] = *reinterpret_cast<const raw_t*>(src + 16);
        }
        if (base == 0) { XSTAMP(2); park_weights(); XSTAMP(3); }
#pragma unroll
        for (int j = 0; j < NB; j++) {
          const int p = base + j * rstride + row0;
          if (p < PIN && v < vecs) {
            raw_t* d = reinterpret_cast<raw_t*>(e_s + p * EP + v * 8);
            d[0] = inside[j] ? x0[j] : raw_t{}; if (!BF16) d[1] = inside[j] ? x1[j] : raw_t{};
          }
        }
      }
    }
  }
  if constexpr (MP == 0) {
  MSTAMP(1); XSTAMP(4);
  __syncthreads();
  MSTAMP(2); XSTAMP(5);
  }

  // ---- phase B: expand 1x1 + bias + swish -> e_s ----
  if (MP == 0 && a.has_expand) {
    // a wave takes TWO m-tiles per weight fragment (one LDS weight read feeds two MFMAs) and keeps
    // three k-steps of fragments in flight
    const int mpairs = (n_in + 31) >> 5;                           // pairs of 16-pixel m-tiles over the inside pixels
    const int ntsh = ntiles <= 1 ? 0 : 32 - __builtin_clz(ntiles - 1);   // pp -> (mp, nt): a shift and a mask
    constexpr int KSMAX = BF16 ? 6 : 12;                           // k-steps of weight fragments a wave keeps in registers (K <= 192)
    if (!F8 && ksteps <= KSMAX && (1 << ntsh) <= MBF_WAVES) {
      // A wave owns ONE n-tile for the whole phase: its weight fragments are read once into registers and every m-tile costs
      // ksteps activation reads + MFMAs + the epilogue.  As (m-pair, n-tile) items dealt round-robin each item re-read its
      // weights and ran ~260 instructions for 8 outputs per lane: the phase was issue-bound at 32 instructions per output.
      const int nt = wave & ((1 << ntsh) - 1), grp = wave >> ntsh, ngrp = MBF_WAVES >> ntsh;
      const int n = nt * 16 + 4 * g;
      if (nt < ntiles) {
        // (k >= K: the fragment is read from the START of the row - staged, finite data - and replaced by zeros.  The row's
        //  pad columns are never written: read as an operand they multiply the other side's zero by whatever an earlier
        //  launch left in LDS, and 0 x NaN is NaN - phi 0 @ 128 produced NaNs that way, K = 24.)
        raw_t wfr[KSMAX];
        const T* wrow = w_s + (nt * 16 + r) * KP;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ks++) {
          const bool kok = ks < ksteps && ks * KSTEP + KLANE * g < K;
          wfr[ks] = *reinterpret_cast<const raw_t*>(wrow + (kok ? ks * KSTEP + KLANE * g : 0));
          if (!kok) wfr[ks] = raw_t{};                              // k >= K: the weight is the zero (the activation read is clamped, finite)
        }
        const f32x4 bias = *reinterpret_cast<const f32x4*>(be_s + min(n, cc - 4));
        const int mtiles = (n_in + 15) >> 4;
        for (int mt = grp; mt < mtiles; mt += 2 * ngrp) {           // two m-tiles at a time: independent MFMA chains
          const int m0 = mt * 16 + r, m1 = m0 + 16 * ngrp;
          const T* arow0 = a_s + min(m0, n_in - 1) * KP;
          const T* arow1 = a_s + min(m1, n_in - 1) * KP;
          f32x4 acc0 = bias, acc1 = bias;
#pragma unroll
          for (int ks = 0; ks < KSMAX; ks++) {
            if (ks < ksteps) {                                      // (uniform)
              const int ko = ks * KSTEP + KLANE * g < K ? ks * KSTEP + KLANE * g : 0;
              const raw_t xa0 = *reinterpret_cast<const raw_t*>(arow0 + ko), xa1 = *reinterpret_cast<const raw_t*>(arow1 + ko);
              if constexpr (BF16) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[ks]), __builtin_bit_cast(bf16x8, xa0), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[ks]), __builtin_bit_cast(bf16x8, xa1), acc1, 0, 0, 0);
              } else {
#pragma unroll
                for (int q = 0; q < 4; q++) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[ks][q], xa0[q], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[ks][q], xa1[q], acc1, 0, 0, 0); }
              }
            }
          }
          if (n < cc) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
              const int m = half ? m1 : m0;
              if (m < n_in) {
                const f32x4 acc = half ? acc1 : acc0;
                const int ri = row_of(m), p = (r0 + ri) * PW + q0 + (m - ri * wi);   // tile position of inside pixel m
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = acc[q];
                swish_n<BF16, 4>(v);
                V::store4(e_s, (int64_t)p * EP + n, v);
              }
            }
          }
        }
      }
    } else
    for (int pp = wave; pp < (mpairs << ntsh); pp += MBF_WAVES) {
      const int nt = pp & ((1 << ntsh) - 1), mp = pp >> ntsh;
      if (nt >= ntiles) continue;
      const int m0 = mp * 32 + r, m1 = m0 + 16;
      const T* wrow = w_s + (int64_t)(nt * 16 + r) * KP;
      const T* arow0 = a_s + (int64_t)m0 * KP;
      const T* arow1 = a_s + (int64_t)m1 * KP;
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const unsigned char* w8row = w8_s + (int64_t)(nt * 16 + r) * KP8;
      const float inv_as = F8 ? 1.0f / a.a_scale : 1.0f;
      // (LDS reads are unconditional on clamped offsets - rows past the staged pixels and k past K read staged data and the
      //  activation fragment is zeroed by a select, so W needs none; a read under a branch costs a full wait per k-step)
      const int mc0 = min(m0, n_in - 1) - m0, mc1 = min(m1, n_in - 1) - m1;      // row clamps as element offsets below
      const bool mok0 = m0 < n_in, mok1 = m1 < n_in;
#pragma unroll 2
      for (int ks = 0; ks < ksteps; ks++) {
        const int k = ks * KSTEP + KLANE * g;
        const bool kok = k < K;
        const int ko = kok ? k : 0;                 // (k >= K: the start of the row, never its unwritten pad columns - 0 x NaN is NaN)
        raw_t xa0 = *reinterpret_cast<const raw_t*>(arow0 + mc0 * KP + ko), xa1 = *reinterpret_cast<const raw_t*>(arow1 + mc1 * KP + ko);
        if (!(kok && mok0)) xa0 = raw_t{};
        if (!(kok && mok1)) xa1 = raw_t{};
        if constexpr (F8) {
          const u32x2 wf8 = *reinterpret_cast<const u32x2*>(w8row + (kok ? ks * 32 + 8 * g : 0));
          const long wl = __builtin_bit_cast(long, wf8);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl, cvt_fp8x8(xa0, nullptr, inv_as), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl, cvt_fp8x8(xa1, nullptr, inv_as), acc1, 0, 0, 0);
        } else {
          const raw_t wf = *reinterpret_cast<const raw_t*>(wrow + ko);
          if constexpr (BF16) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xa0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xa1), acc1, 0, 0, 0);
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xa0[q], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xa1[q], acc1, 0, 0, 0); }
          }
        }
      }
      const int n = nt * 16 + 4 * g;          // lane: 4 consecutive expanded channels of tile pixels m0, m1
      if (n < cc) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(be_s + n);
        f32x4 ws = (f32x4){1.f, 1.f, 1.f, 1.f};
        if constexpr (F8) ws = *reinterpret_cast<const f32x4*>(a.we_scale + c0 + n) * a.a_scale;       // dequantisation: a_scale * w_scale[n]
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const int m = half ? m1 : m0;
          if (m < n_in) {
            const f32x4 acc = half ? acc1 : acc0;
            const int ri = row_of(m), p = (r0 + ri) * PW + q0 + (m - ri * wi);   // tile position of inside pixel m
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = F8 ? fmaf(acc[q], ws[q], bias[q]) : acc[q] + bias[q];
            swish_n<BF16, 4>(v);
            V::store4(e_s, (int64_t)p * EP + n, v);
          }
        }
      }
    }
    MSTAMP(3);
    __syncthreads();
  }
  if constexpr (MP == 0) { MSTAMP(4); }

  // ---- phase C: depthwise taps from LDS -> global, SE sums ----
  // A lane owns FOUR horizontally adjacent output pixels x FOUR channels (round 6; before: two pixels x eight channels): the 3 S + KS
  // input columns of a tap row are read (8 bytes in bf16) and unpacked once for all four pixels and the row's KS weight vectors are read
  // once - per tap row of a 5x5 stride-1 layer 8 + 5 LDS reads of 144 bytes and 32 unpack instructions instead of 6 + 10 reads of 256
  // bytes and 48.  Every output still accumulates its taps in the order (ky, kx) ascending, and the squeeze-excite sums keep their
  // tree (pixel pairs first, then pairs of pairs = this lane's four pixels, then the lanes): bit-identical to the two-pixel form.
  // The (pixel quad, channel quad) items are split over BOTH halves of the workgroup by tap row on the 8x8 tiles: waves 0-3 take the
  // first (KS+1)/2 rows of taps, waves 4-7 the rest and hand their partial sums over through LDS (the dead input tile) - the phase is a
  // chain of dependent LDS reads per tap row, so halving the rows per lane shortens it.
  const int cqs = cc >> 2;                                                 // channel quads of this chunk (cc is a multiple of 8)
  const int cqsh = cqs <= 1 ? 0 : 32 - __builtin_clz(cqs - 1), cqp = 1 << cqsh;
  // (TS 16: 64 pixel quads x 16 channel quads = one item per lane, all tap rows, no hand-over)
  constexpr bool SPLIT = TS == 8;
  constexpr int HALF = SPLIT ? MBF_THREADS / 2 : MBF_THREADS;
  const int half = SPLIT ? threadIdx.x / HALF : 0, tl = SPLIT ? threadIdx.x % HALF : threadIdx.x;
  const int cq = tl & (cqp - 1), qd = tl >> cqsh;                          // pixel quad 0 .. TS*TS/4 - 1
  f32x4* xch = reinterpret_cast<f32x4*>(smem);                             // [4][HALF] float4: partial sums of the second half (SPLIT)
  float sum[4] = {0, 0, 0, 0};
  const bool item = cq < cqs && qd < TS * TS / 4;
  constexpr int KH = SPLIT ? (KS + 1) / 2 : KS;
  constexpr int NX = 3 * S + KS;                                           // input columns under four adjacent outputs
  const int py = qd / (TS / 4), px = (qd % (TS / 4)) * 4;
  float acc[4][4];                                                         // [pixel][channel]
  if (item) {
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bdw_s + cq * 4);
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[j][c] = half ? 0.f : b0[c];
    }
#pragma unroll 1      // a real loop: unrolled, the compiler hoists all KS*KS weight reads and spills
    for (int ky = half ? KH : 0; ky < (half ? KS : KH); ky++) {
      f32x4 wr[KS];
#pragma unroll
      for (int kx = 0; kx < KS; kx++) wr[kx] = *reinterpret_cast<const f32x4*>(wdw_s + (ky * KS + kx) * a.CC + cq * 4);
      const int64_t erow = (int64_t)((py * S + ky) * PW + px * S) * EP + cq * 4;
#pragma unroll
      for (int hx = 0; hx < NX; hx++) {
        float ev[4];
        V::load4(e_s, erow + (int64_t)hx * EP, ev);
        // column hx is tap kx = hx - j S of output j: ascending hx = ascending kx for every output
#pragma unroll
        for (int j = 3; j >= 0; j--) {
          const int kx = hx - j * S;
          if (kx >= 0 && kx < KS) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[j][c] = fmaf(ev[c], wr[kx][c], acc[j][c]);
          }
        }
      }
    }
    if (SPLIT && half) {
#pragma unroll
      for (int j = 0; j < 4; j++) xch[j * HALF + tl] = (f32x4){acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
    }
  }
  __syncthreads();
  if (item && !half) {
    if constexpr (SPLIT) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 pj = xch[j * HALF + tl];
#pragma unroll
        for (int c = 0; c < 4; c++) acc[j][c] += pj[c];
      }
    }
    const int oy = oy0 + py, ox = ox0 + px;
    // (out_frag: the project GEMM's fragment order - row m = (image, pixel); a 16-byte unit holds KLANE channels from k:
    //  unit ((m / 16) * ksteps + k / KSTEP) * 64 + (k % KSTEP) / KLANE * 16 + m % 16; bf16: a lane's four channels are half a unit)
    auto ostore = [&](int ox_, const float* vv) {
      const int m = (b * a.Ho + oy) * a.Wo + ox_, k = c0 + cq * 4;
      if (!a.out_frag) { V::store4(a.out, (int64_t)m * a.Cexp + k, vv); return; }
      constexpr int KST = BF16 ? 32 : 16, KLN = BF16 ? 8 : 4;
      const int kst = (a.Cexp + KST - 1) / KST;
      const int64_t unit = (int64_t)((m >> 4) * kst + k / KST) * 64 + ((k % KST) / KLN) * 16 + (m & 15);
      V::store4(a.out, unit * KLN + (k & (KLN - 1)), vv);
    };
    if (oy < a.Ho) {
      // channel sums in the order of the two-pixel form: (0 + v0) + v1 per pixel pair, then the two pairs
      float sp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        // (the second pixel of a pair is only inside when the first one is: ox + j < Wo is the two-pixel form's nesting)
        if (ox + j < a.Wo) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; c++) v[c] = acc[j][c];
          swish_n<BF16, 4>(v);
#pragma unroll
          for (int c = 0; c < 4; c++) sp[j >> 1][c] += v[c];
          ostore(ox + j, v);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; c++) sum[c] = sp[0][c] + sp[1][c];
    }
  }
  MSTAMP(5);

  // ---- phase D: squeeze-excite.  The mean over the image and the reduce FC are linear in the channel
  //      sums, so every workgroup contributes the partial products of ITS channels and pixels:
  //        hpart[b][workgroup][j] = sum_{c in chunk} wr[j][c0 + c] * (sum of this tile's outputs of channel c)
  //      and the project GEMM (k_pw.hip) adds the rows up, applies 1/HW, bias and swish and runs the
  //      expand FC in its prologue - no squeeze-excite launch.  Fixed order everywhere (wave butterfly,
  //      the depthwise waves added in order, lane butterfly): bit-reproducible, no atomics. ----
  if (a.hpart) {
    constexpr int NWC = HALF / 64;                                        // waves that produced outputs
    float* ssum = reinterpret_cast<float*>(smem + a.off_e);               // [NWC][CC]: the expanded tile is dead
    // The reduce-FC weights of this thread's hidden unit (8 lanes share unit j = thread / 8; sq <= 64: one unit per thread
    // group) are requested HERE: their address depends on nothing computed below, the depthwise registers are dead, and the
    // L2 round trip (~1 us behind the output stores) then runs under the butterfly, the barrier and the channel sums instead of
    // after them.  (Requested at kernel start they cost eight live registers through every phase and lost, round 4.)
    const int part = threadIdx.x & 7, ch = part * 8;                      // 8 lanes share one hidden unit j
    const int jp = threadIdx.x >> 3;
    f32x4 wp0, wp1;
    {
      const f32x4* wp = reinterpret_cast<const f32x4*>(a.se_wr + (int64_t)min(jp, a.sq - 1) * a.Cexp + c0 + min(ch, max(cc - 8, 0)));
      wp0 = wp[0]; wp1 = wp[1];
    }
    if (wave < NWC) {
      for (int off = cqp; off < 64; off <<= 1) {                           // (the lane already holds a pair of pixel pairs: the first level of the old tree)
#pragma unroll
        for (int c = 0; c < 4; c++) sum[c] += __shfl_xor(sum[c], off, 64);
      }
      if (lane < cqs) *reinterpret_cast<f32x4*>(ssum + wave * a.CC + lane * 4) = (f32x4){sum[0], sum[1], sum[2], sum[3]};
    }
    __syncthreads();
    float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ch < cc) {
#pragma unroll
      for (int w4 = 0; w4 < NWC; w4++) {
        const f32x4* sp = reinterpret_cast<const f32x4*>(ssum + w4 * a.CC + ch);
        const f32x4 s0 = sp[0], s1 = sp[1];
#pragma unroll
        for (int c = 0; c < 4; c++) { cs[c] += s0[c]; cs[4 + c] += s1[c]; }
      }
    }
    // (round 4: requesting the reduce weights at kernel start - their addresses depend on nothing the kernel computes, at the end
    //  they are a dependent L2 round trip - measured SLOWER: 48.7k against 49.6k frames/s in bf16, 24.75k against 24.94k in fp32,
    //  three interleaved runs each; eight more live registers through every phase cost more than the round trip)
    float* hrow = a.hpart + ((int64_t)b * gridDim.x + bxl) * a.sqp;         // gridDim.x = tiles * chunks
    for (int j = threadIdx.x >> 3; j < ((a.sq + 63) & ~63); j += MBF_THREADS / 8) {
      float dot = 0.f;
      if (j < a.sq && ch < cc) {
        f32x4 w0 = wp0, w1 = wp1;
        if (j != jp) {                                                    // (more than MBF_THREADS / 8 hidden units: never at the reference's widths)
          const f32x4* wp = reinterpret_cast<const f32x4*>(a.se_wr + (int64_t)j * a.Cexp + c0 + ch);
          w0 = wp[0]; w1 = wp[1];
        }
        dot = ((w0[0] * cs[0] + w0[1] * cs[1]) + (w0[2] * cs[2] + w0[3] * cs[3])) + ((w1[0] * cs[4] + w1[1] * cs[5]) + (w1[2] * cs[6] + w1[3] * cs[7]));
      }
      dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64);
      if (part == 0 && j < a.sq) {
#ifdef HEP_ALT
        if (a.se_tail) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dot), __builtin_amdgcn_make_buffer_rsrc(hrow, 0, 0x7fffffff, 0x00020000), j * 4, 0, 17);   // sc0 sc1: written through
        else
#endif
        hrow[j] = dot;
      }
    }
#ifdef HEP_ALT      // (alternative library only: HEP_SE_TAIL=1, a measured loss against se_finish_kernel - NOTEBOOK.md)
    // ---- tail: the image's LAST workgroup finishes the squeeze-excite (hidden vector, expand FC, sigmoid -> scale[b][Cexp]) ----
    // No workgroup waits for another one: a ticket per image, and whoever draws the last one has every row in memory (each
    // workgroup's stores are acknowledged - vmcnt(0) - and its lanes met at a barrier before its ticket is drawn) and reads the
    // rows past its XCD's L2.  The counter is zero again when the launch ends.  (The memory-model-correct release / acquire pair at
    // agent scope writes back and invalidates the whole L2: 2-6 us per workgroup, MI355X_MICROARCH price list; this costs one
    // atomic round trip per workgroup and ~3 us in the one workgroup that runs the finish, against a ~4 us launch + its boundary.)
    if (a.se_tail) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      unsigned* flag = reinterpret_cast<unsigned*>(smem);                 // (everything in LDS is dead)
      if (threadIdx.x == 0) {
        unsigned* cnt = a.tail_counter + b * 32;
        const bool last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x;
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
      }
      __syncthreads();
      if (*flag) se_finish_body<BF16, MBF_THREADS, true>(a.tail, b, 0, a.tail.C, threadIdx.x, reinterpret_cast<float*>(smem) + 32);
    }
#endif
  }
#ifdef HEP_MBF_TRACE
  MSTAMP(6);
  if (g_mbf_trace && a.trace && lane == 0) {
    unsigned long long* o = g_mbf_trace + ((size_t)(b * gridDim.x + bxl) * MBF_WAVES + wave) * 8;
    for (int i = 0; i < 7; i++) o[i] = stamps[i];
    o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  }
#endif
}

#ifdef HEP_MBF_TRACE
extern "C" int hep_dbg_mbf_trace(unsigned long long* host, int max_waves, int enable) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)1 << 21;
  if (!buf) { if (hipMalloc((void**)&buf, cap * 8) != hipSuccess) return -1; hipMemset(buf, 0, cap * 8); }
  unsigned long long* p = enable ? buf : nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_mbf_trace), &p, sizeof p);
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)max_waves * 64, hipMemcpyDeviceToHost); }
  return (int)(cap / 8);
}
#endif

// rows of the compact input tile: the most in-image pixels any ts x ts-output tile of an H x W map covers
int mbf_max_inside(int H, int W, int k, int s, int pad_t, int pad_l, int ts) {
  const int pw = (ts - 1) * s + k;
  auto span = [&](int n, int pad) {
    const int no = (n + s - 1) / s;                        // output size along this axis (SAME)
    int best = 0;
    for (int t0 = 0; t0 < no; t0 += ts) {
      const int i0 = t0 * s - pad, lo = std::max(0, -i0), hi = std::min(pw, n - i0);
      best = std::max(best, hi - lo);
    }
    return best;
  };
  return span(H, pad_t) * span(W, pad_l);
}

int mbf_threads(int ts) { return ts == 16 ? 1024 : 512; }

// kp: input channels per pass of the multi-pass expand (0 or >= Cin: one pass, the whole K in LDS)
size_t mbf_lds_layout(int Cin, int CC, int k, int s, int bf16, int has_expand, int max_inside, int ts, MbfArgs* a, int kp) {
  const size_t es = bf16 ? 2 : 4, pad = bf16 ? 8 : 4;
  const size_t pw = (size_t)(ts - 1) * s + k, pin = pw * pw;
  const size_t arows = (((size_t)std::min<int>(max_inside, (int)pin) + 31) / 32) * 32;   // phase B reads whole pairs of 16-row m-tiles
  const bool mp = has_expand && kp > 0 && kp < Cin;
  const size_t kcols = mp ? (size_t)kp : (size_t)Cin;
  size_t in_bytes = has_expand ? arows * (kcols + pad) * es : 0;
  // the tap-row hand-over of phase C lives there too (8x8 tiles: two halves of the workgroup; the multi-pass 16x16 form has none)
  in_bytes = std::max(in_bytes, mp && ts == 16 ? (size_t)0 : (size_t)mbf_threads(ts) * 9 * 4);
  in_bytes = (in_bytes + 15) & ~(size_t)15;
  const size_t e_bytes = (std::max(pin * (CC + pad) * es, (size_t)16 * CC * 4) + 15) & ~(size_t)15;   // (phase D: [<= 16][CC] channel sums)
  const size_t w_bytes = (size_t)(k * k + 2) * CC * 4;                    // depthwise weights + the two bias vectors
  const size_t we_bytes = has_expand ? (((size_t)CC * (kcols + pad) * es + 15) & ~(size_t)15) : 0;   // expand-weight chunk (one K slice of it)
  if (a) { a->off_e = in_bytes; a->off_we = in_bytes + e_bytes; a->off_w = a->off_we + we_bytes; a->lds_bytes = a->off_w + w_bytes; }
  return in_bytes + e_bytes + we_bytes + w_bytes;
}

// does a multi-pass plan fit the kernel's compile-time budgets (row batches per lane and pass, k-steps per pass, m-tiles per wave)?
int mbf_mp_fits(int CC, int kp, int bf16, int max_inside, int ts) {
  const int threads = mbf_threads(ts), waves = threads / 64, kstep = bf16 ? 32 : 16;
  if (kp <= 0 || kp % kstep != 0 || kp / kstep > 3) return 0;
  int vp = 1; while (vp < kp / 8) vp <<= 1;
  const int prs = threads / vp, niw = ts == 16 ? 1 : 2, nip = ts == 16 ? 2 : 3;
  const int ntiles = (CC + 15) / 16; int np2 = 1; while (np2 < ntiles) np2 <<= 1;
  if (np2 > waves) return 0;
  const int ngrp = waves / np2, mtiles = (max_inside + 15) / 16, maxmt = ts == 16 ? 7 : 5;
  return ntiles * 16 <= niw * prs && max_inside <= nip * prs && (mtiles + ngrp - 1) / ngrp <= maxmt;
}
// ... and can the whole tile (weight rows + inside pixels, K / 8 vectors each) be held in six vectors per lane (MP = 2)?
int mbf_mp_resident(int Cin, int CC, int max_inside, int ts) {
  const long items = (long)(((CC + 15) / 16) * 16 + max_inside) * (Cin / 8);
  return Cin % 8 == 0 && items <= 6L * mbf_threads(ts);
}

template <bool BF16, int KS, int S, int TS>
static int prep_one() {
  int rc = hipFuncSetAttribute(reinterpret_cast<const void*>(mbf_kernel<BF16, KS, S, TS, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess ? 0 : -1;
  rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(mbf_kernel<BF16, KS, S, TS, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess ? 0 : -1;
  rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(mbf_kernel<BF16, KS, S, TS, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess ? 0 : -1;
#ifdef HEP_WITH_FP8
  if constexpr (BF16) rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(mbf_kernel<true, KS, S, TS, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess ? 0 : -1;
#endif
  return rc;
}
int mbf_prepare(void) {
  return prep_one<true, 3, 1, 8>() | prep_one<true, 3, 2, 8>() | prep_one<true, 5, 1, 8>() | prep_one<true, 5, 2, 8>() |
         prep_one<false, 3, 1, 8>() | prep_one<false, 3, 2, 8>() | prep_one<false, 5, 1, 8>() | prep_one<false, 5, 2, 8>() |
         prep_one<true, 3, 1, 16>() | prep_one<true, 5, 1, 16>() | prep_one<false, 3, 1, 16>() | prep_one<false, 5, 1, 16>();
}

template <bool BF16, bool F8, int MP>
static void launch_mbf_t(const MbfArgs& a, dim3 grid, hipStream_t s) {
  if (a.ts == 16) {       // stride 1 only (the planner never asks for 16x16 tiles on a stride-2 layer)
    if (a.k == 3) hipLaunchKernelGGL((mbf_kernel<BF16, 3, 1, 16, F8, MP>), grid, dim3(1024), a.lds_bytes, s, a);
    else hipLaunchKernelGGL((mbf_kernel<BF16, 5, 1, 16, F8, MP>), grid, dim3(1024), a.lds_bytes, s, a);
  }
  else if (a.k == 3 && a.s == 1) hipLaunchKernelGGL((mbf_kernel<BF16, 3, 1, 8, F8, MP>), grid, dim3(512), a.lds_bytes, s, a);
  else if (a.k == 3 && a.s == 2) hipLaunchKernelGGL((mbf_kernel<BF16, 3, 2, 8, F8, MP>), grid, dim3(512), a.lds_bytes, s, a);
  else if (a.k == 5 && a.s == 1) hipLaunchKernelGGL((mbf_kernel<BF16, 5, 1, 8, F8, MP>), grid, dim3(512), a.lds_bytes, s, a);
  else hipLaunchKernelGGL((mbf_kernel<BF16, 5, 2, 8, F8, MP>), grid, dim3(512), a.lds_bytes, s, a);
}
void launch_mbf(const MbfArgs& a_, hipStream_t s) {
  MbfArgs a = a_;
#ifdef HEP_MBF_TRACE
  {   // profiling build: HEP_MBF_TRACE_SEL="Cexp,H" keeps the stamps of that layer only (default: every launch, last wins)
    static const char* sel = getenv("HEP_MBF_TRACE_SEL");
    int c = 0, h = 0;
    if (sel) sscanf(sel, "%d,%d", &c, &h);
    a.trace = !sel || (a.Cexp == c && a.H == h);
  }
#endif
  const int tiles = ((a.Wo + a.ts - 1) / a.ts) * ((a.Ho + a.ts - 1) / a.ts), chunks = (a.Cexp + a.CC - 1) / a.CC;
  dim3 grid(tiles * chunks, a.B);
  a.chunks = chunks; a.tiles_x = (a.Wo + a.ts - 1) / a.ts;
  a.chunks_rcp = rcp_u32(chunks); a.tiles_x_rcp = rcp_u32(a.tiles_x); a.gx_rcp = rcp_u32(grid.x);
#ifdef HEP_WITH_FP8
  if (a.bf16 && a.fp8) { launch_mbf_t<true, true, 0>(a, grid, s); return; }
#endif
  const int mp = a.has_expand && a.npass > 1 ? (a.mp_resident ? 2 : 1) : 0;
  a.vk_rcp = rcp_u32((uint32_t)std::max(1, a.Cin >> 3)); a.kpv_rcp = rcp_u32((uint32_t)std::max(1, a.kp >> 3));
  if (a.bf16) { if (mp == 2) launch_mbf_t<true, false, 2>(a, grid, s); else if (mp == 1) launch_mbf_t<true, false, 1>(a, grid, s); else launch_mbf_t<true, false, 0>(a, grid, s); }
  else { if (mp == 2) launch_mbf_t<false, false, 2>(a, grid, s); else if (mp == 1) launch_mbf_t<false, false, 1>(a, grid, s); else launch_mbf_t<false, false, 0>(a, grid, s); }
}
