// k_xbf.hip - one MBConv block BOUNDARY of the EfficientNet backbone on the big maps (128x128, 64x64) as ONE
// gfx950 kernel.  The reference runs, between the depthwise conv of block i-1 and the depthwise conv of block i
// (efficientnet/model.py:86-104 of block i-1, :76-89 of block i):
//
//   _se_reduce,_swish,_se_expand,sigmoid,*   (squeeze-excite of block i-1; its means arrive as partial reduce-FC rows)
//   _project_conv,_bn2 (+ inputs)            1x1, K1 -> N1
//   _expand_conv,_bn0,_swish                 1x1, N1 -> Cexp = 6 N1
//   _depthwise_conv,_bn1,_swish              k x k, stride s, TF-SAME
//   adaptive_avg_pool2d (spatial half)       squeeze-excite of block i
//
// As separate launches that is three kernels and ~10x the unavoidable HBM traffic on the early blocks: the project
// writes the block output, the expand reads it and writes the 6x expanded tensor, the depthwise reads that back.
// Here a workgroup owns one output tile of one image: it stages the (halo'd) tile of block i-1's depthwise output in
// LDS, finishes the squeeze-excite scale, runs both 1x1 convs on MFMA with the project result handed to the expand
// conv IN REGISTERS (the transposed product D[n, pixel] = W . A^T leaves a lane with 4 consecutive channels of one
// pixel, which is exactly the k-fragment of v_mfma_f32_16x16x16_bf16 / v_mfma_f32_16x16x4_f32), writes the activated
// expanded tile to LDS and takes the depthwise taps from there.  HBM sees the input tile once (plus halo) and the
// depthwise output once.  Stored only when somebody needs it: the block output (a later residual / BiFPN tap).
//
//   P0  blob of all weights (host-packed in LDS layout) + input tile -> LDS; squeeze-excite prologue -> scale[K1]
//   P1  project: A frags (LDS) * scale -> MFMA -> + bias (+ residual) -> rounded to the session dtype -> x frags (registers)
//   P2  expand (per chunk of expanded channels): x frags . W2 -> + bias, swish -> LDS (zero outside the image)
//   P3  depthwise taps from LDS -> + bias, swish -> global; per-lane channel sums
//   P4  channel sums (fixed order) -> partial reduce-FC products of block i -> hpart_out[b][tile][j]
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;

// Optional per-wave timeline (make trace: -DHEP_XBF_TRACE): s_memrealtime (100 MHz) stamps at the phase boundaries of
// the launches selected by HEP_XBF_TRACE_SEL="Cexp", read back with hep_dbg_xbf_trace() - profiling builds only.
#ifdef HEP_XBF_TRACE
__device__ unsigned long long* g_xbf_trace = nullptr;
#define XSTAMP(i) do { stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XSTAMP(i)
#endif

namespace {
constexpr int XT = 512, XW = XT / 64;      // threads, waves per workgroup
constexpr int XCH = 2;                      // LDS chunks of expanded channels (<=)
}

// K1C / NT2C: the depthwise width of block i-1 and the expanded n-tiles of block i as compile-time constants (the BASELINE
// networks' boundaries: every index split, loop bound and LDS pitch folds), or 0 = taken from the arguments (any width).
template <bool BF16, int KS, int S, int TOH, int TOW, int NT1, int K1C, int NT2C>
__global__ __launch_bounds__(XT, KS == 3 ? 4 : 2) void xbf_kernel(XbfArgs a) {   // (3x3 layers: two workgroups per CU, at most 128 VGPRs)
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;      // 16 bytes of activations / weights
  typedef typename std::conditional<BF16, u32x2, f32x4>::type xfrag_t;    // 4 consecutive channels of one pixel
  constexpr int ES = (int)sizeof(T);
  constexpr int KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4, PAD = BF16 ? 8 : 4;
  constexpr int PH = (TOH - 1) * S + KS, PW = (TOW - 1) * S + KS, PIN = PH * PW;
  constexpr int MT_TOTAL = (PIN + 15) / 16;
  constexpr int K2P = NT1 * 16, W2P = K2P + PAD;
  constexpr int U = 3;                                                    // expanded n-tiles per item of P2
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, a.lds_bytes);
#ifdef HEP_XBF_TRACE
  unsigned long long stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  XSTAMP(0);
  int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;     // (re-derived per tile, see the tile loop)
  const int K1 = K1C ? K1C : a.K1, N1 = a.N1, NT2 = NT2C ? NT2C : a.NT2, Cexp = NT2 * 16;
  const int W1P = K1 + PAD;
  T* a_s = reinterpret_cast<T*>(smem);                                    // [PIN][K1]  input tile (dead after P1)
  T* e_s = reinterpret_cast<T*>(smem);                                    // [PIN][EP]  expanded, activated tile of one chunk
  const T* w1_s = reinterpret_cast<const T*>(smem + a.off_w1);            // [NT1*16][W1P]
  const T* w2_s = reinterpret_cast<const T*>(smem + a.off_w2);            // [NT2*16][W2P]
  const float* wdw_s = reinterpret_cast<const float*>(smem + a.off_f);    // [KS*KS][Cexp]
  const float* bdw_s = wdw_s + KS * KS * Cexp;                            // [Cexp]
  const float* b2_s = bdw_s + Cexp;                                       // [NT2*16]
  const float* b1_s = b2_s + NT2 * 16;                                  // [NT1*16]
  float* scale_s = reinterpret_cast<float*>(smem + a.off_misc);           // [ceil16(K1)]
  float* hid_s = scale_s + ((K1 + 15) & ~15);                             // [16]
  float* red_s = hid_s + 16;                                              // [XW][16]
  float* csum_s = red_s + XW * 16;                                        // [Cexp]

  // XCD-aware order: neighbouring tiles of one image (shared halo rows / columns) are consecutive logical blocks
  const int logical = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
  // a workgroup runs a.tpw consecutive tiles of ONE image: the weight blob, the squeeze-excite prologue and the channel-sum
  // epilogue are per workgroup, and consecutive tiles share halo columns
  const int wgs = gridDim.x;                                              // workgroups per image
  const int b = udiv_rcp(logical, a.tiles_rcp), wgt = logical - b * wgs;
  const int tile0 = wgt * a.tpw;
  int oy0, ox0, iy0, ix0;
  auto set_tile = [&](int tile) {
    const int tyi = udiv_rcp(tile, a.tiles_x_rcp), txi = tile - tyi * a.tiles_x;
    oy0 = tyi * TOH; ox0 = txi * TOW; iy0 = oy0 * S - a.pad_t; ix0 = ox0 * S - a.pad_l;
  };
  set_tile(tile0);

  // ---------------- P0: everything global -> LDS, loads issued before the first wait ----------------
  // (1) weight blob: linear copy, up to WB 16-byte vectors per lane in flight
  constexpr int WB = 6;
  const int nblob = a.blob_bytes >> 4;
  u32x4 wv[WB];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.blob);
#pragma unroll
    for (int j = 0; j < WB; j++) wv[j] = src[min(tid + j * XT, nblob - 1)];
  }
  // (2) squeeze-excite inputs of block i-1: partial reduce-FC rows (summed per lane), this lane's expand-FC row and biases
  const int sqp = a.sqp, sq = a.sq;                       // sqp is 8 or 16
  const int sqsh = sqp == 8 ? 3 : 4;
  const int sj = tid & (sqp - 1), sgrp = tid >> sqsh, SG = XT >> sqsh;
  const float* hp = a.hpart + (int64_t)b * a.se_rows * sqp + sj;
  float hv[4];                                             // (consumed behind the input-tile loads: nothing waits before they are issued)
#pragma unroll
  for (int q = 0; q < 4; q++) hv[q] = hp[(uint32_t)(min(sgrp + q * SG, a.se_rows - 1) * sqp)];
  constexpr int JV = BF16 ? 8 : 4;                         // hidden units per 16-byte vector of the expand FC
  const int nsv = sqp / JV;                                // 1..4
  raw_t sev[4];
  {
    const raw_t* wr = reinterpret_cast<const raw_t*>(reinterpret_cast<const T*>(a.se_we) + (uint32_t)(min(tid, K1 - 1) * sqp));
#pragma unroll
    for (int q = 0; q < 4; q++) sev[q] = wr[min(q, nsv - 1)];
  }
  const float se_bev = a.se_be[min(tid, K1 - 1)], se_brv = a.se_br[min(tid, sq - 1)];
  // (3) input tile: a tile row is ONE contiguous run of PW * K1 elements in NHWC memory; wave w stages the rows w, w + XW, ..
  // with its lanes sweeping the row in 16-byte vectors, so the per-lane address is a uniform row offset + 16 * lane.  Rows /
  // columns outside the image read clamped (valid) addresses: those pixels are masked behind the expand conv.
  constexpr int NB = 8, RPW = (PH + XW - 1) / XW;
  const int RV = (PW * K1 * ES) >> 4;                      // vectors per tile row
  const int SW = (RV + 63) >> 6, nq = RPW * SW;            // sweeps per row, (row, sweep) slots per wave
  const int img_bytes = a.H * a.W * K1 * ES;
  const unsigned char* img = reinterpret_cast<const unsigned char*>(a.in) + (int64_t)b * img_bytes;
  u32x4 x0[NB];
  auto issue_in = [&](int q0) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int q = min(q0 + j, nq - 1), rs = udiv_rcp(q, a.sw_rcp), ty = min(wave + rs * XW, PH - 1), v = lane + ((q - rs * SW) << 6);
      const int rowoff = (min(max(iy0 + ty, 0), a.H - 1) * a.W + ix0) * K1 * ES;          // (uniform)
      x0[j] = *reinterpret_cast<const u32x4*>(img + (uint32_t)min(max(rowoff + (v << 4), 0), img_bytes - 16));
    }
  };
  auto park_in = [&](int q0) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int q = q0 + j, rs = udiv_rcp(min(q, nq - 1), a.sw_rcp), ty = wave + rs * XW, v = lane + ((q - rs * SW) << 6);
      if (q < nq && ty < PH && v < RV) *reinterpret_cast<u32x4*>(smem + (uint32_t)((ty * RV + v) << 4)) = x0[j];
    }
  };
  issue_in(0);
  XSTAMP(1);
  // park the blob (its loads are the oldest)
  {
    u32x4* dst = reinterpret_cast<u32x4*>(smem + a.off_w1);
#pragma unroll
    for (int j = 0; j < WB; j++) if (tid + j * XT < nblob) dst[tid + j * XT] = wv[j];
    const u32x4* src = reinterpret_cast<const u32x4*>(a.blob);
    for (int i = tid + WB * XT; i < nblob; i += XT) dst[i] = src[i];
  }
  XSTAMP(2);
  // squeeze-excite: hidden = swish(inv_hw * sum_rows hpart + br); scale[k] = sigmoid(we[k,:] . hidden + be[k])
  float hsum = 0.f;
#pragma unroll
  for (int q = 0; q < 4; q++) hsum += sgrp + q * SG < a.se_rows ? hv[q] : 0.f;
  for (int row = sgrp + 4 * SG; row < a.se_rows; row += SG) hsum += hp[(uint32_t)(row * sqp)];
  for (int off = sqp; off < 64; off <<= 1) hsum += __shfl_xor(hsum, off, 64);
  if (lane < sqp) red_s[wave * 16 + lane] = hsum;
  __syncthreads();
  if (tid < sqp) {
    float h = 0.f;
    if (tid < sq) {
      float s_ = 0.f;
#pragma unroll
      for (int w = 0; w < XW; w++) s_ += red_s[w * 16 + tid];
      h = swishf(fmaf(s_, a.inv_hw, se_brv));
    }
    hid_s[tid] = h;
  }
  XSTAMP(3);
  park_in(0);
  for (int q0 = NB; q0 < nq; q0 += NB) { issue_in(q0); park_in(q0); }
  XSTAMP(4);
  __syncthreads();
  for (int k = tid; k < K1; k += XT) {
    float e0 = 0.f, e1 = 0.f;
    const raw_t* wr = reinterpret_cast<const raw_t*>(reinterpret_cast<const T*>(a.se_we) + (uint32_t)(k * sqp));
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q < nsv) {
        const raw_t w = k == tid ? sev[q] : wr[q];
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(hid_s + q * JV);
        if constexpr (BF16) {
          const f32x4 h1 = *reinterpret_cast<const f32x4*>(hid_s + q * JV + 4);
          e0 = fmaf(__uint_as_float(w[0] << 16), h0[0], e0); e1 = fmaf(__uint_as_float(w[0] & 0xffff0000u), h0[1], e1);
          e0 = fmaf(__uint_as_float(w[1] << 16), h0[2], e0); e1 = fmaf(__uint_as_float(w[1] & 0xffff0000u), h0[3], e1);
          e0 = fmaf(__uint_as_float(w[2] << 16), h1[0], e0); e1 = fmaf(__uint_as_float(w[2] & 0xffff0000u), h1[1], e1);
          e0 = fmaf(__uint_as_float(w[3] << 16), h1[2], e0); e1 = fmaf(__uint_as_float(w[3] & 0xffff0000u), h1[3], e1);
        } else {
          e0 = fmaf(w[0], h0[0], e0); e1 = fmaf(w[1], h0[1], e1); e0 = fmaf(w[2], h0[2], e0); e1 = fmaf(w[3], h0[3], e1);
        }
      }
    }
    scale_s[k] = sigmoidf((e0 + e1) + (k == tid ? se_bev : a.se_be[k]));
  }
  __syncthreads();
  XSTAMP(5);

  float sum[XCH][8];
#pragma unroll
  for (int ci = 0; ci < XCH; ci++)
#pragma unroll
    for (int c = 0; c < 8; c++) sum[ci][c] = 0.f;
  int cgs_of[XCH] = {0, 0};
#pragma unroll 1
  for (int tt = 0; tt < a.tpw; tt++) {
  if (tile0 + tt >= a.tiles) break;                                    // (uniform)
  // The thread's index split is made opaque per tile: otherwise every address / mask expression of P1..P3 that does not
  // depend on the tile is hoisted out of this loop and kept alive across it (+50 VGPRs: one workgroup per CU instead of two)
  asm volatile("" : "+v"(tid));
  lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6); r = lane & 15; g = lane >> 4;
  if (tt > 0) {                                                        // the first tile was staged by P0
    set_tile(tile0 + tt);
    issue_in(0); park_in(0);
    for (int q0 = NB; q0 < nq; q0 += NB) { issue_in(q0); park_in(q0); }
    __syncthreads();
  }
  // ---------------- P1: project 1x1, each wave its m-tiles (16 tile pixels); results stay in registers ----------------
  const int ksteps1 = (K1 + KSTEP - 1) / KSTEP;
  const T* res_b = reinterpret_cast<const T*>(a.res) + (a.res ? (int64_t)b * a.H * a.W * N1 : 0);
  T* mid_b = reinterpret_cast<T*>(a.mid) + (a.mid ? (int64_t)b * a.H * a.W * N1 : 0);
  constexpr int MAXMT = (MT_TOTAL + XW - 1) / XW;
  xfrag_t xf[MAXMT][NT1];
#pragma unroll
  for (int i = 0; i < MAXMT; i++) {
    const int mt = wave + i * XW;
#pragma unroll
    for (int t = 0; t < NT1; t++) xf[i][t] = xfrag_t{};
    if (mt >= MT_TOTAL) continue;                                     // (uniform)
    const int p = mt * 16 + r, pc = min(p, PIN - 1);
    const int ty = pc / PW, tx = pc - ty * PW;
    const int gy = iy0 + ty, gx = ix0 + tx;
    f32x4 acc[NT1];
#pragma unroll
    for (int t = 0; t < NT1; t++) acc[t] = *reinterpret_cast<const f32x4*>(b1_s + t * 16 + 4 * g);      // bias rides in the accumulator
    const T* arow = a_s + pc * K1;
    for (int ks = 0; ks < ksteps1; ks++) {
      const int k = ks * KSTEP + KLANE * g, kc = min(k, K1 - KLANE);
      raw_t av = *reinterpret_cast<const raw_t*>(arow + kc);
      if constexpr (BF16) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale_s + kc), s1 = *reinterpret_cast<const f32x4*>(scale_s + kc + 4);
        const float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
        for (int q = 0; q < 4; q++)
          av[q] = pack_bf16x2(__uint_as_float(av[q] << 16) * sc[2 * q], __uint_as_float(av[q] & 0xffff0000u) * sc[2 * q + 1]);
        if (k >= K1) av = (u32x4){0, 0, 0, 0};
      } else {
        av *= *reinterpret_cast<const f32x4*>(scale_s + kc);
        if (k >= K1) av = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int t = 0; t < NT1; t++) {
        const raw_t wf = *reinterpret_cast<const raw_t*>(w1_s + (t * 16 + r) * W1P + kc);
        if constexpr (BF16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, av), acc[t], 0, 0, 0);
        else {
#pragma unroll
          for (int q = 0; q < 4; q++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], av[q], acc[t], 0, 0, 0);
        }
      }
    }
    // epilogue: lane holds channels n = 16 t + 4 g + {0..3} of tile pixel p
    const int pix = min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1);
    const bool owned = p < PIN && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W &&
                       ty >= a.pad_t && ty < a.pad_t + TOH * S && tx >= a.pad_l && tx < a.pad_l + TOW * S;
#pragma unroll
    for (int t = 0; t < NT1; t++) {
      const int n = t * 16 + 4 * g;
      float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
      if (a.res && n < N1) {
        float rr[4];
        V::load4(res_b, (int64_t)(pix * N1 + n), rr);
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] += rr[q];
      }
      xfrag_t xv;
      if constexpr (BF16) { xv[0] = pack_bf16x2(v[0], v[1]); xv[1] = pack_bf16x2(v[2], v[3]); }
      else xv = (f32x4){v[0], v[1], v[2], v[3]};
      xf[i][t] = xv;
      if (a.mid && owned && n < N1) *reinterpret_cast<xfrag_t*>(mid_b + (int64_t)(pix * N1 + n)) = xv;
    }
  }
  XSTAMP(6);
  __syncthreads();                                                     // a_s is dead: e_s may be written

  // ---------------- P2 + P3 per chunk of expanded channels ----------------
#pragma unroll
  for (int ci = 0; ci < XCH; ci++) {
    if (ci >= a.nchunks) continue;                                      // (uniform)
    const int nt_begin = ci * a.chunk_tiles, nt_end = min(NT2, nt_begin + a.chunk_tiles);
    const int c0 = nt_begin * 16, cc = (nt_end - nt_begin) * 16;
    const int EP = a.chunk_tiles * 16 + PAD;
    // P2: expand.  A wave keeps the project results of ITS m-tiles in registers (xf) and runs them against every n-tile of
    // the chunk, U at a time: the pixel decode, the mask and the x fragments are per m-tile, not per (m-tile, n-tile)
#pragma unroll
    for (int i = 0; i < MAXMT; i++) {
      const int mt = wave + i * XW;
      if (mt >= MT_TOTAL) continue;                                    // (uniform)
      const int p = mt * 16 + r, pc = min(p, PIN - 1);
      const int ty = pc / PW, tx = pc - ty * PW;
      const int gy = iy0 + ty, gx = ix0 + tx;
      const bool ins = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      T* erow = e_s + p * EP + 4 * g - c0;
#pragma unroll 1     // (unrolled, the compiler hoists the weight-fragment reads of every n-tile: 230 VGPRs)
      for (int n20 = nt_begin; n20 < nt_end; n20 += U) {
        f32x4 acc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int n2 = min(n20 + u, nt_end - 1);
          acc[u] = *reinterpret_cast<const f32x4*>(b2_s + n2 * 16 + 4 * g);
          const T* wrow = w2_s + (n2 * 16 + r) * W2P + 4 * g;
#pragma unroll
          for (int t = 0; t < NT1; t++) {
            if constexpr (BF16) {
              const u32x2 wf = *reinterpret_cast<const u32x2*>(wrow + t * 16);
              acc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, wf), __builtin_bit_cast(s16x4, xf[i][t]), acc[u], 0, 0, 0);
            } else {
              const f32x4 wf = *reinterpret_cast<const f32x4*>(wrow + t * 16);
#pragma unroll
              for (int q = 0; q < 4; q++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xf[i][t][q], acc[u], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (n20 + u < nt_end && p < PIN) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = acc[u][q];
            swish_n<BF16, 4>(v);
            xfrag_t ev;
            if constexpr (BF16) { ev[0] = ins ? pack_bf16x2(v[0], v[1]) : 0u; ev[1] = ins ? pack_bf16x2(v[2], v[3]) : 0u; }
            else ev = ins ? (f32x4){v[0], v[1], v[2], v[3]} : (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<xfrag_t*>(erow + (n20 + u) * 16) = ev;
          }
        }
      }
    }
    if (ci == 0) XSTAMP(7);
    __syncthreads();
    if (ci == 0) XSTAMP(8);
    // P3: depthwise.  Thread -> (pixel-quad group, 4-channel group); the channel quad of a thread is fixed over its items.  An item is FOUR
    // horizontally adjacent output pixels x four channels (round 6; before: two pixels x eight channels): the 3 S + KS input columns of a
    // tap row are read and unpacked once for the four pixels and the row's KS weight vectors once.  Every output accumulates its taps in
    // the order (ky, kx) ascending as before, and the squeeze-excite sums are kept per PIXEL PAIR in the two-pixel form's bookkeeping: the
    // quad group qg carries the running sums of the old pixel-pair groups 2 qg and 2 qg + 1 (their items are the two halves of this
    // thread's quads, in the same order) and P4 reads them from the rows those groups used to write - bit-identical whenever the number
    // of pixel-pair groups XT / cgs is even (every BASELINE boundary; an odd count drops its last group: another fixed order).
    const int cgs = cc >> 3, cqs = cc >> 2;
    cgs_of[ci] = cgs;
    const float cgs_inv = __builtin_amdgcn_rcpf((float)cgs), cqs_inv = __builtin_amdgcn_rcpf((float)cqs);
    const int npg = udiv_f(XT, cgs, cgs_inv) & ~1, nqg = npg >> 1;     // pixel-pair groups (even), pixel-quad groups
    const int qg = udiv_f(tid, cqs, cqs_inv), cq = tid - qg * cqs;
    constexpr int NQ = TOH * TOW / 4, NX = 3 * S + KS;
    if (qg < nqg) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bdw_s + c0 + cq * 4);
      T* out_b = reinterpret_cast<T*>(a.out) + (int64_t)b * a.Ho * a.Wo * Cexp + c0 + cq * 4;
      for (int qd = qg; qd < NQ; qd += nqg) {
        const int py = qd / (TOW / 4), px = (qd - py * (TOW / 4)) * 4;
        float acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int c = 0; c < 4; c++) acc[j][c] = bias[c];
#pragma unroll 1
        for (int ky = 0; ky < KS; ky++) {
          f32x4 wr[KS];
#pragma unroll
          for (int kx = 0; kx < KS; kx++) wr[kx] = *reinterpret_cast<const f32x4*>(wdw_s + (ky * KS + kx) * Cexp + c0 + cq * 4);
          const int erow = ((py * S + ky) * PW + px * S) * EP + cq * 4;
#pragma unroll
          for (int hx = 0; hx < NX; hx++) {
            float ev[4];
            V::load4(e_s, erow + hx * EP, ev);
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
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < a.Ho) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (ox + j < a.Wo) {
              float v[4];
#pragma unroll
              for (int c = 0; c < 4; c++) v[c] = acc[j][c];
              swish_n<BF16, 4>(v);
#pragma unroll
              for (int c = 0; c < 4; c++) sum[ci][(j >> 1) * 4 + c] += v[c];      // [0..3]: the pair group 2 qg, [4..7]: 2 qg + 1
              V::store4(out_b, (int64_t)((oy * a.Wo + ox + j) * Cexp), v);
            }
          }
        }
      }
    }
    if (ci == 0) XSTAMP(9);
    __syncthreads();                                                   // e_s free for the next chunk / the next tile / the reduction
  }
  }   // tiles of this workgroup
  XSTAMP(10);

  // ---------------- P4: channel sums in a fixed order -> partial reduce-FC products of block i ----------------
  float (*red9)[9] = reinterpret_cast<float (*)[9]>(smem);            // [XT][9] (the tile region is dead)
  // the reduce-FC weights of this thread's hidden unit (32 lanes share unit j = thread / 32) are requested before the two
  // barriers of the channel-sum reduction: their L2 round trip runs under it instead of behind it (k_mbf.hip phase D, same move)
  constexpr int NPF = NT2C ? (NT2C * 16 + 31) / 32 : 6;
  const int lp = tid & 31, jp = tid >> 5;
  float wpf[NPF];
#pragma unroll
  for (int i = 0; i < NPF; i++) wpf[i] = a.se_wr[(uint32_t)(min(jp, a.sq2 - 1) * Cexp + min(lp + 32 * i, Cexp - 1))];
#pragma unroll
  for (int ci = 0; ci < XCH; ci++) {
    if (ci >= a.nchunks) continue;
    const int cgs = cgs_of[ci], c0 = ci * a.chunk_tiles * 16, cc = cgs * 8;
    const float cgs_inv = __builtin_amdgcn_rcpf((float)cgs);
    const int npg = udiv_f(XT, cgs, cgs_inv) & ~1;                    // (even: see P3)
    {
      // row (pixel-pair group pg, channel octet cg) of the two-pixel form = this thread's halves: quad group qg = pg / 2, channel quad cq = 2 cg + {0, 1}
      const int cqs = 2 * cgs, qg = udiv_f(tid, cqs, __builtin_amdgcn_rcpf((float)cqs)), cq = tid - qg * cqs;
      if (qg < (npg >> 1)) {
#pragma unroll
        for (int hp = 0; hp < 2; hp++)
#pragma unroll
          for (int c = 0; c < 4; c++) red9[(2 * qg + hp) * cgs + (cq >> 1)][(cq & 1) * 4 + c] = sum[ci][hp * 4 + c];
      }
    }
    __syncthreads();
    for (int c = tid; c < cc; c += XT) {
      const int cg = c >> 3, cl = c & 7;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int q = 0;
      for (; q + 3 < npg; q += 4) {
        s0 += red9[cg + q * cgs][cl]; s1 += red9[cg + (q + 1) * cgs][cl]; s2 += red9[cg + (q + 2) * cgs][cl]; s3 += red9[cg + (q + 3) * cgs][cl];
      }
      for (; q < npg; q++) s0 += red9[cg + q * cgs][cl];
      csum_s[c0 + c] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
  }
  {
    float* hrow = a.hpart_out + ((int64_t)b * wgs + wgt) * a.sqp2;
    for (int j = jp; j < ((a.sq2 + 15) & ~15); j += XT / 32) {
      float dot = 0.f;
      if (j < a.sq2) {
        if (j == jp) {                     // (same products in the same order as the loop below)
#pragma unroll
          for (int i = 0; i < NPF; i++) if (lp + 32 * i < Cexp) dot = fmaf(wpf[i], csum_s[lp + 32 * i], dot);
          for (int c = lp + 32 * NPF; c < Cexp; c += 32) dot = fmaf(a.se_wr[(uint32_t)(j * Cexp + c)], csum_s[c], dot);
        } else {
          for (int c = lp; c < Cexp; c += 32) dot = fmaf(a.se_wr[(uint32_t)(j * Cexp + c)], csum_s[c], dot);
        }
      }
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) dot += __shfl_xor(dot, off, 64);
      if (lp == 0 && j < a.sq2) hrow[j] = dot;
    }
  }
#ifdef HEP_XBF_TRACE
  XSTAMP(11);
  if (g_xbf_trace && a.trace && lane == 0) {
    unsigned long long* o = g_xbf_trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * XW + wave) * 12;
    for (int i = 0; i < 12; i++) o[i] = stamps[i];
  }
#endif
}

#ifdef HEP_XBF_TRACE
extern "C" int hep_dbg_xbf_trace(unsigned long long* host, int max_waves, int enable) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)1 << 21;
  if (!buf) { if (hipMalloc((void**)&buf, cap * 8) != hipSuccess) return -1; hipMemset(buf, 0, cap * 8); }
  unsigned long long* p = enable ? buf : nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_xbf_trace), &p, sizeof p);
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)max_waves * 96, hipMemcpyDeviceToHost); }
  return (int)(cap / 12);
}
#endif

// ---- host side ----
void xbf_tile(int k, int s, int* toh, int* tow) { (void)k; *toh = 8; *tow = s == 2 ? 8 : 16; }
int xbf_supports(int k, int s) { return (k == 3 || k == 5) && (s == 1 || s == 2); }

size_t xbf_layout(XbfArgs* a) {
  const size_t es = a->bf16 ? 2 : 4, pad = a->bf16 ? 8 : 4;
  xbf_tile(a->k, a->s, &a->toh, &a->tow);
  const size_t ph = (size_t)(a->toh - 1) * a->s + a->k, pw = (size_t)(a->tow - 1) * a->s + a->k, pin = ph * pw;
  auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t w1_bytes = al((size_t)a->NT1 * 16 * (a->K1 + pad) * es), w2_bytes = al((size_t)a->NT2 * 16 * (a->NT1 * 16 + pad) * es);
  const size_t f_bytes = al(((size_t)(a->k * a->k + 1) * a->Cexp + (size_t)a->NT2 * 16 + (size_t)a->NT1 * 16) * 4);
  const size_t misc = ((size_t)((a->K1 + 15) & ~15) + 16 + XW * 16 + a->Cexp) * 4;
  const size_t a_bytes = pin * a->K1 * es, red_bytes = (size_t)XT * 9 * 4;
  size_t best = 0; int best_nch = 0;
  // union region: the input tile (P0/P1), then the expanded chunk (P2/P3), then the channel-sum staging (P4)
  auto uni_of = [&](int nch) {
    const int ct = (a->NT2 + nch - 1) / nch;
    const size_t e_bytes = al(pin * ((size_t)ct * 16 + pad) * es);
    return al(std::max(std::max(a_bytes, e_bytes), red_bytes));
  };
  for (int nch = 1; nch <= XCH; nch++) {
    const size_t total = uni_of(nch) + w1_bytes + w2_bytes + f_bytes + misc;
    if (total > 160 * 1024) continue;
    // the fewest chunks that let two workgroups share a CU; otherwise the fewest chunks that fit at all
    if (!best || (best > 80 * 1024 && total <= 80 * 1024)) { best = total; best_nch = nch; }
  }
  if (!best) return 0;
  const int ct = (a->NT2 + best_nch - 1) / best_nch;
  a->chunk_tiles = ct; a->nchunks = (a->NT2 + ct - 1) / ct;
  a->off_w1 = (int)uni_of(best_nch); a->off_w2 = a->off_w1 + (int)w1_bytes;
  a->off_f = a->off_w2 + (int)w2_bytes; a->off_misc = a->off_f + (int)f_bytes;
  a->blob_bytes = a->off_misc - a->off_w1;
  a->lds_bytes = best;
  return best;
}

// specialised boundaries: (k, s, NT1, K1, NT2) of phi 0 (blocks 0|1, 1|2, 2|3) and phi 3 (1|2 .. 7|8); anything else runs
// the generic instantiation (K1C = NT2C = 0)
#define XBF_SPECS(X) \
  X(3, 2, 8, 8, 1, 32, 6) X(3, 1, 8, 16, 2, 96, 9) X(5, 2, 8, 8, 2, 144, 9) \
  X(3, 2, 8, 8, 2, 24, 9) X(3, 1, 8, 16, 2, 144, 12) X(3, 1, 8, 16, 2, 192, 12) X(5, 2, 8, 8, 2, 192, 12) X(5, 1, 8, 16, 3, 288, 18) X(3, 2, 8, 8, 3, 288, 18)

template <bool BF16, int KS, int S, int TOH, int TOW, int NT1, int K1C, int NT2C>
static int xprep1() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(xbf_kernel<BF16, KS, S, TOH, TOW, NT1, K1C, NT2C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : -1;
}
template <bool BF16, int KS, int S, int TOH, int TOW>
static int xprep() { return xprep1<BF16, KS, S, TOH, TOW, 1, 0, 0>() | xprep1<BF16, KS, S, TOH, TOW, 2, 0, 0>() | xprep1<BF16, KS, S, TOH, TOW, 3, 0, 0>(); }
int xbf_prepare(void) {
  int rc = xprep<true, 3, 1, 8, 16>() | xprep<true, 3, 2, 8, 8>() | xprep<true, 5, 1, 8, 16>() | xprep<true, 5, 2, 8, 8>() |
           xprep<false, 3, 1, 8, 16>() | xprep<false, 3, 2, 8, 8>() | xprep<false, 5, 1, 8, 16>() | xprep<false, 5, 2, 8, 8>();
#define X(k, s, th, tw, n1, k1, n2) rc |= xprep1<true, k, s, th, tw, n1, k1, n2>() | xprep1<false, k, s, th, tw, n1, k1, n2>();
  XBF_SPECS(X)
#undef X
  return rc;
}

template <bool BF16, int KS, int S, int TOH, int TOW>
static void launch_xbf_n(const XbfArgs& a, dim3 grid, hipStream_t s) {
  if (a.NT1 == 1) hipLaunchKernelGGL((xbf_kernel<BF16, KS, S, TOH, TOW, 1, 0, 0>), grid, dim3(XT), a.lds_bytes, s, a);
  else if (a.NT1 == 2) hipLaunchKernelGGL((xbf_kernel<BF16, KS, S, TOH, TOW, 2, 0, 0>), grid, dim3(XT), a.lds_bytes, s, a);
  else hipLaunchKernelGGL((xbf_kernel<BF16, KS, S, TOH, TOW, 3, 0, 0>), grid, dim3(XT), a.lds_bytes, s, a);
}
template <bool BF16>
static void launch_xbf_t(const XbfArgs& a, dim3 grid, hipStream_t s) {
  const bool generic = a.generic != 0;           // A/B switch, parity test of the generic path (decided when the plan is built)
#define X(k_, s_, th, tw, n1, k1, n2) \
  if (!generic && a.k == k_ && a.s == s_ && a.NT1 == n1 && a.K1 == k1 && a.NT2 == n2) { hipLaunchKernelGGL((xbf_kernel<BF16, k_, s_, th, tw, n1, k1, n2>), grid, dim3(XT), a.lds_bytes, s, a); return; }
  XBF_SPECS(X)
#undef X
  if (a.k == 3 && a.s == 1) launch_xbf_n<BF16, 3, 1, 8, 16>(a, grid, s);
  else if (a.k == 3 && a.s == 2) launch_xbf_n<BF16, 3, 2, 8, 8>(a, grid, s);
  else if (a.k == 5 && a.s == 1) launch_xbf_n<BF16, 5, 1, 8, 16>(a, grid, s);
  else launch_xbf_n<BF16, 5, 2, 8, 8>(a, grid, s);
}
int xbf_specialised(const XbfArgs& a) {
  if (a.generic) return 0;
#define X(k_, s_, th, tw, n1, k1, n2) if (a.k == k_ && a.s == s_ && a.NT1 == n1 && a.K1 == k1 && a.NT2 == n2) return 1;
  XBF_SPECS(X)
#undef X
  return 0;
}
void launch_xbf(const XbfArgs& a_, hipStream_t s) {
  XbfArgs a = a_;
#ifdef HEP_XBF_TRACE
  { static const char* sel = getenv("HEP_XBF_TRACE_SEL"); a.trace = !sel || a.Cexp == atoi(sel); }
#endif
  const int wgs = (a.tiles + a.tpw - 1) / a.tpw;                        // workgroups per image
  a.tiles_rcp = rcp_u32((uint32_t)wgs); a.tiles_x_rcp = rcp_u32((uint32_t)a.tiles_x);
  {
    const int es = a.bf16 ? 2 : 4, pw = (a.tow - 1) * a.s + a.k, rv = (pw * a.K1 * es) >> 4;
    a.sw_rcp = rcp_u32((uint32_t)((rv + 63) >> 6));
  }
  dim3 grid(wgs, a.B);
  if (a.bf16) launch_xbf_t<true>(a, grid, s); else launch_xbf_t<false>(a, grid, s);
}
