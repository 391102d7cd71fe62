// k_pw_impl.h - pointwise (1x1) convolution as an MFMA GEMM on gfx950.
//
//   out[m, n] = act( sum_k (A[m,k] * se[b(m),k]) * W[n,k] + bias[n] ) (+ res[m,n])
//
// se[b,:] is the squeeze-excite scale of the project convs.  It is finished HERE, in the prologue of
// the GEMM that consumes it (reference efficientnet/model.py:86-93: `_se_reduce,_swish,_se_expand,
// sigmoid`): the depthwise kernels leave hpart[b][row][j], the partial products of the reduce FC with
// their channel sums (mean and FC are linear), and every workgroup adds the rows of its image, applies
// 1/HW, bias and swish, runs the expand FC for all K input channels into LDS and multiplies its A
// fragments by it.  No squeeze-excite launch, no scale tensor in HBM.
//
// A is the NHWC activation viewed as [M = B*H*W, K] row-major, W the folded conv weight
// [N, K] (K contiguous, exactly PyTorch's [Cout, Cin]).  Replaces the library calls behind
// `_expand_conv/_bn0/_swish`, `_project_conv/_bn2/+inputs` (reference efficientnet/model.py:78-81,
// 95-103) and the BiFPN lateral 1x1 convs (efficientdet/model.py:107-140).
//
// Mapping: the product is computed TRANSPOSED, D[n, m] = W . A^T, so that the MFMA "A operand"
// is a W fragment and the "B operand" an activation fragment; both are 16-byte loads of 8
// consecutive k straight from global memory in exactly the lane layout the instruction wants
// (lane l: row l&15, k = 8*(l>>4)+j), and each lane ends with 4 CONSECUTIVE output channels of
// one pixel -> one 8/16-byte store.  bf16: v_mfma_f32_16x16x32_bf16; fp32 (parity mode):
// v_mfma_f32_16x16x4_f32, an exact fp32 fma chain (4 k per issue, fed from the same 16-byte loads).
//
// These GEMMs are small and skinny (M = 1 Ki..256 Ki rows, K and N = 16..1152), i.e. latency- and
// HBM-bound, never MFMA-bound; what matters is enough workgroups and enough loads in flight.  A
// workgroup is 4 waves that the plan arranges per layer (PwArgs::mode):
//   mode 0  waves stacked along M   (big feature maps: each wave MT x NT tiles of 16x16)
//   mode 1  waves side by side in N (expand layers on small maps: one 16-row strip, 4*NT n-tiles)
//   mode 2  waves split K           (project layers on small maps: partial sums meet in LDS)
// The K loop is software-pipelined one step ahead (fragments of step k+1 are in flight while
// step k feeds the MFMAs).
#include "hep_dev.h"
#include "hep_internal.h"

#pragma once
#include <type_traits>
#ifndef HEP_PW_UF32
#define HEP_PW_UF32 2
#endif

template <bool BF16> struct Frag;
template <> struct Frag<true> { typedef u32x4 raw; static constexpr int KSTEP = 32, KLANE = 8; };
template <> struct Frag<false> { typedef f32x4 raw; static constexpr int KSTEP = 16, KLANE = 4; };

// PREC: 0 fp32 (exact-fp32 MFMA), 1 bf16, 2 fp8 - bf16 activations in memory, e4m3 operands in the MFMA: the weights
// are stored as e4m3 with one scale per output channel (folded behind the BN scale), the activation fragments are
// converted on the fly with one power-of-two scale per tensor (calibrated at hep_create), v_mfma_f32_16x16x32_fp8_fp8
// accumulates in fp32 and the epilogue multiplies by a_scale * w_scale[n].
template <int PREC, int MT, int NT>
struct Step {
  typename Frag<PREC != 0>::raw a[MT];                                                        // activation fragments
  typename std::conditional<PREC == 2, u32x2, typename Frag<PREC != 0>::raw>::type w[NT];      // weight fragments (fp8: 8 bytes)
};

// ACT (none | swish) is a template parameter: with a run-time activation the epilogue carried a scalar
// branch per output element
// SEV: squeeze-excite prologue variant - 0 none (expand / lateral convs), 1 one weight row x two vectors in flight
// per lane (K <= 256: the project convs on the big maps, which need their occupancy), 2 five rows x six vectors
// (deep K on the small maps), 3 the scale vector was finished by se_finish_kernel (k_dw.hip) and is only copied into LDS
// (blocks whose K x sq weight matrix is too large to be re-read by every workgroup: phi >= 3).  A template parameter
// because the rows in flight set the kernel's register count.
// profiling build (make trace): s_memrealtime (100 MHz) stamps of every wave of the selected launch (HEP_PW_TRACE_SEL="K,N")
#ifdef HEP_PW_TRACE
extern unsigned long long* g_pw_trace_host;     // device buffer (k_pw.hip); travels in PwArgs::trace_buf
#define PSTAMP(i) do { stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP(i)
#endif

// NWV: waves per workgroup.  4 everywhere except the split-K project convs of fp32 sessions (MODE 2, SEV 0 / 3), which run 8:
// their K loop is the exact-fp32 MFMAs (128 MAC per clock and CU: 3.8 us of a wave's 11.8 us at K = 1152) IN SERIES with the
// fragment loads - one wave per SIMD has nothing to overlap them with.  Eight K slices halve a wave's MFMA chain and put a
// second wave on every SIMD.  (bf16: measured slower in round 3, 11 us against 9 - its MFMAs are 16 x cheaper.)
// FRAG (round 5; bf16 split-K project convs behind a fused front, K a multiple of 32): BOTH operands arrive in MFMA fragment order -
// the weights host-packed [n-tile][k-step][lane][8], the activations stored that way by the front kernel (k_mbf.hip, MbfArgs::out_frag:
// [m-tile][k-step][lane][8]) - so every fragment load of a wave is ONE contiguous kilobyte instead of sixteen 64-byte row segments.
// One CU pulls contiguous fragments at 85-94 GB/s with the loads these kernels keep in flight, row segments at ~41 (tools/wstream,
// NOTEBOOK.md section 2): the K loop of the late project convs was bound by exactly that.  Same fragments, same order: bit-identical.
template <int PREC, int MT, int NT, int MODE, int ACT, int SEV, int NWV = 4, bool FRAG = false>
__global__ __launch_bounds__(NWV * 64) void pw_gemm_kernel(PwArgs a) {
  static_assert(NWV == 4 || (MODE == 2 && (SEV == 0 || SEV == 3)), "eight waves: split-K with the scale copied from se_finish_kernel only");
  constexpr int NTHR = NWV * 64;
  constexpr bool BF16 = PREC != 0, F8 = PREC == 2;
#ifdef HEP_PW_TRACE
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PSTAMP(0);
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef Frag<BF16> F;
  typedef typename F::raw raw_t;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: tile origin and K slice on the scalar unit)
  const int r = lane & 15, g = lane >> 4;
  const int K = a.K, M = a.M;
  const T* A = reinterpret_cast<const T*>(a.A);
  const T* W = reinterpret_cast<const T*>(a.W);
  const unsigned char* W8 = reinterpret_cast<const unsigned char*>(a.W);     // fp8: [N16][K] bytes
  const float inv_as = F8 ? 1.0f / a.a_scale : 1.0f;                          // a_scale is a power of two: exact

  // ---- block / wave -> tile assignment ----
  const int chunksN = a.chunksN;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);     // blocks sharing an A strip stay on one XCD
  const int mblk = udiv_rcp(logical, a.chunksN_rcp), nchunk = logical - mblk * chunksN;   // (launch_pw_prec made the reciprocal)
  int m0, ntile0, kbeg = 0, kend = K;
  if (MODE == 0) { m0 = (mblk * 4 + wave) * (16 * MT); ntile0 = nchunk * NT; }
  else if (MODE == 1) { m0 = mblk * (16 * MT); ntile0 = nchunk * 4 * NT + wave * NT; }
  else {
    m0 = mblk * (16 * MT); ntile0 = nchunk * NT;
    constexpr int UU = BF16 ? 2 : HEP_PW_UF32;
    const int steps = (K + F::KSTEP - 1) / F::KSTEP, per = ((steps + NWV - 1) / NWV + UU - 1) / UU * UU;   // slices start on whole load batches
    kbeg = min(K, wave * per * F::KSTEP); kend = min(K, (wave + 1) * per * F::KSTEP);
  }

  int mrow[MT]; bool mok[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) { mrow[i] = m0 + i * 16 + r; mok[i] = mrow[i] < M; }
  // Fragment loads are UNCONDITIONAL on clamped addresses: a load under a branch makes the compiler wait for every
  // outstanding load at the next use (vmcnt(0)), which turned the ring below back into one round trip per k-step.
  // Rows >= M and n-tiles >= tilesN read a valid row and their results are never stored; k beyond the wave's slice
  // re-reads the slice's last vector and the ACTIVATION fragment is zeroed (finite x 0), so W needs no select.
  const T* arow[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) arow[i] = A + (int64_t)min(mrow[i], M - 1) * K;
  const unsigned char* wrow[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int64_t row = (int64_t)(min(ntile0 + j, a.tilesN - 1) * 16 + r) * K;
    wrow[j] = F8 ? W8 + row : reinterpret_cast<const unsigned char*>(W + row);
  }
  const int klast = max(kend - F::KLANE, 0);
  // FRAG: fragment (tile, k-step) = 64 lanes x 16 bytes at ((tile * ksteps + k-step) * 64 + lane) * 16; K is padded to whole k-steps
  // (the weights' pad columns are zeros, the activations' are whatever the producer left: zeroed here)
  const int kst_all = (K + F::KSTEP - 1) / F::KSTEP;
  const raw_t* afr[MT]; const raw_t* wfr[NT];
  if constexpr (FRAG) {
    const int mtiles_all = M >> 4;
#pragma unroll
    for (int i = 0; i < MT; i++) afr[i] = reinterpret_cast<const raw_t*>(A) + ((size_t)min((m0 >> 4) + i, mtiles_all - 1) * kst_all) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NT; j++) wfr[j] = reinterpret_cast<const raw_t*>(W) + ((size_t)min(ntile0 + j, a.tilesN - 1) * kst_all) * 64 + lane;
  }
  auto load = [&](Step<PREC, MT, NT>& st, int kk) {
    if constexpr (FRAG) {
      const bool kok = kk < kend && kk + F::KLANE * g < K;          // (kk < kend is uniform: the slices are whole k-steps; the lane test only bites in K's last step)
      const int ksx = min(kk, max(kend - 1, 0)) / F::KSTEP;
#pragma unroll
      for (int i = 0; i < MT; i++) { const raw_t v = afr[i][ksx * 64]; st.a[i] = kok ? v : raw_t{}; }
#pragma unroll
      for (int j = 0; j < NT; j++) st.w[j] = wfr[j][ksx * 64];
      return;
    }
    const int k = kk + F::KLANE * g, kc = min(k, klast);
    const bool kok = k < kend;
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const raw_t v = *reinterpret_cast<const raw_t*>(arow[i] + kc);
      st.a[i] = kok ? v : raw_t{};
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if constexpr (F8) st.w[j] = *reinterpret_cast<const u32x2*>(wrow[j] + kc);
      else st.w[j] = *reinterpret_cast<const raw_t*>(wrow[j] + (size_t)kc * sizeof(T));
    }
  };
  // Modes 1 and 2 (small maps) take TWO k-steps per load batch: a lane's fragment is 16 bytes, the four lane groups of a
  // row cover 64 bytes, so one k-step touches half of a 128-byte line per row and the other half one step later - by
  // then the workgroup's own traffic (40 KB per step, 32 KB of L1) had evicted the line and every line came from L2
  // twice.  That doubled traffic, not latency, bounded these GEMMs: a workgroup moves its 100-200 KB at the ~64 GB/s
  // one CU gets out of L2, and a deeper ring changed nothing.  With both halves requested back to back the second one
  // merges with the miss in flight.  One batch stays in flight behind the one being multiplied (two slots); the first
  // does not depend on the squeeze-excite scale and is issued before its prologue.
  // fp32 sessions (HEP_PW_UF32): four k-steps per batch - the same 64 k per batch as bf16 - measured SLOWER than two (round 4:
  // 1152 -> 192 project conv 17.7 us against 16.2 us): the fp32 K loop is not twice the round trips of the bf16 one, it is twice
  // the bytes through one CU's L1 at the same ~40 GB/s plus the exact-fp32 MFMAs (128 MAC per clock and CU) in series with them
  constexpr int U = MODE == 0 ? 1 : (BF16 ? 2 : HEP_PW_UF32), KS2 = U * F::KSTEP;
  Step<PREC, MT, NT> st[2][U];
  auto load2 = [&](int slot, int kk) {
#pragma unroll
    for (int u = 0; u < U; u++) load(st[slot][u], kk + u * F::KSTEP);
  };
  load2(0, kbeg);
  PSTAMP(1);

  // ---- squeeze-excite prologue: scale_s[image - img0][k] for the images this workgroup's rows belong to ----
  // A chain of dependent round trips if written naively (hpart rows -> hidden -> weight rows -> scale), and it
  // sits in front of every project GEMM, so: the first batch of expand-FC weight rows (independent of the hidden
  // vector) is put in flight BEFORE the hpart rows are fetched and reduced, RB rows x VB 16-byte vectors per lane
  // at once (K = 1152, sq = 48 in bf16: everything in one batch, one round trip); the weights are in the session
  // dtype (bf16 sessions: half the bytes; the products are accumulated in fp32).
  extern __shared__ __attribute__((aligned(16))) float se_s[];
  constexpr bool SE = SEV != 0;
  const float hw_inv = SE ? __builtin_amdgcn_rcpf((float)a.HW) : 0.f;
  int img0 = 0;
  if constexpr (SEV == 3) {
    constexpr int ROWS = MODE == 0 ? 64 * MT : 16 * MT;
    const int mfirst = mblk * ROWS, mlast = min(M, mfirst + ROWS) - 1;
    img0 = udiv_f(mfirst, a.HW, hw_inv);
    const int nimg = udiv_f(mlast, a.HW, hw_inv) - img0 + 1;
    const float* sc_g = a.se_scale + (int64_t)img0 * K;
    const int nsc = nimg * K;                                             // K is a multiple of 8
    for (int i0 = threadIdx.x * 4; i0 < nsc; i0 += NTHR * 4 * 4) {        // four vectors in flight per lane: K <= 4096 is one round trip
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const f32x4*>(sc_g + min(i0 + q * NTHR * 4, nsc - 4));
#pragma unroll
      for (int q = 0; q < 4; q++) if (i0 + q * NTHR * 4 < nsc) *reinterpret_cast<f32x4*>(se_s + i0 + q * NTHR * 4) = v[q];
    }
    __syncthreads();
  } else if constexpr (SE) {
    constexpr int ROWS = MODE == 0 ? 64 * MT : 16 * MT;
    constexpr int JV = BF16 ? 8 : 4;              // hidden units per 16-byte weight vector
    constexpr int RB = SEV == 2 ? 5 : 1, VB = SEV == 2 ? 6 : 2;   // weight rows x vectors in flight per lane
    const int mfirst = mblk * ROWS, mlast = min(M, mfirst + ROWS) - 1;
    img0 = udiv_f(mfirst, a.HW, hw_inv);
    const int img1 = udiv_f(mlast, a.HW, hw_inv), sqp = a.sqp, sq = a.sq;
    const int V = sqp / JV, R = (K + 255) >> 8;
    const T* WE = reinterpret_cast<const T*>(a.se_we);
    float* hid_s = se_s + a.se_nimg * K;          // [sqp]
    float* red_s = hid_s + sqp;                   // [G][sqp] row sums of the G helper groups
    const int G = max(1, 256 / sqp);
    const int grp = threadIdx.x / sqp, j = threadIdx.x - grp * sqp;
    raw_t wv[RB][VB];
    auto issue = [&](int r0, int v0) {
#pragma unroll
      for (int rr = 0; rr < RB; rr++) {
        const int k = ((r0 + rr) << 8) + threadIdx.x;
#pragma unroll
        for (int vv = 0; vv < VB; vv++) {
          wv[rr][vv] = raw_t{};
          if (k < K && v0 + vv < V) wv[rr][vv] = *reinterpret_cast<const raw_t*>(WE + (int64_t)k * sqp + (v0 + vv) * JV);
        }
      }
    };
    issue(0, 0);
    // the two bias vectors as well (they sat behind the barriers: two more dependent round trips in the chain)
    const float br_v = a.se_br[min((int)threadIdx.x, sq - 1)];
    float be_v[RB];
#pragma unroll
    for (int rr = 0; rr < RB; rr++) be_v[rr] = a.se_be[min((rr << 8) + (int)threadIdx.x, K - 1)];
    for (int img = img0; img <= img1; img++) {
      // hidden[j] = swish(inv_hw * sum_rows hpart[img][row][j] + br[j]): G groups each add every G-th row,
      // then the groups are added up - fixed order
      if (grp < G && j < sq) {
        const float* hp = a.hpart + (int64_t)img * a.se_rows * sqp + j;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int row = grp;
        for (; row + 3 * G < a.se_rows; row += 4 * G) {
          s0 += hp[(int64_t)row * sqp]; s1 += hp[(int64_t)(row + G) * sqp]; s2 += hp[(int64_t)(row + 2 * G) * sqp]; s3 += hp[(int64_t)(row + 3 * G) * sqp];
        }
        for (; row < a.se_rows; row += G) s0 += hp[(int64_t)row * sqp];
        red_s[grp * sqp + j] = (s0 + s1) + (s2 + s3);
      }
      __syncthreads();
      if (threadIdx.x < sqp) {
        float h = 0.f;
        if (threadIdx.x < sq) {
          float sacc = 0.f;
          for (int q = 0; q < G; q++) sacc += red_s[q * sqp + threadIdx.x];
          h = swishf(fmaf(sacc, a.inv_hw, br_v));
        }
        hid_s[threadIdx.x] = h;                  // padding entries are exact zeros
      }
      __syncthreads();
      // expand FC + sigmoid: lane t owns the weight rows k = t, t + 256, ...
      float* sc = se_s + (img - img0) * K;
      for (int r0 = 0; r0 < R; r0 += RB) {
        float e[RB][2];
#pragma unroll
        for (int rr = 0; rr < RB; rr++) { e[rr][0] = 0.f; e[rr][1] = 0.f; }
        for (int v0 = 0; v0 < V; v0 += VB) {
          if (r0 | v0 | (img - img0)) issue(r0, v0);          // (the first batch is already in flight)
#pragma unroll
          for (int vv = 0; vv < VB; vv++) {
            if (v0 + vv < V) {
              const f32x4 h0 = *reinterpret_cast<const f32x4*>(hid_s + (v0 + vv) * JV);
              f32x4 h1 = h0;
              if constexpr (BF16) h1 = *reinterpret_cast<const f32x4*>(hid_s + (v0 + vv) * JV + 4);
#pragma unroll
              for (int rr = 0; rr < RB; rr++) {
                const raw_t w = wv[rr][vv];
                if constexpr (BF16) {
                  e[rr][0] = fmaf(__uint_as_float(w[0] << 16), h0[0], e[rr][0]); e[rr][1] = fmaf(__uint_as_float(w[0] & 0xffff0000u), h0[1], e[rr][1]);
                  e[rr][0] = fmaf(__uint_as_float(w[1] << 16), h0[2], e[rr][0]); e[rr][1] = fmaf(__uint_as_float(w[1] & 0xffff0000u), h0[3], e[rr][1]);
                  e[rr][0] = fmaf(__uint_as_float(w[2] << 16), h1[0], e[rr][0]); e[rr][1] = fmaf(__uint_as_float(w[2] & 0xffff0000u), h1[1], e[rr][1]);
                  e[rr][0] = fmaf(__uint_as_float(w[3] << 16), h1[2], e[rr][0]); e[rr][1] = fmaf(__uint_as_float(w[3] & 0xffff0000u), h1[3], e[rr][1]);
                } else {
                  e[rr][0] = fmaf(w[0], h0[0], e[rr][0]); e[rr][1] = fmaf(w[1], h0[1], e[rr][1]);
                  e[rr][0] = fmaf(w[2], h0[2], e[rr][0]); e[rr][1] = fmaf(w[3], h0[3], e[rr][1]);
                }
              }
            }
          }
        }
#pragma unroll
        for (int rr = 0; rr < RB; rr++) {
          const int k = ((r0 + rr) << 8) + threadIdx.x;
          if (k < K) sc[k] = sigmoidf((e[rr][0] + e[rr][1]) + (r0 == 0 ? be_v[rr] : a.se_be[k]));
        }
      }
      __syncthreads();
    }
  }

  PSTAMP(2);
  int mimg[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) mimg[i] = (SE && mok[i]) ? udiv_f(mrow[i], a.HW, hw_inv) - img0 : 0;
  // split-K workgroups are the latency chains: their bias and residual are fetched here, under the K loop
  const T* R = reinterpret_cast<const T*>(a.res);
  constexpr bool HOIST = MODE == 2;
  // (unconditional on clamped addresses, raw: converted in the epilogue; wave w owns n-tiles j = w and w + 4)
  typedef typename std::conditional<BF16, u32x2, f32x4>::type res_t;
  constexpr int NH = HOIST ? (NT + 3) / 4 : 1;
  f32x4 bias_h[NH]; res_t res_h[HOIST ? MT : 1][NH];
  if constexpr (HOIST) {
#pragma unroll
    for (int jj = 0; jj < NH; jj++) {
      const int n = min((ntile0 + wave + 4 * jj) * 16 + 4 * g, a.N - 4);                 // N is a multiple of 8
      bias_h[jj] = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
      for (int i = 0; i < MT; i++)
        res_h[i][jj] = *reinterpret_cast<const res_t*>((R ? R : A) + (R ? (int64_t)min(mrow[i], M - 1) * a.N + n : 0));
    }
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](Step<PREC, MT, NT>& st, int kk) {
    // squeeze-excite scales of this lane's k run (LDS; rows outside M or k >= kend hold zeros in A anyway)
    const int ks = min(kk + F::KLANE * g, K - F::KLANE);
    if constexpr (F8) {
      long bfrag[MT];
#pragma unroll
      for (int i = 0; i < MT; i++) {
        if (SE) {
          const f32x4* sp = reinterpret_cast<const f32x4*>(se_s + mimg[i] * K + ks);
          const f32x4 s0 = sp[0], s1 = sp[1];
          const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
          bfrag[i] = cvt_fp8x8(st.a[i], s, inv_as);
        } else bfrag[i] = cvt_fp8x8(st.a[i], nullptr, inv_as);
      }
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const long afrag = __builtin_bit_cast(long, st.w[j]);
#pragma unroll
        for (int i = 0; i < MT; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(afrag, bfrag[i], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (BF16) {
      bf16x8 bfrag[MT];
#pragma unroll
      for (int i = 0; i < MT; i++) {
        u32x4 raw = st.a[i];
        if (SE) {
          const f32x4* sp = reinterpret_cast<const f32x4*>(se_s + mimg[i] * K + ks);
          const f32x4 s0 = sp[0], s1 = sp[1];
          const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
          for (int q = 0; q < 4; q++)
            raw[q] = pack_bf16x2(__uint_as_float(raw[q] << 16) * s[2 * q], __uint_as_float(raw[q] & 0xffff0000u) * s[2 * q + 1]);
        }
        bfrag[i] = __builtin_bit_cast(bf16x8, raw);
      }
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const bf16x8 afrag = __builtin_bit_cast(bf16x8, st.w[j]);
#pragma unroll
        for (int i = 0; i < MT; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[i], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; i++) {
        f32x4 x = st.a[i];
        if (SE) x *= *reinterpret_cast<const f32x4*>(se_s + mimg[i] * K + ks);
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
          for (int q = 0; q < 4; q++)   // lane group g supplies k = kk+4g+q to both operands
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.w[j][q], x[q], acc[i][j], 0, 0, 0);
      }
    }
  };

  auto compute2 = [&](int slot, int kk) {
#pragma unroll
    for (int u = 0; u < U; u++) if (u == 0 || kk + u * F::KSTEP < kend) compute(st[slot][u], kk + u * F::KSTEP);   // (wave-uniform)
  };
  // steady state: both following batches exist, no condition anywhere near a load; the last one or two batches are
  // peeled so that no load is issued past the slice (a single-batch GEMM - K <= 32 on the big maps - loads once)
  int kk = kbeg;
  for (; kk + 2 * KS2 < kend; kk += 2 * KS2) {
    load2(1, kk + KS2); compute2(0, kk);
    load2(0, kk + 2 * KS2); compute2(1, kk + KS2);
  }
  if (kk + KS2 < kend) { load2(1, kk + KS2); compute2(0, kk); compute2(1, kk + KS2); }
  else if (kk < kend) compute2(0, kk);

  PSTAMP(3);
  if (MODE == 2) {   // meet the K-slices in LDS (fixed order); wave w < 4 finishes n-tiles j = w, w+4, ...
    __shared__ f32x4 red[NWV][MT][NT][64];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < NT; j++) red[wave][i][j][lane] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < NT; j++)
        if ((j & 3) == wave) {
          f32x4 sum = (red[0][i][j][lane] + red[1][i][j][lane]) + (red[2][i][j][lane] + red[3][i][j][lane]);
          if constexpr (NWV == 8) sum += (red[4][i][j][lane] + red[5][i][j][lane]) + (red[6][i][j][lane] + red[7][i][j][lane]);
          acc[i][j] = sum;
        }
  }

  PSTAMP(4);
  // ---- epilogue: lane holds n = nt*16 + 4g + {0..3} for pixel row m0 + 16 i + r ----
#pragma unroll
  for (int j = 0; j < NT; j++) {
    if (MODE == 2 && (j & 3) != wave) continue;
    const int n = (ntile0 + j) * 16 + 4 * g;
    if (n >= a.N) continue;
    f32x4 b;
    if constexpr (HOIST) b = bias_h[j >> 2]; else b = *reinterpret_cast<const f32x4*>(a.bias + n);
    f32x4 ws = (f32x4){1.f, 1.f, 1.f, 1.f};
    if constexpr (F8) ws = *reinterpret_cast<const f32x4*>(a.wscale + n) * a.a_scale;     // dequantisation: a_scale * w_scale[n]
#pragma unroll
    for (int i = 0; i < MT; i++) {
      if (!mok[i]) continue;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { const float x = F8 ? fmaf(acc[i][j][q], ws[q], b[q]) : acc[i][j][q] + b[q]; v[q] = ACT == ACT_SWISH ? swish_t<BF16>(x) : x; }
      const int64_t o = (int64_t)mrow[i] * a.N + n;
      if (R) {
        float rr[4];
        if constexpr (HOIST) {
          const res_t rv = res_h[i][j >> 2];
          if constexpr (BF16) { rr[0] = __uint_as_float(rv[0] << 16); rr[1] = __uint_as_float(rv[0] & 0xffff0000u); rr[2] = __uint_as_float(rv[1] << 16); rr[3] = __uint_as_float(rv[1] & 0xffff0000u); }
          else { rr[0] = rv[0]; rr[1] = rv[1]; rr[2] = rv[2]; rr[3] = rv[3]; }
        } else V::load4(R, o, rr);
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] += rr[q];
      }
      V::store4(a.out, o, v);
    }
  }
#ifdef HEP_PW_TRACE
  PSTAMP(5);
  if (a.trace_buf && lane == 0 && wave < 4) {
    unsigned long long* o = a.trace_buf + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 6; i++) o[i] = stamps[i];
  }
#endif
}

template <int PREC, int MT, int MODE>
static void launch_nt(const PwArgs& a, dim3 grid, hipStream_t s) {
  // dynamic LDS of the squeeze-excite prologue: scale [se_nimg][K] | hidden [sqp] | helper-group row sums [<= 256]
  const size_t lds = a.sq > 0 ? ((size_t)a.se_nimg * a.K + a.sqp + 256 + a.sqp) * sizeof(float) : 0;
  const int sev = pw_se_variant(a);
  if constexpr (PREC == 0 && MODE == 2) {
    if (a.nwv == 8 && a.act != ACT_SWISH && (sev == 0 || sev == 3) && a.NT <= 2 && !a.frag) {
      if (sev == 3) { if (a.NT == 2) hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, 2, MODE, ACT_NONE, 3, 8>), grid, dim3(512), lds, s, a); else hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, 1, MODE, ACT_NONE, 3, 8>), grid, dim3(512), lds, s, a); }
      else { if (a.NT == 2) hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, 2, MODE, ACT_NONE, 0, 8>), grid, dim3(512), 0, s, a); else hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, 1, MODE, ACT_NONE, 0, 8>), grid, dim3(512), 0, s, a); }
      return;
    }
  }
  if constexpr (PREC != 2 && MT == 2 && MODE == 2) {
    if (a.frag && a.act != ACT_SWISH && sev >= 1 && a.NT <= PW_FRAG_MAX_NT) {        // fragment-ordered operands (add_pw decides; hep_kernel_symbol names it)
      if (PREC == 0 && a.nwv == 8 && sev == 3 && a.NT <= 2) {
        if (a.NT == 2) hipLaunchKernelGGL((pw_gemm_kernel<PREC, 2, 2, 2, ACT_NONE, 3, 8, true>), grid, dim3(512), lds, s, a);
        else hipLaunchKernelGGL((pw_gemm_kernel<PREC, 2, 1, 2, ACT_NONE, 3, 8, true>), grid, dim3(512), lds, s, a);
        return;
      }
      switch (a.NT) {
#define FCASE(n) case n: if (sev == 1) hipLaunchKernelGGL((pw_gemm_kernel<PREC, 2, n, 2, ACT_NONE, 1, 4, true>), grid, dim3(256), lds, s, a); \
                 else if (sev == 2) hipLaunchKernelGGL((pw_gemm_kernel<PREC, 2, n, 2, ACT_NONE, 2, 4, true>), grid, dim3(256), lds, s, a); \
                 else hipLaunchKernelGGL((pw_gemm_kernel<PREC, 2, n, 2, ACT_NONE, 3, 4, true>), grid, dim3(256), lds, s, a); return;
        FCASE(1) FCASE(2) FCASE(3) FCASE(4)
#ifdef HEP_ALT
        FCASE(5) FCASE(6) FCASE(7) FCASE(8)      // (HEP_PW_WIDE: measured loss, NOTEBOOK.md section 11)
#endif
#undef FCASE
      }
    }
  }
  switch (a.NT) {
#define CASE(n) case n: if (a.act == ACT_SWISH) hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, n, MODE, ACT_SWISH, 0>), grid, dim3(256), 0, s, a); \
                else if (sev == 0) hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, n, MODE, ACT_NONE, 0>), grid, dim3(256), 0, s, a); \
                else if (sev == 1) hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, n, MODE, ACT_NONE, 1>), grid, dim3(256), lds, s, a); \
                else if (sev == 3) hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, n, MODE, ACT_NONE, 3>), grid, dim3(256), lds, s, a); \
                else hipLaunchKernelGGL((pw_gemm_kernel<PREC, MT, n, MODE, ACT_NONE, 2>), grid, dim3(256), lds, s, a); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
}

// one translation unit per precision (k_pw_f32.hip, k_pw_bf16.hip, k_pw_fp8.hip) instantiates this: the three sets of
// ~190 kernels compile in parallel
template <int PREC>
void launch_pw_prec(const PwArgs& a_, hipStream_t s) {
  const int per_block_n = a_.mode == 1 ? 4 * a_.NT : a_.NT;
  const int chunksN = (a_.tilesN + per_block_n - 1) / per_block_n;
  const int rows = a_.mode == 0 ? 64 * a_.MT : 16 * a_.MT;
  dim3 grid(((a_.M + rows - 1) / rows) * chunksN);
  PwArgs a = a_; a.chunksN = chunksN; a.chunksN_rcp = rcp_u32(chunksN);
#ifdef HEP_PW_TRACE
  { static const char* sel = getenv("HEP_PW_TRACE_SEL"); int k = 0, n = 0; if (sel) sscanf(sel, "%d,%d", &k, &n); a.trace_buf = sel && a.K == k && a.N == n ? g_pw_trace_host : nullptr; }
#endif
  if (a.mode == 0) { if (a.MT == 2) launch_nt<PREC, 2, 0>(a, grid, s); else launch_nt<PREC, 1, 0>(a, grid, s); }
  else if (a.mode == 1) { if (a.MT == 2) launch_nt<PREC, 2, 1>(a, grid, s); else launch_nt<PREC, 1, 1>(a, grid, s); }
  else { if (a.MT == 2) launch_nt<PREC, 2, 2>(a, grid, s); else launch_nt<PREC, 1, 2>(a, grid, s); }
}
