// se_finish.h - the second half of a squeeze-excite (reference efficientnet/model.py:92-97: _se_reduce bias, swish, _se_expand,
// sigmoid) from the partial reduce-FC rows a fused front left: one body for se_finish_kernel (k_dw.hip: a launch of its own)
// and for the tail of the fused front (k_mbf.hip: the last workgroup of an image to arrive runs it - no launch).
#pragma once
#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

// NTHR threads finish image b for channels [c0, c1).  The hidden vector is summed by the first 256 threads whatever NTHR is
// (G = 256 / sqp helper groups, fixed order): both callers produce the same bits.  SC1: the partial rows were written by other
// workgroups of the SAME launch with write-through stores - read them past this XCD's L2 (sc0 sc1 loads).
template <bool BF16, int NTHR, bool SC1>
__device__ __forceinline__ void se_finish_body(const SeFinishArgs& a, const int b, const int c0, const int c1, const int tid, float* se_sm) {
  typedef typename Vec8<BF16>::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
  constexpr int JV = BF16 ? 8 : 4, NV = BF16 ? 6 : 12;     // hidden units per 16-byte vector; vectors held per lane (sqp <= 48)
  float* hid_s = se_sm;                    // hidden [sqp] | helper-group row sums [G][sqp]
  float* red_s = se_sm + a.sqp;
  const int sqp = a.sqp, sq = a.sq;
  const int G = max(1, 256 / sqp);
  const int grp = tid / sqp, j = tid - grp * sqp;
  auto part = [&](int64_t i) -> float {
    if constexpr (SC1) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.hpart), 0, 0x7fffffff, 0x00020000), (int)(i * 4), 0, 17));
    else return a.hpart[i];
  };
  // Everything that does not depend on the hidden vector is requested FIRST: this lane's expand-FC weight row, its bias
  // and the reduce bias.  (No measurable effect on the launch: 4.6 us before and after - the smallest kernels of this
  // library all measure 3.3-5 us, which is the floor a launch costs on this GPU; kept because it is the shorter chain.)
  const int V = sqp / JV;
  const int k0 = c0 + tid;
  const raw_t* wrow = reinterpret_cast<const raw_t*>(reinterpret_cast<const T*>(a.we) + (int64_t)min(k0, a.C - 1) * sqp);
  raw_t wv[NV];
#pragma unroll
  for (int q = 0; q < NV; q++) wv[q] = wrow[min(q, V - 1)];
  const float bev = a.be[min(k0, a.C - 1)], brv = a.br[min(tid, sq - 1)];
  if (grp < G && j < sq) {
    const int64_t hp = (int64_t)b * a.rows * sqp + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int row = grp;
#pragma unroll 1
    for (; row + 3 * G < a.rows; row += 4 * G) {
      s0 += part(hp + (int64_t)row * sqp); s1 += part(hp + (int64_t)(row + G) * sqp); s2 += part(hp + (int64_t)(row + 2 * G) * sqp); s3 += part(hp + (int64_t)(row + 3 * G) * sqp);
    }
#pragma unroll 1
    for (; row < a.rows; row += G) s0 += part(hp + (int64_t)row * sqp);
    red_s[grp * sqp + j] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (tid < sqp) {
    float h = 0.f;
    if (tid < sq) {
      float sacc = 0.f;
      for (int q = 0; q < G; q++) sacc += red_s[q * sqp + tid];
      h = swishf(fmaf(sacc, a.inv_hw, brv));
    }
    hid_s[tid] = h;
  }
  __syncthreads();
  // (the same pairing of partial sums as the prologue of the project GEMM: even / odd hidden units)
  auto fma_vec = [&](const raw_t& w, int v, float& e0, float& e1) {
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(hid_s + v);
    if constexpr (BF16) {
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(hid_s + v + 4);
      e0 = fmaf(__uint_as_float(w[0] << 16), h0[0], e0); e1 = fmaf(__uint_as_float(w[0] & 0xffff0000u), h0[1], e1);
      e0 = fmaf(__uint_as_float(w[1] << 16), h0[2], e0); e1 = fmaf(__uint_as_float(w[1] & 0xffff0000u), h0[3], e1);
      e0 = fmaf(__uint_as_float(w[2] << 16), h1[0], e0); e1 = fmaf(__uint_as_float(w[2] & 0xffff0000u), h1[1], e1);
      e0 = fmaf(__uint_as_float(w[3] << 16), h1[2], e0); e1 = fmaf(__uint_as_float(w[3] & 0xffff0000u), h1[3], e1);
    } else {
      e0 = fmaf(w[0], h0[0], e0); e1 = fmaf(w[1], h0[1], e1); e0 = fmaf(w[2], h0[2], e0); e1 = fmaf(w[3], h0[3], e1);
    }
  };
  // a channel per thread and pass; all of a row's vectors are requested at once (the first pass's row was requested at the top).
  // Kept register-lean on purpose (one row in flight, the rare tail of a row longer than NV vectors not unrolled): the body is also
  // the tail of the 1024-thread fused fronts (128 registers per lane).
#pragma unroll 1
  for (int k = k0; k < c1; k += NTHR) {
    const raw_t* wr = reinterpret_cast<const raw_t*>(reinterpret_cast<const T*>(a.we) + (int64_t)k * sqp);
    if (k != k0) {
#pragma unroll
      for (int q = 0; q < NV; q++) wv[q] = wr[min(q, V - 1)];
    }
    float e0 = 0.f, e1 = 0.f;
#pragma unroll
    for (int q = 0; q < NV; q++) if (q < V) fma_vec(wv[q], q * JV, e0, e1);
#pragma unroll 1
    for (int q = NV; q < V; q++) fma_vec(wr[q], q * JV, e0, e1);
    a.scale[(int64_t)b * a.C + k] = sigmoidf((e0 + e1) + (k == k0 ? bev : a.be[k]));
  }
}
