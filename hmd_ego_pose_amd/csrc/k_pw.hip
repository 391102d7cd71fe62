// k_pw.hip - pointwise (1x1) convolution as an MFMA GEMM on gfx950.
//
//   out[m, n] = act( sum_k (A[m,k] * se[b(m),k]) * W[n,k] + bias[n] ) (+ res[m,n])
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
// A wave owns MT x NT tiles of 16x16; a block is 4 waves stacked along M.
#include "hep_dev.h"
#include "hep_internal.h"

template <bool BF16, int MT, int NT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwArgs a) {
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int chunksN = (a.tilesN + NT - 1) / NT;
  const int nblocks = gridDim.x;
  const int logical = xcd_remap(blockIdx.x, nblocks);
  const int mblk = logical / chunksN, nchunk = logical % chunksN;   // blocks sharing an A tile stay on one XCD
  const int m0 = (mblk * 4 + wave) * (16 * MT);
  const int ntile0 = nchunk * NT;
  const T* A = reinterpret_cast<const T*>(a.A);
  const T* W = reinterpret_cast<const T*>(a.W);
  const int K = a.K, M = a.M;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int mrow[MT]; bool mok[MT]; int mimg[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) {
    mrow[i] = m0 + i * 16 + r; mok[i] = mrow[i] < M;
    mimg[i] = a.se ? (mok[i] ? mrow[i] / a.HW : 0) : 0;
  }

  if constexpr (BF16) {
    for (int kk = 0; kk < K; kk += 32) {
      const int k = kk + 8 * g;
      const bool kok = k < K;
      bf16x8 bfrag[MT];
#pragma unroll
      for (int i = 0; i < MT; i++) {
        u32x4 raw = (u32x4){0, 0, 0, 0};
        if (kok && mok[i]) raw = *reinterpret_cast<const u32x4*>(A + (int64_t)mrow[i] * K + k);
        if (a.se && kok && mok[i]) {
          const f32x4* sp = reinterpret_cast<const f32x4*>(a.se + (int64_t)mimg[i] * K + k);
          f32x4 s0 = sp[0], s1 = sp[1];
          float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
          for (int q = 0; q < 4; q++) {
            float lo = __uint_as_float(raw[q] << 16) * s[2 * q];
            float hi = __uint_as_float(raw[q] & 0xffff0000u) * s[2 * q + 1];
            raw[q] = pack_bf16x2(lo, hi);
          }
        }
        bfrag[i] = __builtin_bit_cast(bf16x8, raw);
      }
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int nt = ntile0 + j;
        if (nt < a.tilesN) {
          u32x4 raw = (u32x4){0, 0, 0, 0};
          if (kok) raw = *reinterpret_cast<const u32x4*>(W + (int64_t)(nt * 16 + r) * K + k);
          bf16x8 afrag = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
          for (int i = 0; i < MT; i++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[i], acc[i][j], 0, 0, 0);
        }
      }
    }
  } else {
    for (int kk = 0; kk < K; kk += 16) {
      const int k = kk + 4 * g;
      const bool kok = k < K;
      f32x4 bfrag[MT];
#pragma unroll
      for (int i = 0; i < MT; i++) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kok && mok[i]) v = *reinterpret_cast<const f32x4*>(A + (int64_t)mrow[i] * K + k);
        if (a.se && kok && mok[i]) v *= *reinterpret_cast<const f32x4*>(a.se + (int64_t)mimg[i] * K + k);
        bfrag[i] = v;
      }
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int nt = ntile0 + j;
        if (nt < a.tilesN) {
          f32x4 w = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (kok) w = *reinterpret_cast<const f32x4*>(W + (int64_t)(nt * 16 + r) * K + k);
#pragma unroll
          for (int i = 0; i < MT; i++)
#pragma unroll
            for (int q = 0; q < 4; q++)   // lane group g supplies k = kk+4g+q to both operands
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q], bfrag[i][q], acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: lane holds n = nt*16 + 4g + {0..3} for pixel row m = ... + r
  const T* R = reinterpret_cast<const T*>(a.res);
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int n = (ntile0 + j) * 16 + 4 * g;
    if (n >= a.N) continue;
    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
    for (int i = 0; i < MT; i++) {
      if (!mok[i]) continue;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = apply_act(acc[i][j][q] + b[q], a.act);
      const int64_t o = (int64_t)mrow[i] * a.N + n;
      if (R) {
        float rr[4]; V::load4(R, o, rr);
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] += rr[q];
      }
      V::store4(a.out, o, v);
    }
  }
}

template <bool BF16, int MT>
static void launch_nt(const PwArgs& a, dim3 grid, hipStream_t s) {
  switch (a.NT) {
#define CASE(n) case n: hipLaunchKernelGGL((pw_gemm_kernel<BF16, MT, n>), grid, dim3(256), 0, s, a); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
}

void launch_pw(const PwArgs& a, hipStream_t s) {
  const int chunksN = (a.tilesN + a.NT - 1) / a.NT;
  const int mblocks = (a.M + 64 * a.MT - 1) / (64 * a.MT);
  dim3 grid(mblocks * chunksN);
  if (a.bf16) { if (a.MT == 2) launch_nt<true, 2>(a, grid, s); else launch_nt<true, 1>(a, grid, s); }
  else        { if (a.MT == 2) launch_nt<false, 2>(a, grid, s); else launch_nt<false, 1>(a, grid, s); }
}
