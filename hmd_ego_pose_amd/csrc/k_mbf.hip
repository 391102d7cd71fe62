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
//            LDS [PIN][CC]; pixels outside the image are written as ZERO (the depthwise pads the
//            *activated* map, reference utils_extra.py:33-44)
//   phase C  depthwise taps from LDS (weights in LDS) -> +bias, swish -> global, 16 bytes per lane;
//            per-lane channel sums for squeeze-excite
//   phase D  deterministic LDS reduction of the sums -> partial[b][tile][c]   (no atomics)
// Blocks without an expand conv (first block of the net) skip phase B: the input tile IS the
// depthwise input.
#include <stdlib.h>

#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

#define MBF_THREADS 512
#define MBF_WAVES 8

template <bool BF16, int KS, int S>
__global__ __launch_bounds__(MBF_THREADS) void mbf_kernel(MbfArgs a) {
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
  constexpr int KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4, PAD = BF16 ? 8 : 4;
  constexpr int TS = 8;                              // output tile side
  constexpr int PW = (TS - 1) * S + KS;              // input tile side
  constexpr int PIN = PW * PW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int chunks = (a.Cexp + a.CC - 1) / a.CC;
  const int tile = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int b = blockIdx.y;
  const int tiles_x = (a.Wo + TS - 1) / TS;
  const int oy0 = (tile / tiles_x) * TS, ox0 = (tile % tiles_x) * TS;
  const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;     // input coords of tile pixel (0,0)
  const int c0 = chunk * a.CC;                                     // first expanded channel of this block
  const int cc = min(a.CC, a.Cexp - c0);                           // channels of this block (multiple of 8)
  const int K = a.Cin;
  const int KP = K + PAD, EP = a.CC + PAD;                         // LDS row pitches (elements)
  T* a_s = reinterpret_cast<T*>(smem);                             // [PIN][KP]   input tile
  T* e_s = reinterpret_cast<T*>(smem + a.off_e);                   // [PIN][EP]   expanded (activated) tile
  T* w_s = reinterpret_cast<T*>(smem + a.off_we);                  // [CC][KP]    expand-weight chunk
  float* wdw_s = reinterpret_cast<float*>(smem + a.off_w);         // [KS*KS][CC]
  float* be_s = wdw_s + KS * KS * a.CC;                            // [CC] expand bias
  float* bdw_s = be_s + a.CC;                                      // [CC] depthwise bias
  float* red = reinterpret_cast<float*>(smem);                     // phase D scratch over the dead input tile

  // ---- phase A: everything this workgroup needs, global -> LDS, all loads issued in batches of 8
  //      before the first LDS store (one memory round trip per batch): depthwise weights + biases,
  //      the expand-weight chunk [cc][K], and the input tile (zero outside the image) ----
  const int mtiles = (PIN + 15) / 16, ntiles = (cc + 15) / 16;
  const int ksteps = (K + KSTEP - 1) / KSTEP;
  const int npairs = a.has_expand ? mtiles * ntiles : 0;           // pair = nt * mtiles + mt
  for (int i = threadIdx.x; i < KS * KS * cc; i += MBF_THREADS) {
    const int tap = i / cc, c = i % cc;
    wdw_s[tap * a.CC + c] = a.wdw[(int64_t)tap * a.Cexp + c0 + c];
  }
  for (int c = threadIdx.x; c < cc; c += MBF_THREADS) { be_s[c] = a.has_expand ? a.be[c0 + c] : 0.f; bdw_s[c] = a.bdw[c0 + c]; }
  {
    constexpr int NB = BF16 ? 8 : 4;                               // 16-byte vectors (bf16) / 32-byte pairs (fp32) in flight per lane
    const int kv = K >> 3;                                         // 8-channel vectors per input pixel / weight row
    const int vecs = a.has_expand ? kv : cc >> 3;
    const int cbase = a.has_expand ? 0 : c0;
    T* dst = a.has_expand ? a_s : e_s;
    const int pitch = a.has_expand ? KP : EP;
    const int64_t img = (int64_t)b * a.H * a.W * K;
    const int n_in = PIN * vecs, n_w = a.has_expand ? ntiles * 16 * kv : 0;
    const T* Wg = reinterpret_cast<const T*>(a.we) + (int64_t)c0 * K;
    for (int base = 0; base < ((a.dbg_skip & 1) ? 0 : n_in + n_w); base += MBF_THREADS * NB) {
      raw_t x0[NB], x1[NB];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const int item = base + j * MBF_THREADS + threadIdx.x;
        x0[j] = raw_t{}; x1[j] = raw_t{};
        if (item < n_w) {                                          // expand-weight rows c0 .. c0 + 16*ntiles
          const T* src = Wg + (int64_t)(item / kv) * K + (item % kv) * 8;
          x0[j] = *reinterpret_cast<const raw_t*>(src);
          if (!BF16) x1[j] = *reinterpret_cast<const raw_t*>(src + 4);
        } else if (item < n_w + n_in) {
          const int it = item - n_w, p = it / vecs, v = it % vecs;
          const int gy = iy0 + p / PW, gx = ix0 + p % PW;
          if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const T* src = reinterpret_cast<const T*>(a.in) + img + ((int64_t)gy * a.W + gx) * K + cbase + v * 8;
            x0[j] = *reinterpret_cast<const raw_t*>(src);
            if (!BF16) x1[j] = *reinterpret_cast<const raw_t*>(src + 4);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const int item = base + j * MBF_THREADS + threadIdx.x;
        raw_t* d = nullptr;
        if (item < n_w) d = reinterpret_cast<raw_t*>(w_s + (int64_t)(item / kv) * KP + (item % kv) * 8);
        else if (item < n_w + n_in) { const int it = item - n_w; d = reinterpret_cast<raw_t*>(dst + (int64_t)(it / vecs) * pitch + (it % vecs) * 8); }
        if (d) { d[0] = x0[j]; if (!BF16) d[1] = x1[j]; }
      }
    }
  }
  __syncthreads();

  // ---- phase B: expand 1x1 + bias + swish -> e_s ----
  if (a.has_expand) {
    const int my_pairs = npairs > wave ? (npairs - wave + MBF_WAVES - 1) / MBF_WAVES : 0;
    const int my_items = my_pairs * ksteps;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto step = [&](int it, raw_t wfrag) {
      const int pair = wave + MBF_WAVES * (it / ksteps), ks = it % ksteps;
      const int mt = pair % mtiles, nt = pair / mtiles;
      const int m = mt * 16 + r;
      const int k = ks * KSTEP + KLANE * g;
      raw_t xa = {};
      if (k < K && m < PIN) xa = *reinterpret_cast<const raw_t*>(a_s + (int64_t)m * KP + k);
      if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[q], xa[q], acc, 0, 0, 0);
      }
      if (ks != ksteps - 1) return;
      const int n = nt * 16 + 4 * g;          // lane: 4 consecutive expanded channels of tile pixel m
      if (m < PIN && n < cc) {
        const int gy = iy0 + m / PW, gx = ix0 + m % PW;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(be_s + n);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = inside ? swish_t<BF16>(acc[q] + bias[q]) : 0.f;
        V::store4(e_s, (int64_t)m * EP + n, v);
      }
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    for (int it = 0; it < ((a.dbg_skip & 2) ? 0 : my_items); it++) {
      const int pair = wave + MBF_WAVES * (it / ksteps), ks = it % ksteps;
      const int k = ks * KSTEP + KLANE * g;
      raw_t wf = {};
      if (k < K) wf = *reinterpret_cast<const raw_t*>(w_s + (int64_t)((pair / mtiles) * 16 + r) * KP + k);
      step(it, wf);
    }
    __syncthreads();
  }

  // ---- phase C: depthwise taps from LDS -> global, SE sums ----
  // lane -> (channel group, pixel slot): the channel group of a lane is FIXED (cgs rounded up to a
  // power of two divides the workgroup), so its squeeze-excite sums stay in registers
  const int cgs = cc >> 3;
  int cgp = 1; while (cgp < cgs) cgp <<= 1;
  const int cg = threadIdx.x & (cgp - 1), pslot = threadIdx.x / cgp, pstride = MBF_THREADS / cgp;
  float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cg < cgs) {
    float bias[8];
    {
      const f32x4* bp = reinterpret_cast<const f32x4*>(bdw_s + cg * 8);
      const f32x4 b0 = bp[0], b1 = bp[1];
#pragma unroll
      for (int c = 0; c < 4; c++) { bias[c] = b0[c]; bias[4 + c] = b1[c]; }
    }
    for (int p = pslot; p < ((a.dbg_skip & 4) ? 0 : TS * TS); p += pstride) {
      const int py = p / TS, px = p % TS;
      const int oy = oy0 + py, ox = ox0 + px;
      float acc[8];
#pragma unroll
      for (int c = 0; c < 8; c++) acc[c] = bias[c];
#pragma unroll
      for (int ky = 0; ky < KS; ky++)
#pragma unroll
        for (int kx = 0; kx < KS; kx++) {
          float ev[8];
          V::load(e_s, (int64_t)((py * S + ky) * PW + px * S + kx) * EP + cg * 8, ev);
          const f32x4* wp = reinterpret_cast<const f32x4*>(wdw_s + (ky * KS + kx) * a.CC + cg * 8);
          const f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
          for (int c = 0; c < 4; c++) { acc[c] = fmaf(ev[c], w0[c], acc[c]); acc[4 + c] = fmaf(ev[4 + c], w1[c], acc[4 + c]); }
        }
      if (oy < a.Ho && ox < a.Wo) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) { v[c] = swish_t<BF16>(acc[c]); sum[c] += v[c]; }
        V::store(a.out, (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.Cexp + c0 + cg * 8, v);
      }
    }
  }
  __syncthreads();      // a_s is dead: reuse as reduction scratch [MBF_THREADS][9]

  // ---- phase D: per-block channel sums, fixed order ----
  if (a.partial && !(a.dbg_skip & 8)) {
#pragma unroll
    for (int c = 0; c < 8; c++) red[threadIdx.x * 9 + c] = sum[c];
    __syncthreads();
    const int tiles = tiles_x * ((a.Ho + TS - 1) / TS);
    for (int o = threadIdx.x; o < cc; o += MBF_THREADS) {
      const int ocg = o >> 3, oc = o & 7;
      float s = 0.f;
      for (int t = ocg; t < MBF_THREADS; t += cgp) s += red[t * 9 + oc];     // lane t owns channel group t % cgp
      a.partial[((int64_t)b * tiles + tile) * a.Cexp + c0 + o] = s;
    }
  }
}

size_t mbf_lds_layout(int Cin, int CC, int k, int s, int bf16, int has_expand, MbfArgs* a) {
  const size_t es = bf16 ? 2 : 4, pad = bf16 ? 8 : 4;
  const size_t pw = (size_t)(8 - 1) * s + k, pin = pw * pw;
  size_t in_bytes = has_expand ? pin * (Cin + pad) * es : 0;
  in_bytes = std::max(in_bytes, (size_t)MBF_THREADS * 9 * 4);            // phase D scratch lives there too
  in_bytes = (in_bytes + 15) & ~(size_t)15;
  const size_t e_bytes = (pin * (CC + pad) * es + 15) & ~(size_t)15;
  const size_t w_bytes = (size_t)(k * k + 2) * CC * 4;                    // depthwise weights + the two bias vectors
  const size_t we_bytes = has_expand ? (((size_t)CC * (Cin + pad) * es + 15) & ~(size_t)15) : 0;   // expand-weight chunk
  if (a) { a->off_e = in_bytes; a->off_we = in_bytes + e_bytes; a->off_w = a->off_we + we_bytes; a->lds_bytes = a->off_w + w_bytes; }
  return in_bytes + e_bytes + we_bytes + w_bytes;
}

template <bool BF16, int KS, int S>
static int prep_one() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(mbf_kernel<BF16, KS, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess ? 0 : -1;
}
int mbf_prepare(void) {
  return prep_one<true, 3, 1>() | prep_one<true, 3, 2>() | prep_one<true, 5, 1>() | prep_one<true, 5, 2>() |
         prep_one<false, 3, 1>() | prep_one<false, 3, 2>() | prep_one<false, 5, 1>() | prep_one<false, 5, 2>();
}

template <bool BF16>
static void launch_mbf_t(const MbfArgs& a, dim3 grid, hipStream_t s) {
  if (a.k == 3 && a.s == 1) hipLaunchKernelGGL((mbf_kernel<BF16, 3, 1>), grid, dim3(MBF_THREADS), a.lds_bytes, s, a);
  else if (a.k == 3 && a.s == 2) hipLaunchKernelGGL((mbf_kernel<BF16, 3, 2>), grid, dim3(MBF_THREADS), a.lds_bytes, s, a);
  else if (a.k == 5 && a.s == 1) hipLaunchKernelGGL((mbf_kernel<BF16, 5, 1>), grid, dim3(MBF_THREADS), a.lds_bytes, s, a);
  else hipLaunchKernelGGL((mbf_kernel<BF16, 5, 2>), grid, dim3(MBF_THREADS), a.lds_bytes, s, a);
}
void launch_mbf(const MbfArgs& a_, hipStream_t s) {
  MbfArgs a = a_;
  static const int skip = getenv("HEP_MBF_SKIP") ? atoi(getenv("HEP_MBF_SKIP")) : 0;   // timing experiments only (results are wrong)
  a.dbg_skip = skip;
  const int tiles = ((a.Wo + 7) / 8) * ((a.Ho + 7) / 8), chunks = (a.Cexp + a.CC - 1) / a.CC;
  dim3 grid(tiles * chunks, a.B);
  if (a.bf16) launch_mbf_t<true>(a, grid, s); else launch_mbf_t<false>(a, grid, s);
}
