// k_dw.hip - the HBM-bound side of the EfficientNet backbone on gfx950: stem conv, depthwise
// kxk conv (+BN+swish, + the squeeze-excite channel sums folded into partial reduce-FC products) and the
// zero-padded 3x3/2 max-pool.  NHWC activations, 8 channels (16 B in bf16) per lane so every
// wave-level access is a run of full 16-byte vectors along C.
#include <stdlib.h>

#include "hep_dev.h"
#include "hep_internal.h"
#include "se_finish.h"

// ------------------------------------------------------------------------------------------------
// stem: conv3x3 stride 2, TF-SAME pad (0,1) on even sizes, 3 -> Cout, folded BN + swish.
// Replaces `_conv_stem,_bn0,_swish` (reference efficientdet/model.py:437-439).  The caller's
// fp32 tensor is read through its own strides, so the NHWC-memory view that
// eval/common.py:397 produces needs no copy.
// ------------------------------------------------------------------------------------------------
// The conv is a [pixels x 27] x [27 x Cout] product and runs on the matrix cores: an m-tile is 16 consecutive output
// pixels of one row, k = (ky, kx, ci).  The B operand (activations) is gathered straight from the caller's tensor in the
// MFMA lane layout - lane (r, g) loads the taps k = 8g .. 8g+7 (bf16 sessions) / k = 4s + g (fp32) of pixel r: 8 / 7
// four-byte loads per m-tile, no LDS, no barrier - and the A operand (weights) sits in registers for the whole wave.
//   fp32 sessions  v_mfma_f32_16x16x4_f32, 7 k-steps: an exact fp32 fma chain in k order (the arithmetic of the VALU kernel
//                  this replaces: lane = pixel, 27 x 8 FMAs per channel group with scalar weight operands, 14-17 us)
//   bf16 sessions  split-bf16 operands (x = xh + xl, w = wh + wl; wh xh + wh xl + wl xh: 2^-16 relative, far below the bf16
//                  rounding of the output) on v_mfma_f32_16x16x32_bf16: ONE k-step holds all 27 taps
// Weight rows are permuted per pair of n-tiles so that a lane ends with 8 consecutive channels of its pixel: one 16-byte
// store (bf16) straight into the NHWC map.  The next m-tile's taps are in flight while the current one is multiplied.
namespace {
// MFMA row i (0..15) of n-tile nt -> output channel.  Full pairs of n-tiles (32 channels) interleave so that lane group g
// holds channels 32p + 8g .. + 7 (tile 2p: the first four, tile 2p+1: the last four); a trailing single tile is natural.
__device__ __forceinline__ int stem_chan(int nt, int i, int C) {
  const int p = nt >> 1;
  if (32 * p + 32 <= C) return 32 * p + 8 * (i >> 2) + 4 * (nt & 1) + (i & 3);
  return 16 * nt + i;
}
}  // namespace

template <bool BF16, int NT>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
  typedef typename Vec8<BF16>::elem T;
  constexpr int NK = BF16 ? 8 : 7;                       // taps per lane and m-tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int C = a.Cout;
  // ---- this lane's taps: k -> (ky, kx, ci) -> element offset in the caller's tensor (relative to the pixel's window) ----
  int kky[NK], kkx[NK]; int64_t koff[NK]; bool kok[NK];
#pragma unroll
  for (int j = 0; j < NK; j++) {
    const int k = BF16 ? 8 * g + j : 4 * j + g, kc = min(k, 26);
    const int tap = kc / 3, ci = kc - tap * 3;
    kky[j] = tap / 3; kkx[j] = tap - kky[j] * 3; kok[j] = k < 27;
    koff[j] = (int64_t)ci * a.sc;
  }
  // ---- weights (A operand) and bias, once per wave ----
  bf16x8 wh[NT], wl[NT]; float wf[NT][7];
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int n = stem_chan(nt, r, C);                   // channel of MFMA row r
    const bool nok = n < C;
    if constexpr (BF16) {
      u32x4 h, l;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float w2[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int k = 8 * g + 2 * q + e;
          const float wv_ = a.w[min(k, 26) * C + min(n, C - 1)];
          w2[e] = (nok && k < 27) ? wv_ : 0.f;
        }
        h[q] = pack_bf16x2(w2[0], w2[1]);
        l[q] = pack_bf16x2(w2[0] - __uint_as_float(h[q] << 16), w2[1] - __uint_as_float(h[q] & 0xffff0000u));
      }
      wh[nt] = __builtin_bit_cast(bf16x8, h); wl[nt] = __builtin_bit_cast(bf16x8, l);
    } else {
#pragma unroll
      for (int s_ = 0; s_ < 7; s_++) {
        const int k = 4 * s_ + g;
        const float wv_ = a.w[min(k, 26) * C + min(n, C - 1)];
        wf[nt][s_] = (nok && k < 27) ? wv_ : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { const int nq = stem_chan(nt, 4 * g + q, C); bias[nt][q] = a.bias[min(nq, C - 1)]; }
  }
  // ---- m-tiles of this wave ----
  const int TX = (a.Wo + 15) >> 4, total = a.B * a.Ho * TX;
  const int t0 = (blockIdx.x * 4 + wave) * a.mpw, t1 = min(total, t0 + a.mpw);
  float x[NK], xn[NK];
  auto decode = [&](int t, int* b, int* oy, int* ox) { const int row = udiv_rcp(t, a.tx_rcp); *ox = (t - row * TX) * 16 + r; *b = udiv_rcp(row, a.ho_rcp); *oy = row - *b * a.Ho; };
  auto load = [&](int t, float* dst) {
    int b, oy, ox; decode(t, &b, &oy, &ox);
    const float* img = a.in + (int64_t)b * a.sn;
#pragma unroll
    for (int j = 0; j < NK; j++) {
      const int iy = oy * 2 - a.pad_t + kky[j], ix = ox * 2 - a.pad_l + kkx[j];
      const bool ok = kok[j] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float v = img[koff[j] + (int64_t)min(max(iy, 0), a.H - 1) * a.sh + (int64_t)min(max(ix, 0), a.W - 1) * a.sw];
      dst[j] = ok ? v : 0.f;
    }
  };
  if (t0 < t1) load(t0, x);
  for (int t = t0; t < t1; t++) {
    if (t + 1 < t1) load(t + 1, xn);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[nt] = bias[nt];
    if constexpr (BF16) {
      u32x4 xh_, xl_;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        xh_[q] = pack_bf16x2(x[2 * q], x[2 * q + 1]);
        xl_[q] = pack_bf16x2(x[2 * q] - __uint_as_float(xh_[q] << 16), x[2 * q + 1] - __uint_as_float(xh_[q] & 0xffff0000u));
      }
      const bf16x8 xh = __builtin_bit_cast(bf16x8, xh_), xl = __builtin_bit_cast(bf16x8, xl_);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nt], xh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nt], xl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nt], xh, acc[nt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int s_ = 0; s_ < 7; s_++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][s_], x[s_], acc[nt], 0, 0, 0);
    }
    int b, oy, ox; decode(t, &b, &oy, &ox);
    if (ox < a.Wo) {
      T* o = reinterpret_cast<T*>(a.out) + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * C;
#pragma unroll
      for (int nt = 0; nt < NT; nt += 2) {
        if (nt + 1 < NT && 32 * (nt >> 1) + 32 <= C) {            // a full pair: 8 consecutive channels per lane
          float v[8];
#pragma unroll
          for (int q = 0; q < 4; q++) { v[q] = acc[nt][q]; v[4 + q] = acc[nt + 1][q]; }
          swish_n<BF16, 8>(v);
          Vec8<BF16>::store(o, 32 * (nt >> 1) + 8 * g, v);
        } else {
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (nt + h < NT) {
              const int n = 16 * (nt + h) + 4 * g;
              if (n < C) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = acc[nt + h][q];
                swish_n<BF16, 4>(v);
                Vec8<BF16>::store4(o, n, v);
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NK; j++) x[j] = xn[j];
  }
}

// ---- the VALU form (kept: faster than the MFMA form on 32-channel stems, see launch_stem) ----
// (round 4: a two-pixels-per-lane form for W-contiguous inputs - the five input columns of a pixel pair as one aligned 16-byte
//  load + one 4-byte load per row and channel, 18 load instructions per pair instead of 54 - measured SLOWER: 18.7 us against
//  15.9 us in bf16, 34 us against 19 us in fp32 at phi 0 b16; 114-126 registers = four waves per SIMD instead of seven, and the
//  kernel is not short of load slots: 29 MB in 16 us with ~5 us of fma / swish issue per SIMD is latency hiding, i.e. waves)
// One lane = one output pixel: its 27 inputs are loaded once (zero outside the image, branch-free)
// and reused for every group of 8 output channels; the weights of a group are the same for the whole
// wave, so they arrive through the scalar cache and feed the FMAs as scalar operands.
template <bool BF16>
__global__ __launch_bounds__(256) void stem_valu_kernel(StemArgs a) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (ox >= a.Wo || oy >= a.Ho) return;
  float x[27];
  const float* img = a.in + (int64_t)b * a.sn;
#pragma unroll
  for (int ky = 0; ky < 3; ky++) {
    const int iy = oy * 2 - a.pad_t + ky;
    const int cy = min(max(iy, 0), a.H - 1);
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
      const int ix = ox * 2 - a.pad_l + kx;
      const int cx = min(max(ix, 0), a.W - 1);
      const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
      for (int ci = 0; ci < 3; ci++) {
        const float v = img[ci * a.sc + cy * a.sh + cx * a.sw];
        x[(ky * 3 + kx) * 3 + ci] = ok ? v : 0.f;
      }
    }
  }
  const int64_t pix = ((int64_t)b * a.Ho + oy) * a.Wo + ox;
#pragma unroll 1
  for (int cg = 0; cg < (a.Cout >> 3); cg++) {
    // uniform addresses in read-only memory: the constant address space makes them scalar loads
    // (the compiler cannot prove that the stores below do not alias them)
    typedef const __attribute__((address_space(4))) float* cptr;
    cptr w = (cptr)a.w + cg * 8;
    cptr bias = (cptr)a.bias + cg * 8;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = bias[c];
#pragma unroll
    for (int t = 0; t < 27; t++) {
#pragma unroll
      for (int c = 0; c < 8; c++) acc[c] = fmaf(x[t], w[t * a.Cout + c], acc[c]);
    }
    swish_n<BF16, 8>(acc);
    Vec8<BF16>::store(a.out, pix * a.Cout + cg * 8, acc);
  }
}

// (read at every call - plan naming and launch - never cached: a session created after the environment changed gets the plan it asked for;
//  stems wider than 64 channels have no MFMA instantiation and keep the VALU form)
int stem_uses_mfma(int cout, int force) { return cout <= 64 && (force >= 0 ? force != 0 : cout > 32); }      // force: Knobs::stem_mfma (-1 = by width)

void launch_stem(const StemArgs& a_, hipStream_t s) {
  StemArgs a = a_;
  // Measured (b16 / b8, stand-alone): phi 0 (32 channels) bf16 MFMA 18.5-19.7 us, VALU 14.5 us; fp32 22.2 / 20.2 us;
  // phi 3 (40 channels, 512 x 512) MFMA 34.1 us, VALU 46.8 us.  The MFMA form's four-byte gathers cost what its matrix
  // pipe saves on the narrow stem; it wins once the VALU form needs a fifth channel group.  HEP_STEM_MFMA=0|1 overrides.
  if (!a.mfma) {
    dim3 grid((unsigned)((a.Wo + 63) / 64), (unsigned)((a.Ho + 3) / 4), (unsigned)a.B);
    if (a.bf16) hipLaunchKernelGGL(stem_valu_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(stem_valu_kernel<false>, grid, dim3(256), 0, s, a);
    return;
  }
  const int TX = (a.Wo + 15) >> 4, total = a.B * a.Ho * TX;
  a.mpw = 2;                     // m-tiles per wave (measured 1 / 2 / 3 / 4 / 8: 21.2 / 18.5 / 21.4 / 19.7 / 26.1 us at phi 0)
  a.tx_rcp = rcp_u32((uint32_t)TX); a.ho_rcp = rcp_u32((uint32_t)a.Ho);
  dim3 grid((unsigned)((total + 4 * a.mpw - 1) / (4 * a.mpw)));
  const int nt = (a.Cout + 15) >> 4;
#define STEM_CASE(N) case N: if (a.bf16) hipLaunchKernelGGL((stem_kernel<true, N>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((stem_kernel<false, N>), grid, dim3(256), 0, s, a); break;
  switch (nt) { STEM_CASE(1) STEM_CASE(2) STEM_CASE(3) STEM_CASE(4) default: break; }     // (stem widths are 32 .. 64)
#undef STEM_CASE
}

// ------------------------------------------------------------------------------------------------
// depthwise k x k, stride S, TF-SAME zero padding, folded BN + swish; replaces
// `_depthwise_conv,_bn1,_swish` and the spatial half of `adaptive_avg_pool2d`
// (reference efficientnet/model.py:83-89).
// A lane owns 8 channels of a strip of TW output pixels along W: every input vector it loads
// feeds up to K taps, and the K x K x 8 weights are loaded once per row of taps.
// Squeeze-excite: each lane sums its post-swish outputs, the block reduces them through LDS in a
// FIXED order and writes their partial reduce-FC products hpart[b][block][j] - no atomics, so the
// result is bit-reproducible.
// ------------------------------------------------------------------------------------------------
// exact x / d for x * d < 2^32 with m = floor(2^32 / d) + 1 (two VALU instructions instead of ~30)
__device__ __forceinline__ int fast_div(int x, uint32_t m) { return (int)__umulhi((uint32_t)x, m); }

template <bool BF16, int KS, int S, int TW>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a) {
  typedef Vec8<BF16> V;
  extern __shared__ __attribute__((aligned(16))) float dw_smem[];
  HEP_POISON(dw_smem, ((size_t)(KS * KS + 1) * a.C + 256 * 9 + a.C) * sizeof(float));      // (launch_dw_tw's size)
  float* w_s = dw_smem;                         // [KS*KS][C] weights, then [C] bias
  float (*red)[9] = reinterpret_cast<float (*)[9]>(dw_smem + (KS * KS + 1) * a.C);
  const int CG = a.C >> 3;
  const int SW = (a.Wo + TW - 1) / TW;
  const int nitems = a.Ho * SW * CG;
  // XCD-aware block order: consecutive blocks are vertical neighbours (a block is a run of strips of one or two output
  // rows) and every input row feeds KS / S output rows; dealt round-robin over the 8 XCDs, the neighbours fetched the
  // same rows into 3 different L2s (PMC: 2.4x the algorithmic reads on the 64x64 maps).  The remap gives each XCD a
  // contiguous band of rows (b2.dw 20.3 -> 18.5 us alone, 15.8 -> 14.0 with four streams).
  // (Tried and dropped: unconditional clamped tap loads - the compiler then hoists all 18 of them, 196 VGPRs, two waves
  //  per SIMD, 25 us.  This kernel lives on occupancy, not on the latency of one lane.)
  int blk, b;
  xcd_remap2(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, &blk, &b);
  const int item = blk * 256 + threadIdx.x;
  const bool valid = item < nitems;
  const int strip = valid ? fast_div(item, a.cg_magic) : 0;
  const int cg = valid ? item - strip * CG : 0;
  const int oy = fast_div(strip, a.sw_magic), ox0 = (strip - oy * SW) * TW;
  constexpr int WIN = (TW - 1) * S + KS;

  // every lane needs the K*K weight vectors of its channel group: staged in LDS once per workgroup
  // (as global loads they were half of all vector-memory instructions of the kernel); both requests leave before the
  // first store waits (they were two dependent round trips)
  {
    const int nwv = (KS * KS * a.C) >> 2, nbv = a.C >> 2;
    const f32x4* wg = reinterpret_cast<const f32x4*>(a.w);
    const f32x4 v0 = wg[min((int)threadIdx.x, nwv - 1)], v1 = wg[min((int)threadIdx.x + 256, nwv - 1)];
    const f32x4 vb = reinterpret_cast<const f32x4*>(a.bias)[min((int)threadIdx.x, nbv - 1)];
    if ((int)threadIdx.x < nwv) reinterpret_cast<f32x4*>(w_s)[threadIdx.x] = v0;
    if ((int)threadIdx.x + 256 < nwv) reinterpret_cast<f32x4*>(w_s)[threadIdx.x + 256] = v1;
    for (int i = threadIdx.x + 512; i < nwv; i += 256) reinterpret_cast<f32x4*>(w_s)[i] = wg[i];
    if ((int)threadIdx.x < nbv) reinterpret_cast<f32x4*>(w_s + KS * KS * a.C)[threadIdx.x] = vb;
    for (int i = threadIdx.x + 256; i < nbv; i += 256) reinterpret_cast<f32x4*>(w_s + KS * KS * a.C)[i] = reinterpret_cast<const f32x4*>(a.bias)[i];   // (C > 1024)
  }
  __syncthreads();

  float acc[TW][8];
#pragma unroll
  for (int p = 0; p < TW; p++)
#pragma unroll
    for (int c = 0; c < 8; c++) acc[p][c] = 0.f;
  float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  if (valid) {
    // One row of taps at a time: its WIN input vectors are requested together, UNCONDITIONALLY on clamped addresses
    // (taps outside the image are zeroed by a select), then consumed.  As first written every tap load sat under its own
    // bounds check and the compiler waited for each one before the next (18 dependent round trips per lane for 3x3 and
    // four pixels: the kernel lived on occupancy alone); with no condition at all and the rows unrolled it hoisted all
    // 18 loads (196 VGPRs, two waves per SIMD, slower).  The row loop is a real loop: WIN raw vectors live at a time.
    typedef typename std::conditional<BF16, u32x4, f32x4>::type raw_t;
    typedef typename V::elem T;
    const unsigned char* in_b = reinterpret_cast<const unsigned char*>(a.in) + (int64_t)b * a.H * a.W * a.C * (int)sizeof(T);
    auto load_row = [&](int ky, raw_t* x0, raw_t* x1) {
      const int iy = oy * S - a.pad_t + ky;
      const uint32_t rowoff = (uint32_t)(min(max(iy, 0), a.H - 1) * a.W);
#pragma unroll
      for (int c0 = 0; c0 < WIN; c0++) {
        const int ix = ox0 * S - a.pad_l + c0;
        const unsigned char* src = in_b + (size_t)((rowoff + (uint32_t)min(max(ix, 0), a.W - 1)) * (uint32_t)a.C + (uint32_t)(cg * 8)) * sizeof(T);
        x0[c0] = *reinterpret_cast<const raw_t*>(src);
        if (!BF16) x1[c0] = *reinterpret_cast<const raw_t*>(src + 16);
      }
    };
    auto use_row = [&](int ky, const raw_t* xr0, const raw_t* xr1) {
      const int iy = oy * S - a.pad_t + ky;
      const bool rok = iy >= 0 && iy < a.H;
      float w[KS][8];
#pragma unroll
      for (int kx = 0; kx < KS; kx++) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(w_s + (ky * KS + kx) * a.C + cg * 8);
        f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
        for (int c = 0; c < 4; c++) { w[kx][c] = w0[c]; w[kx][4 + c] = w1[c]; }
      }
#pragma unroll
      for (int c0 = 0; c0 < WIN; c0++) {
        const int ix = ox0 * S - a.pad_l + c0;
        const bool ok = rok && ix >= 0 && ix < a.W;
        float x[8];
        if constexpr (BF16) {
          // the zero padding is selected on the four RAW words (zero bits unpack to 0.0f), not on the eight unpacked values
          const raw_t xz = ok ? xr0[c0] : raw_t{};
#pragma unroll
          for (int q = 0; q < 4; q++) { x[2 * q] = __uint_as_float(xz[q] << 16); x[2 * q + 1] = __uint_as_float(xz[q] & 0xffff0000u); }
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++) { x[q] = ok ? xr0[c0][q] : 0.f; x[4 + q] = ok ? xr1[c0][q] : 0.f; }
        }
#pragma unroll
        for (int kx = 0; kx < KS; kx++) {
          if ((c0 - kx) % S != 0 || c0 - kx < 0) continue;
          const int p = (c0 - kx) / S;
          if (p >= TW) continue;
#pragma unroll
          for (int c = 0; c < 8; c++) acc[p][c] = fmaf(x[c], w[kx][c], acc[p][c]);
        }
      }
    };
    raw_t cur0[WIN], cur1[WIN];
#pragma unroll 1      // a real loop: unrolled, all rows' loads are hoisted together (196 VGPRs, two waves per SIMD, 25 us);
                      // two rows in flight (double buffer) cost 137 VGPRs, three waves per SIMD: 18.0 us against 15.5
    for (int ky = 0; ky < KS; ky++) { load_row(ky, cur0, cur1); use_row(ky, cur0, cur1); }
    float bias[8];
    {
      const f32x4* bp = reinterpret_cast<const f32x4*>(w_s + KS * KS * a.C + cg * 8);
      f32x4 b0 = bp[0], b1 = bp[1];
#pragma unroll
      for (int c = 0; c < 4; c++) { bias[c] = b0[c]; bias[4 + c] = b1[c]; }
    }
    const int64_t oimg = (int64_t)b * a.Ho * a.Wo * a.C;
#pragma unroll
    for (int p = 0; p < TW; p++) {
      if (ox0 + p >= a.Wo) continue;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = acc[p][c] + bias[c];
      if (a.act == ACT_SWISH) swish_n<BF16, 8>(v);                    // (uniform; the packed form: same operations, two values per issue slot)
      else {
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = apply_act_t<BF16>(v[c], a.act);
      }
#pragma unroll
      for (int c = 0; c < 8; c++) sum[c] += v[c];
      V::store(a.out, oimg + ((int64_t)oy * a.Wo + ox0 + p) * a.C + cg * 8, v);
    }
  }

  if (a.hpart) {   // squeeze-excite: deterministic per-block channel sums -> partial reduce-FC products
    // thread t owns channel group (first_item + t) % CG.  Two steps, both in a fixed order: G = 256 / C
    // helper groups each add every G-th contribution of a channel, then the G helpers are added up.
    float* chs = dw_smem + (KS * KS + 1) * a.C + 256 * 9;          // [C] channel sums of this block
    // (this thread's reduce-FC weights - 32 lanes per hidden unit j = thread / 32 - are requested before the four barriers of
    //  the channel-sum reduction: their L2 round trip runs under it instead of behind it)
    const int lp = threadIdx.x & 31, jp = threadIdx.x >> 5;
    constexpr int NPF = 3;
    float wpf[NPF];
#pragma unroll
    for (int i = 0; i < NPF; i++) wpf[i] = a.se_wr[(int64_t)min(jp, a.sq - 1) * (CG * 8) + min(lp + 32 * i, CG * 8 - 1)];
#pragma unroll
    for (int c = 0; c < 8; c++) red[threadIdx.x][c] = sum[c];
    __syncthreads();
    const int C = CG * 8, G = max(1, 256 / C);
    const int first_cg = blk * 256 - fast_div(blk * 256, a.cg_magic) * CG;   // channel group of thread 0
    float part = 0.f;
    const int gq = fast_div(threadIdx.x, a.c_magic), o = threadIdx.x - gq * C;            // helper group, channel
    if (gq < G) {
      const int ocg = o >> 3, oc = o & 7;
      int t = ocg - first_cg; if (t < 0) t += CG;          // first thread of this block owning channel group ocg
      for (t += gq * CG; t < 256; t += G * CG) part += red[t][oc];
    }
    __syncthreads();
    if (gq < G) red[threadIdx.x][8] = part;
    __syncthreads();
    for (int oo = threadIdx.x; oo < C; oo += 256) {
      float s_ = 0.f;
      if (C <= 256) { for (int q = 0; q < G; q++) s_ += red[q * C + oo][8]; }
      else {                                             // more channels than threads: no helpers, walk directly
        const int ocg = oo >> 3, oc = oo & 7;
        int t = ocg - first_cg; if (t < 0) t += CG;
        for (; t < 256; t += CG) s_ += red[t][oc];
      }
      chs[oo] = s_;
    }
    __syncthreads();
    // the mean and the reduce FC are linear in these sums: hpart[b][block][j] = sum_c wr[j][c] * chs[c]; the
    // project GEMM adds the blocks up and finishes the squeeze-excite in its prologue (k_pw.hip).  32 lanes
    // per hidden unit, lane butterfly in a fixed order.
    float* hrow = a.hpart + ((int64_t)b * a.blocks_per_image + blk) * a.sqp;
    for (int j = jp; j < ((a.sq + 7) & ~7); j += 8) {
      float dot = 0.f;
      if (j < a.sq) {
        if (j == jp) {                     // (same products in the same order as the loop below)
#pragma unroll
          for (int i = 0; i < NPF; i++) if (lp + 32 * i < C) dot = fmaf(wpf[i], chs[lp + 32 * i], dot);
          for (int c = lp + 32 * NPF; c < C; c += 32) dot = fmaf(a.se_wr[(int64_t)j * C + c], chs[c], dot);
        } else {
          for (int c = lp; c < C; c += 32) dot = fmaf(a.se_wr[(int64_t)j * C + c], chs[c], dot);
        }
      }
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) dot += __shfl_xor(dot, off, 64);
      if (lp == 0 && j < a.sq) hrow[j] = dot;
    }
  }
}

int dw_blocks_per_image(int Ho, int Wo, int C, int TW) {
  const int SW = (Wo + TW - 1) / TW;
  return (Ho * SW * (C >> 3) + 255) / 256;
}

template <bool BF16, int KS, int S>
static void launch_dw_tw(const DwArgs& a, dim3 grid, hipStream_t s) {
  const size_t lds = ((size_t)(KS * KS + 1) * a.C + 256 * 9 + a.C) * sizeof(float);
  switch (a.TW) {
    case 1: hipLaunchKernelGGL((dw_kernel<BF16, KS, S, 1>), grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL((dw_kernel<BF16, KS, S, 2>), grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL((dw_kernel<BF16, KS, S, 4>), grid, dim3(256), lds, s, a); break;
  }
}
template <bool BF16>
static void launch_dw_t(const DwArgs& a, dim3 grid, hipStream_t s) {
  if (a.k == 3 && a.s == 1) launch_dw_tw<BF16, 3, 1>(a, grid, s);
  else if (a.k == 3 && a.s == 2) launch_dw_tw<BF16, 3, 2>(a, grid, s);
  else if (a.k == 5 && a.s == 1) launch_dw_tw<BF16, 5, 1>(a, grid, s);
  else launch_dw_tw<BF16, 5, 2>(a, grid, s);
}
void launch_dw(const DwArgs& a_, hipStream_t s) {
  DwArgs a = a_;
  const uint32_t CG = (uint32_t)a.C >> 3, SW = (uint32_t)(a.Wo + a.TW - 1) / a.TW;
  a.cg_magic = (uint32_t)(0x100000000ull / CG) + 1; a.sw_magic = (uint32_t)(0x100000000ull / SW) + 1;
  a.c_magic = (uint32_t)(0x100000000ull / (uint32_t)a.C) + 1;
  dim3 grid(a.blocks_per_image, a.B);
  if (a.bf16) launch_dw_t<true>(a, grid, s); else launch_dw_t<false>(a, grid, s);
}

// ------------------------------------------------------------------------------------------------
// squeeze-excite finish as its own launch: hidden = swish(inv_hw * sum_rows hpart + br), scale = sigmoid(we . hidden + be)
// (reference efficientnet/model.py:90-93).  Only for blocks whose expand-FC matrix (C x sq) is too large to be re-read by
// every workgroup of the project GEMM (phi >= 3: 2304 x 96); everywhere else the project GEMM finishes the SE in its
// prologue (k_pw_impl.h).  grid = (B, SE_SPLIT): every block rebuilds the (tiny) hidden vector and produces one slice
// of the channels; fixed summation order.
// ------------------------------------------------------------------------------------------------
#define SE_SPLIT 8
template <bool BF16>
__global__ __launch_bounds__(256) void se_finish_kernel(SeFinishArgs a) {
  extern __shared__ float se_sm[];          // hidden [sqp] | helper-group row sums [G][sqp]
  const int per = ((a.C + SE_SPLIT - 1) / SE_SPLIT + 7) & ~7;
  const int c0 = blockIdx.y * per, c1 = min(a.C, c0 + per);
  se_finish_body<BF16, 256, false>(a, blockIdx.x, c0, c1, threadIdx.x, se_sm);      // (se_finish.h)
}
void launch_se_finish(const SeFinishArgs& a, hipStream_t s) {
  const size_t lds = ((size_t)a.sqp + 256 + a.sqp) * sizeof(float);
  if (a.bf16) hipLaunchKernelGGL(se_finish_kernel<true>, dim3(a.B, SE_SPLIT), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(se_finish_kernel<false>, dim3(a.B, SE_SPLIT), dim3(256), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// MaxPool2dStaticSamePadding(3,2): the pad value is ZERO, not -inf (reference
// efficientnet/utils_extra.py:72-86), so windows that overhang the border take max(...,0).
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
  const int CG = a.C >> 3;
  const int64_t total = (int64_t)a.B * a.Ho * a.Wo * CG;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % CG);
  int64_t p = idx / CG;
  const int ox = (int)(p % a.Wo); p /= a.Wo;
  const int oy = (int)(p % a.Ho);
  const int b = (int)(p / a.Ho);
  float m[8];
  bool first = true;
#pragma unroll
  for (int ky = 0; ky < 3; ky++)
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
      const int iy = oy * 2 - a.pad_t + ky, ix = ox * 2 - a.pad_l + kx;
      float x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        Vec8<BF16>::load(a.in, (((int64_t)b * a.H + iy) * a.W + ix) * a.C + cg * 8, x);
#pragma unroll
      for (int c = 0; c < 8; c++) m[c] = first ? x[c] : fmaxf(m[c], x[c]);
      first = false;
    }
  Vec8<BF16>::store(a.out, idx * 8, m);
}
void launch_pool(const PoolArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * a.Ho * a.Wo * (a.C >> 3);
  dim3 grid((unsigned)((total + 255) / 256));
  if (a.bf16) hipLaunchKernelGGL(pool_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(pool_kernel<false>, grid, dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// feature export: NHWC (dtype) -> NCHW fp32, the layout `features` has in the reference
// (backbone.py:125) and in the ONNX outputs feat1..feat5.
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void export_kernel(ExportArgs a) {
  const int64_t total = (int64_t)a.B * a.C * a.H * a.W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int64_t p = idx;
  const int x = (int)(p % a.W); p /= a.W;
  const int y = (int)(p % a.H); p /= a.H;
  const int c = (int)(p % a.C);
  const int b = (int)(p / a.C);
  const int64_t src = (((int64_t)b * a.H + y) * a.W + x) * a.C + c;
  float v;
  if (BF16) v = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(a.in)[src]);
  else v = reinterpret_cast<const float*>(a.in)[src];
  a.out[idx] = v;
}
void launch_export(const ExportArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * a.C * a.H * a.W;
  dim3 grid((unsigned)((total + 255) / 256));
  if (a.bf16) hipLaunchKernelGGL(export_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(export_kernel<false>, grid, dim3(256), 0, s, a);
}
