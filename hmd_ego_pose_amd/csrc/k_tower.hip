// k_tower.hip - the head layers (tower SeparableConvBlocks and the header convs of all five heads)
// as barrier-free, wave-independent work on gfx950:
//
//   depthwise 3x3 SAME (no bias) -> pointwise 1x1 (+bias, per-level BN folded) -> swish | sigmoid | none
//
// reference: efficientdet/model.py:361-417 (Regressor / Classifier), hmdegopose/model.py:55-90,
// 127-156, 191-228 (rotation / translation / hand nets): `conv(feat); bn(feat); swish(feat)` per
// tower layer, then the header conv + permute/view/cat into [B, N_anchors, K].
//
// k_sep.hip runs the same math one 8x8 tile per workgroup through five LDS phases; with ~1800 tiny
// workgroups per layer its time is the per-workgroup latency chain, not bandwidth.  Here every WAVE
// owns one 16-pixel MFMA m-tile (a 4x4 patch; 4 waves = one 8x8 tile so neighbours share L1 lines):
//   1. each lane loads the 9 taps of ITS pixel x 8 (bf16) / 4 (fp32) channels straight from global
//      memory - exactly the channels it holds in the MFMA operand - and accumulates the depthwise
//      conv in registers; the operand never exists in memory or LDS,
//   2. maps:    D[n, pixel] = W . X^T  (weights as A operand): a lane ends with 4 consecutive
//               channels of one pixel; the planner permutes the weight rows so that two n-tiles give
//               8 consecutive channels -> one 16-byte store straight to the NHWC map,
//      headers: the same product; a lane's 4 consecutive columns of its pixel leave as one 16-byte
//               store into the [B, N_anchors, K] result (4-byte aligned: rows are 9*K floats).
// LDS holds the depthwise weights + bias (one barrier) and a wave-private slot for the finished
// operand fragments.
//
// Two kernels share the arithmetic (same taps in the same order, same k-step order: bit-identical results):
//   tower_kernel       wave-independent, as above (bf16 widths 64-112 where the four 6x6 halos fit next to the weights)
//   tower_coop_kernel  the workgroup owns the tile: one 10x10 halo, the waves split k-steps (depthwise) and output
//                      units (pointwise, weights in registers); see the comment in front of it
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

// Optional per-wave timeline (make EXTRA=-DHEP_TOWER_TRACE): every wave stamps s_memrealtime (100 MHz)
// at its phase boundaries into a device buffer read back with hep_dbg_tower_trace() - a profiling build
// only, the stamps are compiled out of the product library.
#ifdef HEP_TOWER_TRACE
__device__ unsigned long long* g_tower_trace = nullptr;
__device__ int g_tower_trace_maps = 0;      // 0: the header launch writes the stamps, 1: the map layers (the last one wins)
#define TSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TSTAMP_NOWAIT(i) do { stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP(i)
#define TSTAMP_NOWAIT(i)
#endif

#define GLOBAL __attribute__((address_space(1)))   // pointers read from the descriptor are global, not flat

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

// One MFMA operand fragment (8 bf16 / 4 fp32 channels of one pixel) and the depthwise accumulation over
// it.  bf16: the even channels of two packed words are unpacked with one shift each, the odd ones with
// one mask each, and every pair feeds one v_pk_fma_f32 - so the accumulators (and the depthwise
// weights in LDS, see `swz`) are kept in the order 0,2,1,3 | 4,6,5,7.
template <bool BF16> struct Frag;
template <> struct Frag<true> {
  typedef u32x4 raw;
  struct Acc { f32x2 a[4]; };      // channels (0,2) (1,3) (4,6) (5,7)
  static __device__ __forceinline__ f32x4 swz(f32x4 w) { return (f32x4){w[0], w[2], w[1], w[3]}; }
  static __device__ __forceinline__ void zero(Acc& c) { for (int i = 0; i < 4; i++) c.a[i] = (f32x2){0.f, 0.f}; }
  static __device__ __forceinline__ void fma_tap(Acc& c, const raw& x, const float* w /* 8 swizzled weights in LDS */) {
    const f32x4 wa = reinterpret_cast<const f32x4*>(w)[0], wb = reinterpret_cast<const f32x4*>(w)[1];
    const f32x2 e0 = {__uint_as_float(x[0] << 16), __uint_as_float(x[1] << 16)}, o0 = {__uint_as_float(x[0] & 0xffff0000u), __uint_as_float(x[1] & 0xffff0000u)};
    const f32x2 e1 = {__uint_as_float(x[2] << 16), __uint_as_float(x[3] << 16)}, o1 = {__uint_as_float(x[2] & 0xffff0000u), __uint_as_float(x[3] & 0xffff0000u)};
    c.a[0] = __builtin_elementwise_fma(e0, (f32x2){wa[0], wa[1]}, c.a[0]);
    c.a[1] = __builtin_elementwise_fma(o0, (f32x2){wa[2], wa[3]}, c.a[1]);
    c.a[2] = __builtin_elementwise_fma(e1, (f32x2){wb[0], wb[1]}, c.a[2]);
    c.a[3] = __builtin_elementwise_fma(o1, (f32x2){wb[2], wb[3]}, c.a[3]);
  }
  static __device__ __forceinline__ void fma_tap_r(Acc& c, const raw& x, const f32x4& wa, const f32x4& wb) {   // weights in registers (swizzled)
    const f32x2 e0 = {__uint_as_float(x[0] << 16), __uint_as_float(x[1] << 16)}, o0 = {__uint_as_float(x[0] & 0xffff0000u), __uint_as_float(x[1] & 0xffff0000u)};
    const f32x2 e1 = {__uint_as_float(x[2] << 16), __uint_as_float(x[3] << 16)}, o1 = {__uint_as_float(x[2] & 0xffff0000u), __uint_as_float(x[3] & 0xffff0000u)};
    c.a[0] = __builtin_elementwise_fma(e0, (f32x2){wa[0], wa[1]}, c.a[0]);
    c.a[1] = __builtin_elementwise_fma(o0, (f32x2){wa[2], wa[3]}, c.a[1]);
    c.a[2] = __builtin_elementwise_fma(e1, (f32x2){wb[0], wb[1]}, c.a[2]);
    c.a[3] = __builtin_elementwise_fma(o1, (f32x2){wb[2], wb[3]}, c.a[3]);
  }
  static __device__ __forceinline__ raw pack(const Acc& c) {
    return (raw){pack_bf16x2(c.a[0][0], c.a[1][0]), pack_bf16x2(c.a[0][1], c.a[1][1]), pack_bf16x2(c.a[2][0], c.a[3][0]), pack_bf16x2(c.a[2][1], c.a[3][1])};
  }
  static __device__ __forceinline__ f32x4 mma(const raw& a, const raw& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Frag<false> {
  typedef f32x4 raw;
  struct Acc { f32x4 a; };
  static __device__ __forceinline__ f32x4 swz(f32x4 w) { return w; }
  static __device__ __forceinline__ void zero(Acc& c) { c.a = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ void fma_tap(Acc& c, const raw& x, const float* w) {
    c.a = __builtin_elementwise_fma(x, *reinterpret_cast<const f32x4*>(w), c.a);
  }
  static __device__ __forceinline__ void fma_tap_r(Acc& c, const raw& x, const f32x4& wa, const f32x4&) { c.a = __builtin_elementwise_fma(x, wa, c.a); }
  static __device__ __forceinline__ raw pack(const Acc& c) { return c.a; }
  static __device__ __forceinline__ f32x4 mma(const raw& a, const raw& b, f32x4 c) {
#pragma unroll
    for (int q = 0; q < 4; q++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], c, 0, 0, 0);
    return c;
  }
};

}  // namespace

// LDS layout of one workgroup (4 waves), shared with tower_lds_bytes():
//   [9*CW] f32 depthwise weights | [BIAS] f32 bias (16 per n-tile of the widest segment) | 4 waves x KS x 64 lanes operand fragments
//   | (when it fits in 48 KB) the segment's pointwise weights, rows padded against bank conflicts
//   | (when they fit in 40 KB) the 6x6-pixel input halos of the 4 waves
template <bool BF16, int CW, bool HDR> struct TowerCfg {
  static constexpr int ES = BF16 ? 2 : 4;
  static constexpr int KL = BF16 ? 8 : 4, KSTEP = 4 * KL, KS = (CW + KSTEP - 1) / KSTEP;
  // Waves per workgroup: the four 4x4 patches of one 8x8 tile, images one after the other.  (Measured at width 160, phi 3 @
  // 512 b8: 16 waves = the same patches of four images side by side around ONE copy of the weights, 16 waves per CU
  // instead of 8 - 76.5 us per tower layer against 72.5: the layer is not short of waves.)
  static constexpr int NW = 4;
  static constexpr int NTMAP = (((CW + 15) / 16) + 1) & ~1;           // n-tiles of a map layer (even)
  static constexpr int WROWS = (HDR ? tower_hdr_tiles(CW, BF16) : NTMAP) * 16;  // most weight rows a segment has
  static constexpr int WP = CW + (BF16 ? 8 : 4);                      // LDS row pitch (elements)
  // (width 160 in bf16: 53.8 KB of weights - with the other regions 80.6 KB, two workgroups per CU)
  static constexpr size_t XA_BYTES = (size_t)NW * KS * 64 * 16;
  static constexpr bool WLDS = (size_t)WROWS * WP * ES <= (BF16 ? TOWER_WLDS_MAX : 48 * 1024) && (size_t)WROWS * WP * ES + (size_t)9 * CW * 4 + WROWS * 4 + XA_BYTES <= 158 * 1024;
  static constexpr int HP = CW + 16 / ES;                             // halo pixel pitch (elements): +16 bytes
  // (round 4, fp32 at width 64: without the halos - 36 KB of LDS, four workgroups per CU instead of two - a tower layer takes 42 us
  //  against 31 us and the headers 96 against 66: the halo's coalesced loads are worth more than the occupancy; three waves per
  //  SIMD in the launch bounds with the halos: 29.3 against 30.9 us, inside the noise of the step)
  static constexpr bool HALO = (size_t)4 * 36 * HP * ES <= 40 * 1024;  // the 6x6-pixel halos of 4 waves fit
  static constexpr int BIAS = WROWS;                                  // bias floats staged per segment (one per weight row)
  static constexpr size_t OFF_XA = ((size_t)9 * CW + BIAS) * 4;
  static constexpr size_t OFF_W = OFF_XA + XA_BYTES;
  static constexpr size_t OFF_HALO = OFF_W + (WLDS ? (size_t)WROWS * WP * ES : 0);
  static constexpr size_t LDS = OFF_HALO + (HALO ? (size_t)4 * 36 * HP * ES : 0);
};

template <bool BF16, int CW, bool HDR>
__global__ __launch_bounds__(256, BF16 ? 4 : 2) void tower_kernel(const SepSeg* __restrict__ segs, const int* __restrict__ tile_seg, int B, int ipb) {
  typedef Frag<BF16> F;
  typedef typename F::raw raw_t;
  typedef typename Vec8<BF16>::elem T;
  typedef TowerCfg<BF16, CW, HDR> Cfg;
  constexpr int KL = Cfg::KL, KSTEP = Cfg::KSTEP, KS = Cfg::KS, ES = Cfg::ES, WP = Cfg::WP;
  constexpr bool KFULL = CW % KSTEP == 0, WLDS = Cfg::WLDS;
  constexpr int NTH = Cfg::NW * 64, IPAR = Cfg::NW / 4;            // threads; images side by side
  constexpr int G = KS <= 2 ? KS : (BF16 ? 1 : 2), NG = (KS + G - 1) / G;   // k-steps whose 9 tap loads are in flight together
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, Cfg::LDS);
#ifdef HEP_TOWER_TRACE
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TSTAMP_NOWAIT(0);
#endif
  float* wdw_s = reinterpret_cast<float*>(smem);
  float* bias_s = wdw_s + 9 * CW;
  T* w_s = reinterpret_cast<T*>(smem + Cfg::OFF_W);

  // XCD-aware order: consecutive LOGICAL blocks are neighbouring 8x8 tiles of one map, whose 6x6 patch halos overlap:
  // on one XCD the overlap is an L2 hit instead of a second fetch (PMC round 1: 1.37x the algorithmic reads)
  int bxl, byl;
  xcd_remap2(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, &bxl, &byl);
  const int si = __builtin_amdgcn_readfirstlane(tile_seg[bxl]);
  const SepSeg* __restrict__ sg = segs + si;           // uniform + read-only: descriptor fields arrive as scalar loads
  const int h = sg->h, w = sg->w, tiles_x = sg->tiles_x, tilesN = sg->tilesN;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // (uniform: the image base below must stay in scalar registers)
  const int lane = threadIdx.x & 63, wave = wv & 3, iw = wv >> 2, r = lane & 15, g = lane >> 4;   // wave: patch of the tile; iw: image slot
  const GLOBAL T* W = (const GLOBAL T*)sg->wpw;
  TSTAMP(1);                           // descriptor arrived
  // ---- stage the layer's weights (overlaps with the tap loads issued below) ----
  // (every load of the staging is issued before the first LDS store: one memory round trip, not one per
  //  loop iteration)
  {
    constexpr int NDW = (9 * CW / 4 + NTH - 1) / NTH, NB = (Cfg::BIAS / 4 + NTH - 1) / NTH;
    constexpr int VPR = CW / KL, NW = WLDS ? (Cfg::WROWS * VPR + NTH - 1) / NTH : 0;
    const GLOBAL f32x4* gdw = (const GLOBAL f32x4*)sg->wdw;
    const GLOBAL f32x4* gb = (const GLOBAL f32x4*)sg->bias;
    f32x4 vdw[NDW], vb[NB];
    raw_t vw[NW > 0 ? NW : 1];
#pragma unroll
    for (int j = 0; j < NDW; j++) { const int i = threadIdx.x + j * NTH; if (i < 9 * CW / 4) vdw[j] = gdw[i]; }
#pragma unroll
    for (int j = 0; j < NB; j++) { const int i = threadIdx.x + j * NTH; if (i < tilesN * 4) vb[j] = gb[i]; }
#pragma unroll
    for (int j = 0; j < NW; j++) {
      const int i = threadIdx.x + j * NTH, row = i / VPR, v = i - row * VPR;
      if (i < tilesN * 16 * VPR) vw[j] = *(const GLOBAL raw_t*)(W + row * CW + v * KL);
    }
#pragma unroll
    for (int j = 0; j < NDW; j++) { const int i = threadIdx.x + j * NTH; if (i < 9 * CW / 4) reinterpret_cast<f32x4*>(wdw_s)[i] = F::swz(vdw[j]); }
#pragma unroll
    for (int j = 0; j < NB; j++) { const int i = threadIdx.x + j * NTH; if (i < tilesN * 4) reinterpret_cast<f32x4*>(bias_s)[i] = vb[j]; }
#pragma unroll
    for (int j = 0; j < NW; j++) {
      const int i = threadIdx.x + j * NTH, row = i / VPR, v = i - row * VPR;
      if (i < tilesN * 16 * VPR) *reinterpret_cast<raw_t*>(w_s + row * WP + v * KL) = vw[j];
    }
  }

  // one workgroup = one 8x8 tile position of `ipb` consecutive images: the staged weights are shared
  // and the taps of image i+1 are in flight while image i goes through the MFMAs
  const int t = bxl - sg->tile_begin;
  const int b0 = byl * ipb + iw, nimg = (min(ipb, B - byl * ipb) - iw + IPAR - 1) / IPAR;     // this wave's images: b0, b0 + IPAR, ...
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * 8 + (wave >> 1) * 4, x0 = tx * 8 + (wave & 1) * 4;   // this wave's 4x4 patch
  const int y = y0 + (r >> 2), x = x0 + (r & 3);                            // this lane's pixel (MFMA row/col r)
  const bool pix_ok = y < h && x < w;

  // ---- depthwise 3x3 in registers: lane = (pixel r, channels ks*KSTEP + KL*g ..) ----
  // Input pixels arrive by raw buffer loads of one image: anything in the SAME padding (or belonging to a
  // lane without a pixel) gets an out-of-range offset and the hardware returns zeros - no branches, no
  // masking.  Two ways to get the 9 taps into the MFMA lane layout:
  //   HALO  the wave's 6x6-pixel halo is fetched once with fully coalesced 16-byte lanes (consecutive
  //         lanes = consecutive channels of a pixel) into a wave-private LDS block and the taps are LDS
  //         reads.  In the MFMA layout consecutive lanes are different PIXELS, so direct tap loads touch
  //         four cache lines per quad and were measured ~1.6x slower through the texture addresser.
  //   else  (wide layers whose halos do not fit) the taps are loaded straight into registers.
  constexpr bool HALO = Cfg::HALO;
  constexpr int HP = Cfg::HP, CPP = CW / KL, NV = HALO ? (36 * CPP + 63) / 64 : 1;
  const int img_elems = h * w * CW;
  const GLOBAL T* X0 = (const GLOBAL T*)sg->src[0] + (int64_t)b0 * img_elems;    // image bi of this wave: X0 + bi * IPAR images
  constexpr uint32_t OOB = 0x80000000u;
  T* halo = reinterpret_cast<T*>(smem + Cfg::OFF_HALO) + wave * 36 * HP;
  uint32_t toff[9];                    // direct taps: byte offsets; HALO: LDS element offsets of the 9 taps
  uint32_t hoff[NV]; int hdst[NV];     // HALO: this lane's vectors of the halo (global byte offset, LDS element offset)
  raw_t hv[NV];
  if constexpr (HALO) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int v = lane + 64 * j, hp = v / CPP, ch = v - hp * CPP;     // halo slot hp = hx * 6 + hy (column-major:
      const int hx = hp / 6, hy = hp - hx * 6;                          //  conflict-free tap reads with the 16-byte pad)
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = v < 36 * CPP && iy >= 0 && iy < h && ix >= 0 && ix < w;
      hoff[j] = ok ? (uint32_t)((iy * w + ix) * CW + ch * KL) * ES : OOB;
      hdst[j] = v < 36 * CPP ? hp * HP + ch * KL : -1;
    }
#pragma unroll
    for (int q = 0; q < 9; q++) toff[q] = (uint32_t)((((r & 3) + q % 3) * 6 + (r >> 2) + q / 3) * HP + KL * g);
  } else {
#pragma unroll
    for (int q = 0; q < 9; q++) {
      const int iy = y + q / 3 - 1, ix = x + q % 3 - 1;
      const bool ok = pix_ok && iy >= 0 && iy < h && ix >= 0 && ix < w;
      toff[q] = ok ? (uint32_t)((iy * w + ix) * CW + KL * g) * ES : OOB;
    }
  }
  raw_t tp[G][9];
  auto load_group = [&](int bi, int gi) {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(X0 + (int64_t)bi * IPAR * img_elems), 0, img_elems * ES, 0x00020000);
    if constexpr (HALO) {
      if (gi == 0) {
#pragma unroll
        for (int j = 0; j < NV; j++) hv[j] = __builtin_bit_cast(raw_t, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[j], 0, 0));
      }
    } else {
#pragma unroll
      for (int j = 0; j < G; j++) {
        const int ks = gi * G + j;
        if (ks < KS) {
          const bool kok = KFULL || ks * KSTEP + KL * g < CW;
#pragma unroll
          for (int q = 0; q < 9; q++)
            tp[j][q] = __builtin_bit_cast(raw_t, __builtin_amdgcn_raw_buffer_load_b128(xrs, kok ? toff[q] : OOB, ks * KSTEP * ES, 0));
        }
      }
    }
  };
  raw_t* xa_w = reinterpret_cast<raw_t*>(smem + Cfg::OFF_XA) + wv * KS * 64;      // this wave's operand slots [k-step][lane]
  raw_t* xa_s = xa_w + lane;
  // quad form of the depthwise phase (width 64, halos in LDS): lane = (channel quad, patch row), see the image loop
  constexpr bool DWQ = Cfg::HALO && CW == 64;
  const int dq_cq = lane & 15, dq_py = lane >> 4;
  // operand slot (byte offset) of channels 4 cq .. 4 cq + 3 of pixel r = 4 py + px.  bf16: k-step cq / 8, lane group (cq % 8) / 2, half cq & 1
  // of the 16-byte fragment; fp32: k-step cq / 4, lane group cq % 4, the whole 16-byte fragment
  const int dq_slot = BF16 ? (((dq_cq >> 3) * 64 + ((dq_cq & 7) >> 1) * 16 + dq_py * 4) << 4) + (dq_cq & 1) * 8
                           : (((dq_cq >> 2) * 64 + (dq_cq & 3) * 16 + dq_py * 4) << 4);
  if (nimg > 0) load_group(0, 0);
  TSTAMP_NOWAIT(2);                    // staging + first taps issued
  __syncthreads();                     // weights are in LDS; the first taps are in flight
  TSTAMP_NOWAIT(3);                    // barrier passed
  if (y0 >= h || x0 >= w || nimg <= 0) return;      // patch entirely outside the map / no image for this slot

  // pointwise weight fragment of MFMA row `row`, k-step ks (zero where k >= CW)
  auto wfrag = [&](int row, int ks) -> raw_t {
    const bool kok = KFULL || ks * KSTEP + KL * g < CW;
    const int k = kok ? ks * KSTEP + KL * g : 0;
    raw_t v;
    if constexpr (WLDS) v = *reinterpret_cast<const raw_t*>(w_s + row * WP + k);
    else v = *(const GLOBAL raw_t*)(W + row * CW + k);
    if (!KFULL && !kok) v = raw_t{};
    return v;
  };
  const int N = sg->N, act = sg->act;
  constexpr int KU = KS < 4 ? KS : 4;   // k-steps of weight fragments in flight

#pragma unroll 1
  for (int bi = 0; bi < nimg; bi++) {
  const int b = b0 + bi * IPAR;
  if constexpr (HALO) {
    // park this image's halo (the previous image's tap reads are done: LDS executes a wave in order),
    // then put the next image's halo in flight
    {
#pragma unroll
      for (int j = 0; j < NV; j++) if (hdst[j] >= 0) *reinterpret_cast<raw_t*>(halo + hdst[j]) = hv[j];
      if (bi + 1 < nimg) load_group(bi + 1, 0);
    }
  }
  if constexpr (DWQ) {
    // depthwise, quad form: lane = (channel quad cq, patch row py) computes the four pixels of its row for four channels.  The six halo
    // columns of a tap row are read (8 bytes) and unpacked ONCE for the four pixels and a row's three weight vectors once: 18 + 9 LDS reads
    // and 72 unpack instructions per image and lane instead of 18 + 36 reads (16 bytes each) and 144.  Taps in the order q = ky * 3 + kx
    // into the same (c0, c2) / (c1, c3) accumulator pairs as Frag::fma_tap: bit-identical.
    f32x2 ae[4], ao[4];
#pragma unroll
    for (int px = 0; px < 4; px++) { ae[px] = (f32x2){0.f, 0.f}; ao[px] = (f32x2){0.f, 0.f}; }
    const T* hrow = halo + dq_py * HP + dq_cq * 4;
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      // the row's three weight vectors (one 16-byte LDS read each, shared by the four pixels), then the six halo columns one at a time:
      // column hx feeds pixel px = hx - kx with tap kx, so every output still sees its taps in the order kx = 0, 1, 2
      f32x2 we[3], wo[3];
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const f32x4 wv4 = *reinterpret_cast<const f32x4*>(wdw_s + (ky * 3 + kx) * CW + dq_cq * 4);
        we[kx] = (f32x2){wv4[0], wv4[1]}; wo[kx] = (f32x2){wv4[2], wv4[3]};
      }
#pragma unroll
      for (int hx = 0; hx < 6; hx++) {
        f32x2 e, o;                                                  // bf16: channels (c0, c2) / (c1, c3) as Frag<true>; fp32: (c0, c1) / (c2, c3)
        if constexpr (BF16) {
          const u32x2 v = *reinterpret_cast<const u32x2*>(hrow + (hx * 6 + ky) * HP);
          e = (f32x2){__uint_as_float(v[0] << 16), __uint_as_float(v[1] << 16)};
          o = (f32x2){__uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] & 0xffff0000u)};
        } else {
          const f32x4 v = *reinterpret_cast<const f32x4*>(hrow + (hx * 6 + ky) * HP);
          e = (f32x2){v[0], v[1]}; o = (f32x2){v[2], v[3]};
        }
#pragma unroll
        for (int kx = 2; kx >= 0; kx--) {
          const int px = hx - kx;
          if (px >= 0 && px < 4) {
            ae[px] = __builtin_elementwise_fma(e, we[kx], ae[px]);
            ao[px] = __builtin_elementwise_fma(o, wo[kx], ao[px]);
          }
        }
      }
    }
#pragma unroll
    for (int px = 0; px < 4; px++) {
      unsigned char* slot = reinterpret_cast<unsigned char*>(xa_w) + dq_slot + px * 16;
      if constexpr (BF16) *reinterpret_cast<u32x2*>(slot) = (u32x2){pack_bf16x2(ae[px][0], ao[px][0]), pack_bf16x2(ae[px][1], ao[px][1])};
      else *reinterpret_cast<f32x4*>(slot) = (f32x4){ae[px][0], ae[px][1], ao[px][0], ao[px][1]};
    }
  } else
#pragma unroll 1
  for (int gi = 0; gi < NG; gi++) {
#pragma unroll
    for (int j = 0; j < G; j++) {
      const int ks = gi * G + j;
      if (ks < KS) {
        const int k = ks * KSTEP + KL * g;
        const bool kok = KFULL || k < CW;
        const float* wl = wdw_s + (kok ? k : 0);
        typename F::Acc acc;
        F::zero(acc);
#pragma unroll
        for (int q = 0; q < 9; q++) {
          if constexpr (HALO) F::fma_tap(acc, *reinterpret_cast<const raw_t*>(halo + toff[q] + (kok ? ks * KSTEP : 0)), wl + q * CW);
          else F::fma_tap(acc, tp[j][q], wl + q * CW);
        }
        raw_t xv = F::pack(acc);
        if (HALO && !KFULL && !kok) xv = raw_t{};      // k >= CW: the operand must be exactly zero
        xa_s[ks * 64] = xv;
      }
    }
    if (!HALO && gi + 1 < NG) load_group(bi, gi + 1);
  }
  if (!HALO && bi + 1 < nimg) load_group(bi + 1, 0);     // next image's taps fly during the MFMA phase
  TSTAMP_NOWAIT(4);                    // depthwise done (fragments in LDS)
  if constexpr (!HDR) {
    // ---- maps: lane (r, g) owns channels g*RUN + 4*nt .. +3 of pixel r for every n-tile nt ----
    constexpr int NT = Cfg::NTMAP, RUN = 4 * NT;
    GLOBAL T* O = (GLOBAL T*)sg->out + (int64_t)b * sg->out_bstride + sg->out_off + ((int64_t)y * w + x) * sg->out_rowstride;
#pragma unroll(NT <= 4 ? 2 : 1)
    for (int nt = 0; nt < NT; nt += 2) {
      const int ch = g * RUN + nt * 4;
      f32x4 acc0 = *reinterpret_cast<const f32x4*>(bias_s + ch), acc1 = *reinterpret_cast<const f32x4*>(bias_s + ch + 4);
#pragma unroll(KU)
      for (int ks = 0; ks < KS; ks++) {
        const raw_t w0 = wfrag(nt * 16 + r, ks), w1 = wfrag(nt * 16 + 16 + r, ks);
        const raw_t xv = xa_s[ks * 64];
        acc0 = F::mma(w0, xv, acc0);
        acc1 = F::mma(w1, xv, acc1);
      }
      if (pix_ok && ch < N) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; q++) { v[q] = acc0[q]; v[4 + q] = acc1[q]; }
        if (act == ACT_SWISH) {           // uniform: one branch per n-tile pair, not one per element
          swish_n<BF16, 8>(v);
        } else if (act == ACT_SIGMOID) {
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = sigmoid_t<BF16>(v[q]);
        }
        if constexpr (BF16) {
          u32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; e++) pk[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
          if (ch + 8 <= N) *(GLOBAL u32x4*)(O + ch) = pk;
          else *(GLOBAL u32x2*)(O + ch) = (u32x2){pk[0], pk[1]};
        } else {
          *(GLOBAL f32x4*)(O + ch) = (f32x4){v[0], v[1], v[2], v[3]};
          if (ch + 8 <= N) *(GLOBAL f32x4*)(O + ch + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        }
      }
    }
  } else {
    // ---- headers: lane (r, g) owns columns nt*16 + 4g .. +3 of pixel r: one (4-byte aligned) 16-byte
    //      store per n-tile where the head's columns are contiguous in [B, N_anchors, K] ----
    typedef float __attribute__((ext_vector_type(4), aligned(4))) f32x4_u;
    const int kin = sg->col_kin, kout = sg->col_kout, coff = sg->col_off, nbase = sg->n_base;
    GLOBAL float* O = (GLOBAL float*)sg->out + (int64_t)b * sg->out_bstride + sg->out_off + ((int64_t)y * w + x) * sg->out_rowstride;
    const bool contiguous = kin == kout && coff == 0;
#pragma unroll 2
    for (int nt = 0; nt < tilesN; nt++) {
      const int n = nt * 16 + 4 * g;
      f32x4 acc = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll(KU)
      for (int ks = 0; ks < KS; ks++) acc = F::mma(wfrag(nt * 16 + r, ks), xa_s[ks * 64], acc);
      if (pix_ok && n < N) {
        if (act == ACT_SIGMOID) {
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] = sigmoid_t<BF16>(acc[q]);
        } else if (act == ACT_SWISH) {
#pragma unroll
          for (int q = 0; q < 4; q++) acc[q] = swish_t<BF16>(acc[q]);
        }
        if (contiguous && n + 4 <= N) {
          *(GLOBAL f32x4_u*)(O + nbase + n) = acc;
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int nn = nbase + n + q;
            if (n + q < N) O[(nn / kin) * kout + nn % kin + coff] = acc[q];
          }
        }
      }
    }
  }
  }   // image loop
#ifdef HEP_TOWER_TRACE
  TSTAMP_NOWAIT(5);                    // MFMA + stores issued
  TSTAMP(6);                           // stores acknowledged
  if (g_tower_trace && lane == 0 && (g_tower_trace_maps != 0) == !HDR) {
    unsigned long long* o = g_tower_trace + ((size_t)(byl * gridDim.x + bxl) * Cfg::NW + (threadIdx.x >> 6)) * 8;
    for (int i = 0; i < 7; i++) o[i] = stamps[i];
    o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
  }
#endif
}


// ---- cooperative form: the layers whose wave-private halos do not fit next to the pointwise weights (bf16 from width 160) ----
// tower_kernel above keeps a layer's [N][C] pointwise weights in LDS for its four independent waves; at width 160 that is 54 KB,
// the four 6x6 halos (48 KB) no longer fit and every tap became a direct load in the MFMA lane layout (four cache lines per
// quad: 1.95 TB/s at phi 3 @ 512, and per wave and image 100 weight-fragment + 90 depthwise-weight LDS reads).  Here the
// WORKGROUP owns the 8x8 tile and the waves divide the layer between them:
//   * wave w holds the weight fragments of output unit w (bf16: the n-tile pair 2w, 2w+1 = 8 consecutive channels per lane,
//     one 16-byte store; fp32: n-tile w) for ALL k-steps in registers (KS x UN x 4 VGPRs) - loaded once per workgroup, reused
//     for the four m-tiles of every image; headers: units w, w + NWV, ...
//   * the tile's ONE 10x10 halo arrives by coalesced 16-byte lanes (consecutive lanes = consecutive channels of a pixel) into
//     LDS; the next image's halo is in flight in registers while this one is worked on,
//   * depthwise: wave w computes k-step w of all four m-tiles (its 9 x 8 weights in registers) and parks the operand fragments
//     in LDS in MFMA lane order; after the barrier every wave multiplies its unit against all of them.
// The widths pair up: k-steps == units for every BiFPN width in bf16 (in fp32 units >= k-steps), so NWV = units.
// Two barriers per image; LDS at width 160: 33.6 KB halo + 20 KB fragments + bias = 55 KB, two workgroups (10 waves) per CU.
template <bool BF16, int CW, bool HDR> struct CoopCfg {
  static constexpr int ES = BF16 ? 2 : 4, KL = BF16 ? 8 : 4, KSTEP = 4 * KL, KS = (CW + KSTEP - 1) / KSTEP;
  static constexpr int NTMAP = (((CW + 15) / 16) + 1) & ~1;
  static constexpr int UN = (BF16 && !HDR) ? 2 : 1;                   // n-tiles per output unit
  static constexpr int NU = BF16 ? NTMAP / 2 : NTMAP;                 // output units of a map layer (>= KS)
  // Narrow layers (bf16 width 64: two units, two k-steps) split the four patches as well: NU * PGM waves, wave w = (unit w % NU,
  // patch group w / NU) in the MFMA phase and (k-step w % KS, patch group w / KS) in the depthwise phase
  // (the header launch of bf16 width 64: eight waves - 36 n-tiles are 5 per wave, not 9: 40 weight registers, and one depthwise task each)
  static constexpr int PGM = NU >= 4 ? 1 : 2, NWV = (HDR && BF16 && CW == 64) ? 8 : NU * PGM, PPM = 4 / PGM;    // MFMA phase: patch groups, waves, patches per wave
  static constexpr int PGD = NWV / KS >= 4 ? 4 : (NWV / KS >= 2 ? 2 : 1), PPD = 4 / PGD;   // depthwise phase (waves >= KS * PGD: none)
  static constexpr int HTILES = tower_coop_hdr_tiles(CW, BF16);       // most n-tiles a header segment has (planner: the same function)
  static constexpr int MAXU = HDR ? (HTILES + NWV - 1) / NWV : 1;     // units per wave
  // halo pixel pitch: a multiple of 16 bytes that is 2 mod 4 in 16-byte units.  ds_read_b128 is served in the lane groups
  // {0-3, 12-15, 20-27} ... (MI355X_MICROARCH.md, LDS): patch rows 0 and 3 of one channel group with rows 1 and 2 of the next; with
  // column-major halo slots (slot = hx * 10 + hy) those 16 reads fall on 16 different 16-byte bank slots for every tap exactly
  // when the pitch is 2, 6, 10 or 14 (mod 16) units (enumerated); the natural +16-byte pad (21 units at width 160) is 2-way
  static constexpr int PU = CW * ES / 16, HP = (PU + (6 - PU % 4) % 4) * 16 / ES;
  static constexpr int BIAS = (HDR ? HTILES : NTMAP) * 16;
  // Image slots: the hardware places a workgroup as ceil(waves / 4) waves on EVERY SIMD, so a 5-wave workgroup at 3 waves per
  // SIMD (<= 168 VGPRs) is alone on its CU (traced: 256 workgroups resident, not 512).  Such layers run TWO images side by
  // side in one 10-wave workgroup - two independent halves (own halo, own fragments) sharing bias, depthwise weights and barriers.
  static constexpr int IPAR = NWV == 5 ? 2 : 1;
  static constexpr size_t OFF_XA = ((size_t)BIAS + 9 * CW) * 4;          // [BIAS] f32 bias | [9][CW] f32 depthwise weights | per slot: fragments | halo
  static constexpr size_t XA_BYTES = (size_t)4 * KS * 64 * 16, HALO_BYTES = (size_t)100 * HP * ES, SLOT_BYTES = XA_BYTES + HALO_BYTES;
  static constexpr size_t LDS = OFF_XA + IPAR * SLOT_BYTES;
  static constexpr int MINW = (BF16 && CW == 64) ? 4 : (NWV <= 5 ? 3 : 1);   // waves per SIMD the register allocation must leave room for
  static_assert(NWV >= KS, "one depthwise k-step per wave");
};

template <bool BF16, int CW, bool HDR>
__global__ __launch_bounds__((CoopCfg<BF16, CW, HDR>::IPAR * CoopCfg<BF16, CW, HDR>::NWV * 64), (CoopCfg<BF16, CW, HDR>::MINW)) void tower_coop_kernel(const SepSeg* __restrict__ segs, const int* __restrict__ tile_seg, int B, int ipb) {
  typedef Frag<BF16> F;
  typedef typename F::raw raw_t;
  typedef typename Vec8<BF16>::elem T;
  typedef CoopCfg<BF16, CW, HDR> Cfg;
  constexpr int KL = Cfg::KL, KSTEP = Cfg::KSTEP, KS = Cfg::KS, ES = Cfg::ES, HP = Cfg::HP, NWV = Cfg::NWV, UN = Cfg::UN, MAXU = Cfg::MAXU;
  constexpr int NU = Cfg::NU, PPM = Cfg::PPM, PGD = Cfg::PGD, PPD = Cfg::PPD;
  constexpr bool KFULL = CW % KSTEP == 0;
  constexpr int IPAR = Cfg::IPAR, NTHA = IPAR * NWV * 64;             // image slots; threads of the whole workgroup
  constexpr int NTH = NWV * 64, CPP = CW / KL, NV = (100 * CPP + NTH - 1) / NTH, PPJ = NTH / CPP;   // PPJ: halo pixels one pass of the threads covers
  static_assert(NTH % CPP == 0, "a thread keeps its channel vector from pass to pass");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, Cfg::LDS);
#ifdef HEP_TOWER_TRACE
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TSTAMP_NOWAIT(0);
#endif
  float* bias_s = reinterpret_cast<float*>(smem);
  float* wdw_s = bias_s + Cfg::BIAS;

  int bxl, byl;
  xcd_remap2(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, &bxl, &byl);     // neighbouring tiles of a map on one XCD (shared halo lines)
  const int si = __builtin_amdgcn_readfirstlane(tile_seg[bxl]);
  const SepSeg* __restrict__ sg = segs + si;
  const int h = sg->h, w = sg->w, tiles_x = sg->tiles_x, tilesN = sg->tilesN;
  const int wva = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int iw = IPAR > 1 ? (wva >= NWV ? 1 : 0) : 0, wv = wva - iw * NWV, tid = (int)threadIdx.x - iw * NTH;   // image slot; wave / thread within it
  raw_t* xa_s = reinterpret_cast<raw_t*>(smem + Cfg::OFF_XA + iw * Cfg::SLOT_BYTES);
  T* halo = reinterpret_cast<T*>(smem + Cfg::OFF_XA + iw * Cfg::SLOT_BYTES + Cfg::XA_BYTES);
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const GLOBAL T* W = (const GLOBAL T*)sg->wpw;
  const int pgd = PGD == 1 ? 0 : wv / KS, kdw = PGD == 1 ? min(wv, KS - 1) : wv - pgd * KS;   // depthwise phase: this wave's patch group and k-step (waves >= KS * PGD: none)
  const bool dw_wave = wv < KS * PGD, kok_dw = KFULL || kdw * KSTEP + KL * g < CW;
  const int pgm = (HDR || Cfg::PGM == 1) ? 0 : wv / NU, unit = (HDR || Cfg::PGM == 1) ? wv : wv - pgm * NU;   // MFMA phase of a map layer: patch group and output unit

  // ---- this wave's weights -> registers (every load issued before the first use) ----
  raw_t wfr[MAXU * UN][KS];
#pragma unroll
  for (int u = 0; u < MAXU; u++)
#pragma unroll
    for (int j = 0; j < UN; j++) {
      const int nt = (unit + u * NWV) * UN + j;          // (headers: unit = the wave, every wave walks all four patches)
      const GLOBAL T* wr = W + (int64_t)(min(nt, tilesN - 1) * 16 + r) * CW;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const bool kok = KFULL || ks * KSTEP + KL * g < CW;
        wfr[u * UN + j][ks] = *(const GLOBAL raw_t*)(wr + (kok ? ks * KSTEP + KL * g : 0));
        if (!KFULL && !kok) wfr[u * UN + j][ks] = raw_t{};
      }
    }
  // depthwise weights: staged once (swizzled for the packed FMAs); a wave re-reads the 9 x KL of ITS (k-step, lane group) at the
  // start of every image's depthwise phase - 18 LDS reads per image instead of 72 registers held through the MFMA phase
  for (int i = threadIdx.x; i < 9 * CW / 4; i += NTHA) reinterpret_cast<f32x4*>(wdw_s)[i] = F::swz(((const GLOBAL f32x4*)sg->wdw)[i]);
  const float* wdl = wdw_s + (kok_dw ? kdw * KSTEP + KL * g : 0);
  for (int i = threadIdx.x; i < tilesN * 4; i += NTHA) reinterpret_cast<f32x4*>(bias_s)[i] = ((const GLOBAL f32x4*)sg->bias)[i];   // visible after the first barrier

  // ---- geometry: the tile, its 10x10 halo (column-major slots: slot = hx * 10 + hy), the four 4x4 patches = m-tiles ----
  const int t = bxl - sg->tile_begin;
  const int ty = t / tiles_x, tx = t - ty * tiles_x, Y0 = ty * 8, X0 = tx * 8;
  const int b0 = byl * ipb + iw, nall = min(ipb, B - byl * ipb);        // this slot's images: b0, b0 + IPAR, ...
  const int nimg = (nall - iw + IPAR - 1) / IPAR, nloop = (nall + IPAR - 1) / IPAR;   // (every wave runs nloop iterations: the barriers are shared)
  const int img_elems = h * w * CW;
  const GLOBAL T* Ximg = (const GLOBAL T*)sg->src[0] + (int64_t)b0 * img_elems;
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t hoff[NV];
  const int hp0 = tid / CPP, hch = tid - hp0 * CPP;     // pass j: halo pixel hp0 + j * PPJ, channel vector hch
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int hp = hp0 + j * PPJ, hx = hp / 10, hy = hp - hx * 10;
    const int iy = Y0 - 1 + hy, ix = X0 - 1 + hx;
    const bool ok = hp < 100 && iy >= 0 && iy < h && ix >= 0 && ix < w;
    hoff[j] = ok ? (uint32_t)((iy * w + ix) * CW + hch * KL) * ES : OOB;     // out of range: the hardware returns the zero padding
  }
  T* hdst0 = halo + hp0 * HP + hch * KL;
  int toff[9];
#pragma unroll
  for (int q = 0; q < 9; q++) toff[q] = (((r & 3) + q % 3) * 10 + (r >> 2) + q / 3) * HP + (kok_dw ? kdw * KSTEP + KL * g : 0);
  int pokm = 0;                                          // (uniform) bit p: the 4x4 patch p has pixels inside the map
#pragma unroll
  for (int p = 0; p < 4; p++) pokm |= (Y0 + (p >> 1) * 4 < h && X0 + (p & 1) * 4 < w) ? 1 << p : 0;
  raw_t hv[NV];
  auto load_halo = [&](int bi) {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Ximg + (int64_t)bi * IPAR * img_elems), 0, img_elems * ES, 0x00020000);
#pragma unroll
    for (int j = 0; j < NV; j++) hv[j] = __builtin_bit_cast(raw_t, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[j], 0, 0));
  };
  if (nimg > 0) load_halo(0);
  const int N = sg->N, act = sg->act;
  TSTAMP_NOWAIT(1);                    // weights, staging and the first halo issued

#pragma unroll 1
  for (int bi = 0; bi < nloop; bi++) {
    const int b = b0 + bi * IPAR;
    const bool have = bi < nimg;                       // (uniform per wave) this slot has an image in this iteration
#pragma unroll
    for (int j = 0; j < NV; j++) if (have && hp0 + j * PPJ < 100) *reinterpret_cast<raw_t*>(hdst0 + j * PPJ * HP) = hv[j];
#ifdef HEP_TOWER_TRACE
    if (bi == 0) TSTAMP(2);            // first halo arrived and parked
#endif
    __syncthreads();                                   // halo (and, first image, bias and depthwise weights) in LDS
#ifdef HEP_TOWER_TRACE
    if (bi == 0) TSTAMP_NOWAIT(3);
#endif
    if (dw_wave && have) {
      // one row of taps at a time over this wave's patches: 3 x KL weights + PPD accumulators live, not 9 x KL weights
      typename F::Acc acc[PPD];
#pragma unroll
      for (int pi = 0; pi < PPD; pi++) F::zero(acc[pi]);
#pragma unroll
      for (int qr = 0; qr < 3; qr++) {
        f32x4 wa[3], wb[3];
#pragma unroll
        for (int qc = 0; qc < 3; qc++) {
          wa[qc] = *reinterpret_cast<const f32x4*>(wdl + (qr * 3 + qc) * CW);
          wb[qc] = BF16 ? *reinterpret_cast<const f32x4*>(wdl + (qr * 3 + qc) * CW + 4) : wa[qc];
        }
#pragma unroll
        for (int pi = 0; pi < PPD; pi++) {
          const int p = pgd * PPD + pi;
          if ((pokm >> p) & 1) {                                             // (uniform) patch inside the map
            const T* hb = halo + ((p & 1) * 40 + (p >> 1) * 4) * HP;
#pragma unroll
            for (int qc = 0; qc < 3; qc++) F::fma_tap_r(acc[pi], *reinterpret_cast<const raw_t*>(hb + toff[qr * 3 + qc]), wa[qc], wb[qc]);
          }
        }
      }
#pragma unroll
      for (int pi = 0; pi < PPD; pi++) {
        const int p = pgd * PPD + pi;
        if ((pokm >> p) & 1) {
          raw_t xv = F::pack(acc[pi]);
          if (!KFULL && !kok_dw) xv = raw_t{};          // k >= CW: the operand must be exactly zero
          xa_s[(p * KS + kdw) * 64 + lane] = xv;
        }
      }
    }
#ifdef HEP_TOWER_TRACE
    if (bi == 0) TSTAMP(4);            // depthwise done
#endif
    __syncthreads();                                   // operand fragments complete; the halo may be overwritten
#ifdef HEP_TOWER_TRACE
    if (bi == 0) TSTAMP_NOWAIT(5);
#endif
    // the next image's halo flies during the MFMA phase (issued here, not before the depthwise phase: its 7 vectors per lane
    // next to the 72 depthwise weights spilled)
    if (bi + 1 < nimg) load_halo(bi + 1);
    // (fp32, measured and left out: walking the four patches' accumulator chains - 4 * KS dependent exact-fp32 MFMAs each - together:
    //  a map layer alone 27.8 -> 24.7 us, four batches in flight 26.6k frames/s either way; the header units walked together: slower,
    //  53 against 49 us)
#pragma unroll 1
    for (int p = pgm * PPM; p < (have ? (HDR ? 4 : pgm * PPM + PPM) : 0); p++) {
      const int y = Y0 + (p >> 1) * 4 + (r >> 2), x = X0 + (p & 1) * 4 + (r & 3);
      if (!((pokm >> p) & 1)) continue;
      const bool pix_ok = y < h && x < w;
      raw_t xv[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ks++) xv[ks] = xa_s[(p * KS + ks) * 64 + lane];
      if constexpr (!HDR) {
        constexpr int RUN = 4 * Cfg::NTMAP;
        GLOBAL T* O = (GLOBAL T*)sg->out + (int64_t)b * sg->out_bstride + sg->out_off + ((int64_t)y * w + x) * sg->out_rowstride;
        const int ch = g * RUN + unit * UN * 4;
        f32x4 acc[UN];
#pragma unroll
        for (int j = 0; j < UN; j++) acc[j] = *reinterpret_cast<const f32x4*>(bias_s + ch + 4 * j);
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
          for (int j = 0; j < UN; j++) acc[j] = F::mma(wfr[j][ks], xv[ks], acc[j]);
        if (pix_ok && ch < N) {
          float v[4 * UN];
#pragma unroll
          for (int j = 0; j < UN; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) v[4 * j + q] = acc[j][q];
          if (act == ACT_SWISH) {
            swish_n<BF16, 4 * UN>(v);
          } else if (act == ACT_SIGMOID) {
#pragma unroll
            for (int q = 0; q < 4 * UN; q++) v[q] = sigmoid_t<BF16>(v[q]);
          }
          if constexpr (BF16) {
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; e++) pk[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            if (ch + 8 <= N) *(GLOBAL u32x4*)(O + ch) = pk;
            else *(GLOBAL u32x2*)(O + ch) = (u32x2){pk[0], pk[1]};
          } else {
            *(GLOBAL f32x4*)(O + ch) = (f32x4){v[0], v[1], v[2], v[3]};
          }
        }
      } else {
        typedef float __attribute__((ext_vector_type(4), aligned(4))) f32x4_u;
        const int kin = sg->col_kin, kout = sg->col_kout, coff = sg->col_off, nbase = sg->n_base;
        GLOBAL float* O = (GLOBAL float*)sg->out + (int64_t)b * sg->out_bstride + sg->out_off + ((int64_t)y * w + x) * sg->out_rowstride;
        const bool contiguous = kin == kout && coff == 0;
#pragma unroll
        for (int u = 0; u < MAXU; u++) {
          const int nt = wv + u * NWV;
          if (nt >= tilesN) continue;                   // (uniform)
          const int n = nt * 16 + 4 * g;
          f32x4 acc = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
          for (int ks = 0; ks < KS; ks++) acc = F::mma(wfr[u][ks], xv[ks], acc);
          if (pix_ok && n < N) {
            if (act == ACT_SIGMOID) {
#pragma unroll
              for (int q = 0; q < 4; q++) acc[q] = sigmoid_t<BF16>(acc[q]);
            } else if (act == ACT_SWISH) {
#pragma unroll
              for (int q = 0; q < 4; q++) acc[q] = swish_t<BF16>(acc[q]);
            }
            if (contiguous && n + 4 <= N) {
              *(GLOBAL f32x4_u*)(O + nbase + n) = acc;
            } else {
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const int nn = nbase + n + q;
                if (n + q < N) O[(nn / kin) * kout + nn % kin + coff] = acc[q];
              }
            }
          }
        }
      }
    }
#ifdef HEP_TOWER_TRACE
    if (bi == 0) TSTAMP_NOWAIT(6);     // first image: MFMA phase done, stores issued
#endif
  }   // image loop
#ifdef HEP_TOWER_TRACE
  TSTAMP(7);                           // all images done, stores acknowledged
  if (g_tower_trace && lane == 0 && (g_tower_trace_maps != 0) == !HDR) {
    unsigned long long* o = g_tower_trace + ((size_t)(byl * gridDim.x + bxl) * NWV * IPAR + (threadIdx.x >> 6)) * 8;
    for (int i = 0; i < 8; i++) o[i] = stamps[i];
  }
#endif
}

static const int kTowerWidths[] = {64, 88, 112, 160, 224, 288, 384};   // BiFPN widths of phi 0..6 (arch.py)

int tower_supports(int C) {
  for (int cw : kTowerWidths) if (cw == C) return 1;
  return 0;
}

int tower_map_tiles(int C) { return (((C + 15) / 16) + 1) & ~1; }      // n-tiles of a map layer (even)

// widths / dtypes the cooperative form is instantiated for
template <bool BF16, int CW> struct CoopBuilt { static constexpr bool value = BF16 ? (CW >= 160 || CW == 64) : CW == 64; };
int tower_coop_supported(int C, int bf16) {
  if (bf16) return C == 64 || C == 160 || C == 224 || C == 288 || C == 384;
  return C == 64;
}

template <bool BF16, int CW, bool HDR>
static void launch_coop(const SepArgs& a, dim3 grid, hipStream_t s, int ipb) {
  if constexpr (CoopBuilt<BF16, CW>::value) {
    typedef CoopCfg<BF16, CW, HDR> Cfg;
    hipLaunchKernelGGL((tower_coop_kernel<BF16, CW, HDR>), grid, dim3(Cfg::IPAR * Cfg::NWV * 64), Cfg::LDS, s, a.segs, a.tile_seg, a.B, ipb);
  }
}

template <bool BF16, int CW>
static int prepare_coop() {
  if constexpr (CoopBuilt<BF16, CW>::value) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(tower_coop_kernel<BF16, CW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(tower_coop_kernel<BF16, CW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;
  }
  return 0;
}

template <bool BF16, int CW, bool HDR>
static void launch_one(const SepArgs& a, dim3 grid, hipStream_t s, int ipb) {
  typedef TowerCfg<BF16, CW, HDR> Cfg;
  hipLaunchKernelGGL((tower_kernel<BF16, CW, HDR>), grid, dim3(Cfg::NW * 64), Cfg::LDS, s, a.segs, a.tile_seg, a.B, ipb);
}

template <int CW>
static void launch_w(const SepArgs& a, dim3 grid, hipStream_t s, int ipb) {
  const bool hdr = a.direct == 2;
  if (a.coop) {
    if (a.bf16) { if (hdr) launch_coop<true, CW, true>(a, grid, s, ipb); else launch_coop<true, CW, false>(a, grid, s, ipb); }
    else { if (hdr) launch_coop<false, CW, true>(a, grid, s, ipb); else launch_coop<false, CW, false>(a, grid, s, ipb); }
    return;
  }
  if (a.bf16) { if (hdr) launch_one<true, CW, true>(a, grid, s, ipb); else launch_one<true, CW, false>(a, grid, s, ipb); }
  else { if (hdr) launch_one<false, CW, true>(a, grid, s, ipb); else launch_one<false, CW, false>(a, grid, s, ipb); }
}

template <int CW>
static int prepare_w() {
  const void* fns[4] = {reinterpret_cast<const void*>(tower_kernel<true, CW, true>), reinterpret_cast<const void*>(tower_kernel<true, CW, false>),
                        reinterpret_cast<const void*>(tower_kernel<false, CW, true>), reinterpret_cast<const void*>(tower_kernel<false, CW, false>)};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return -1;
  return prepare_coop<true, CW>() | prepare_coop<false, CW>();
}

// raises the dynamic-LDS limit of every instantiation (call once per device)
int tower_prepare(void) {
  return prepare_w<64>() | prepare_w<88>() | prepare_w<112>() | prepare_w<160>() | prepare_w<224>() | prepare_w<288>() | prepare_w<384>();
}

#ifdef HEP_TOWER_TRACE
// profiling build: stamps of the LAST tower launch, [blocks*4 waves][8]; returns the number of waves
extern "C" int hep_dbg_tower_trace(unsigned long long* host, int max_waves, int enable) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)1 << 20;
  if (!buf) { if (hipMalloc((void**)&buf, cap * 8) != hipSuccess) return -1; hipMemset(buf, 0, cap * 8); }
  unsigned long long* p = enable ? buf : nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_tower_trace), &p, sizeof p);
  const int maps = enable == 2;      // enable: 1 = stamps of the header launch, 2 = of the (last) map layer
  hipMemcpyToSymbol(HIP_SYMBOL(g_tower_trace_maps), &maps, sizeof maps);
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)max_waves * 64, hipMemcpyDeviceToHost); }
  return (int)(cap / 8);
}
#endif

void launch_tower(const SepArgs& a, hipStream_t s) {
  // images per workgroup: amortises descriptor, weight staging and the first halo's round trip (2 of a wave's 6 us at two images).  Headers keep
  // >= ~900 workgroups in the launch.  Map layers >= ~400 since round 6 (phi 0 b16: 460 workgroups of four images instead of 920 of two): with
  // the quad-form depthwise an image is 2 us of a wave, so the prologue weighs more - swept separately, sustained / one batch: maps 2 / 4 / 8 / 16
  // images 53.29k / 53.65k / 53.67k / 53.62k and 27.79k / 27.71k / 27.27k / 25.89k; headers 1 / 2 / 4 / 8 / 16 52.45k / 53.29k / 53.20k / 53.12k / 52.77k
  // (profiles/r06/p_tower_images_per_workgroup_sweep.txt).  Bit-identical: only the partition of the images changes.
  const int64_t min_wgs = a.direct == 2 ? 900 : 400;
  int ipb = 1;
  while (ipb < 4 && (int64_t)a.total_tiles * ((a.B + 2 * ipb - 1) / (2 * ipb)) >= min_wgs) ipb *= 2;
  // cooperative form: the prologue (descriptor, 40 weight registers per lane, staging: 2.8 us traced) is paid per workgroup -
  // as many images as still leave ~1.5 workgroups per CU (phi 3 @ 512 b8: 8 -> 430 workgroups, 5.10k -> 5.14k frames/s; fp32 phi 0
  // b16: 4 -> 460; 8 -> 230 measured slower)
  // (round 6, fp32 phi 0 b16 headers: 4 -> 2 images, 460 -> 920 workgroups: 28.98k -> 29.12k frames/s; map layers 2 / 4 / 8 / 16 images
  //  28.89k / 28.98k / 28.84k / 28.03k: four stays)
  if (a.coop) { ipb = 8; while (ipb > 2 && (int64_t)a.total_tiles * ((a.B + ipb - 1) / ipb) < min_wgs) ipb /= 2; }
  ipb = std::min(ipb, a.B);
  const dim3 grid(a.total_tiles, (a.B + ipb - 1) / ipb);
  switch (a.C) {
    case 64: launch_w<64>(a, grid, s, ipb); break;
    case 88: launch_w<88>(a, grid, s, ipb); break;
    case 112: launch_w<112>(a, grid, s, ipb); break;
    case 160: launch_w<160>(a, grid, s, ipb); break;
    case 224: launch_w<224>(a, grid, s, ipb); break;
    case 288: launch_w<288>(a, grid, s, ipb); break;
    case 384: launch_w<384>(a, grid, s, ipb); break;
    default: break;   // the planner only selects this kernel for tower_supports(C)
  }
}
