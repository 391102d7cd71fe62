// k_chain.hip - the small pyramid levels of a BiFPN cell boundary as ONE LDS-resident chain per image (bf16 and fp32 sessions).
//
// Between two visits of the 16x16 / 32x32 levels a BiFPN walks up to five nodes on maps of 8x8, 4x4 and 2x2 pixels
// (reference efficientdet/model.py:212-264: p5_out -> p6_out -> p7_out -> next cell's p6_up -> p5_up; in cell 0
// also the two zero-padded max-pools that make p6_in / p7_in, model.py:198-203).  Each node is
//   swish(sum_i w_i * gather_i)  ->  depthwise 3x3 SAME  ->  pointwise 1x1 (+bias, BN folded)
// on at most 64 pixels x 64 channels: microseconds of arithmetic.  As a chain inside k_sep.hip (mode 2) every node
// still paid three to four dependent global round trips (descriptor, gather, weights, store drain): 4.6 us per node,
// 23 us per chain, 16 workgroups busy - the worst line of the round-1 roofline table (0.0065 of HBM peak).
//
// Here one workgroup per image
//   1. fetches EVERYTHING the chain will need in one burst: the depthwise / pointwise weights and biases of all its
//      nodes and every external input map (a map that needs the 3x3/2 max-pool is pooled while it is loaded), all into
//      LDS - one memory round trip for the whole chain;
//   2. runs the nodes out of LDS: gathers read LDS map slots (external inputs or earlier nodes' outputs), results are
//      written to their slot and streamed to global memory without waiting (later launches read them from there);
//   3. three workgroup barriers per node, no global round trip until the kernel ends.
// Arithmetic, rounding points and operation order are those of sep_kernel (k_sep.hip), so the stage-parity tests
// gate it like every other BiFPN node.  Widths other than what fits in LDS keep the k_sep.hip path.
//
// fp32 sessions (round 4): maps and weights are twice the bytes, so (a) the map slots are shared by liveness (host:
// Planner::add_chain - a node's output takes the slot of a map nobody reads any more) and (b) the node weights are STREAMED
// (template parameter): LDS holds two nodes' weights; the next node's are requested at the head of a node by LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave instruction, no registers) into the buffer the previous node has left.  The chains of
// an fp32 session ran as sep_kernel<false, 2> before: 35 us per five-node chain against 20 us here in bf16.
#include <type_traits>

#include "hep_dev.h"
#include "hep_internal.h"

#ifndef CHAIN_THREADS
#define CHAIN_THREADS 1024
#endif
#define CHAIN_WAVES (CHAIN_THREADS / 64)

// profiling build (make trace): wave 0 of every workgroup stamps s_memrealtime (100 MHz) at the phase boundaries
#ifdef HEP_MBF_TRACE
__device__ unsigned long long* g_chain_trace = nullptr;
#define CSTAMP() do { if (g_chain_trace && lane == 0 && nst < 60) st_buf[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)   // (every wave stamps)
#else
#define CSTAMP()
#endif

namespace {

// 8 consecutive channels as they lie in memory: 16 bytes (bf16) or two 16-byte vectors (fp32)
template <bool BF16> struct Raw8;
template <> struct Raw8<true> {
  u32x4 v;
  static __device__ __forceinline__ Raw8 load(const void* p) { Raw8 r; r.v = *reinterpret_cast<const u32x4*>(p); return r; }
  __device__ __forceinline__ void store(void* p) const { *reinterpret_cast<u32x4*>(p) = v; }
  __device__ __forceinline__ void zero() { v = (u32x4){0, 0, 0, 0}; }
  __device__ __forceinline__ void unpack(float o[8]) const {
#pragma unroll
    for (int i = 0; i < 4; i++) { o[2 * i] = __uint_as_float(v[i] << 16); o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
  }
};
template <> struct Raw8<false> {
  f32x4 a, b;
  static __device__ __forceinline__ Raw8 load(const void* p) { Raw8 r; r.a = reinterpret_cast<const f32x4*>(p)[0]; r.b = reinterpret_cast<const f32x4*>(p)[1]; return r; }
  __device__ __forceinline__ void store(void* p) const { reinterpret_cast<f32x4*>(p)[0] = a; reinterpret_cast<f32x4*>(p)[1] = b; }
  __device__ __forceinline__ void zero() { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }
  __device__ __forceinline__ void unpack(float o[8]) const {
#pragma unroll
    for (int i = 0; i < 4; i++) { o[i] = a[i]; o[4 + i] = b[i]; }
  }
};

// 8 channels of a map slot at pixel (y, x) as floats; zero outside the map (SAME padding of the pool / the depthwise halo).
// The load is unconditional (clamped address) and the zero is a select: conditional loads would each sit in a basic
// block of their own and serialise into one memory round trip per tap.
template <bool BF16, typename T> __device__ __forceinline__ Raw8<BF16> slot_raw(const T* slot, int h, int w, int C, int y, int x, int c0) {
  const bool ok = y >= 0 && y < h && x >= 0 && x < w;
  const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
  Raw8<BF16> r = Raw8<BF16>::load(slot + (int64_t)(yc * w + xc) * C + c0);
  if (!ok) r.zero();
  return r;
}
template <bool BF16, typename T> __device__ __forceinline__ void slot_load(const T* slot, int h, int w, int C, int y, int x, int c0, float v[8]) {
  slot_raw<BF16>(slot, h, w, C, y, x, c0).unpack(v);
}

// zero-padded 3x3 stride-2 max-pool of a slot at output pixel (y, x) (utils_extra.py:72-86: the pad value takes part);
// all nine loads are in flight before the first is consumed
// (fp32: two passes of four channels - nine 32-byte vectors in flight per lane spilled)
__device__ __forceinline__ void slot_pool4(const float* slot, int h, int w, int C, int pad, int y, int x, int c0, float* m) {
  f32x4 raw[9];
#pragma unroll
  for (int q = 0; q < 9; q++) {
    const int yy = 2 * y - pad + q / 3, xx = 2 * x - pad + q % 3;
    const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
    raw[q] = *reinterpret_cast<const f32x4*>(slot + (int64_t)(min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)) * C + c0);
    if (!ok) raw[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int q = 0; q < 9; q++)
#pragma unroll
    for (int c = 0; c < 4; c++) m[c] = q == 0 ? raw[q][c] : fmaxf(m[c], raw[q][c]);
}
template <bool BF16, typename T> __device__ __forceinline__ void slot_pool(const T* slot, int h, int w, int C, int pad, int y, int x, int c0, float m[8]) {
  if constexpr (!BF16) { slot_pool4(slot, h, w, C, pad, y, x, c0, m); slot_pool4(slot, h, w, C, pad, y, x, c0 + 4, m + 4); return; }
  Raw8<BF16> raw[9];
#pragma unroll
  for (int q = 0; q < 9; q++) raw[q] = slot_raw<BF16>(slot, h, w, C, 2 * y - pad + q / 3, 2 * x - pad + q % 3, c0);
#pragma unroll
  for (int q = 0; q < 9; q++) {
    float v[8];
    raw[q].unpack(v);
#pragma unroll
    for (int c = 0; c < 8; c++) m[c] = q == 0 ? v[c] : fmaxf(m[c], v[c]);
  }
}

}  // namespace

// WMODE: where a node's weights live.  0: all nodes' weights resident in LDS (bf16, width 64); 1: two nodes' weights in LDS, the next
// node's streamed in by LDS-DMA under the running node (fp32, width 64); 2: only the depthwise weights and biases in LDS, the pointwise
// weight fragments of an (m-tile, n-tile) pair requested together straight from global memory (L2) in front of its MFMAs (width 160:
// one node's pointwise weights are 60 KB and do not fit next to three 8x8x160 maps).
template <bool BF16, int WMODE>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_kernel(ChainArgs a) {
  constexpr bool STREAM = WMODE == 1, WGLOBAL = WMODE == 2;
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef Raw8<BF16> R8;
  typedef typename std::conditional<BF16, u32x4, f32x4>::type frag_t;
  constexpr int PAD = BF16 ? 8 : 4, KSTEP = BF16 ? 32 : 16, KLANE = BF16 ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, a.lds_bytes);
  T* slots = reinterpret_cast<T*>(smem);
  unsigned char* wreg = smem + a.off_w;
  T* halo = reinterpret_cast<T*>(smem + a.off_halo);
  T* atile = reinterpret_cast<T*>(smem + a.off_atile);
  const int C = a.C, CG = C >> 3, CH = C + PAD;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
  int cgsh = 0; while ((1 << cgsh) < CG) cgsh++;
  const int cg = tid & ((1 << cgsh) - 1), prow = tid >> cgsh, pstride = CHAIN_THREADS >> cgsh;
#ifdef HEP_MBF_TRACE
  unsigned long long st_buf[60]; int nst = 0;
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
  CSTAMP();
  kernarg_warm<(int)sizeof(ChainArgs)>();      // (hep_dev.h: the prologue below fetched its eight argument lines one miss after the other)

  // ---- 1. one burst: node descriptors, all weights and all external maps -> LDS.  Every load is issued before the
  //         first LDS store that waits for it, so the whole prologue is about one memory round trip. ----
  {
    // the chain's weights: one contiguous blob already in the LDS layout (per node [9*C] f32 depthwise | [C] f32 bias |
    // [C][C+PAD] bf16 pointwise rows)
    constexpr int WB = 4096 / CHAIN_THREADS;
    // (WGLOBAL: the first 10 * C floats of every node - depthwise weights and bias - packed node after node)
    const int svecs = (10 * C * 4) >> 4;
    const int wvecs = WGLOBAL ? a.nconv * svecs : (int)(((size_t)(STREAM ? 1 : a.nconv) * a.wnode_bytes) >> 4);
    auto wsrc_of = [&](int i) -> const u32x4* {
      if constexpr (WGLOBAL) { const int nd_ = i / svecs, o = i - nd_ * svecs; return reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.wblob) + (size_t)nd_ * a.wnode_bytes) + o; }
      else return reinterpret_cast<const u32x4*>(a.wblob) + i;
    };
    u32x4 wv[WB];
#pragma unroll
    for (int j = 0; j < WB; j++) { const int i = tid + j * CHAIN_THREADS; if (i < wvecs) wv[j] = *wsrc_of(i); }
    // external maps that are copied as they are (their own resolution)
    constexpr int EPF = BF16 ? CH_MAX_EXT : 4;          // maps whose first row of vectors is prefetched (fp32: 8 registers each)
    R8 ev[EPF];
#pragma unroll
    for (int e = 0; e < EPF; e++) {
      const ChainExt& x = a.ext[e];
      ev[e].zero();
      if (e < a.next && x.kind != SRC_DOWN && cg < CG && prow < x.h * x.w)
        ev[e] = R8::load(reinterpret_cast<const T*>(x.src) + ((int64_t)b * x.sh * x.sw + prow) * C + cg * 8);
    }
    // The node descriptors are read by scalar loads at the head of every node (below): a scalar-cache miss there is ~0.4 us in
    // front of every node, for all sixteen waves (per-wave stamps: waves without an item spend 0.5 us per node).  Every descriptor
    // line is touched HERE - every load of the prologue has been issued, the first LDS store below waits for them anyway - so that
    // the later loads hit the scalar cache.
    {
      typedef const __attribute__((address_space(4))) uint32_t* cptr;
      const int nbytes = a.nnodes * (int)sizeof(ChainNode);
      cptr np = (cptr)a.nodes;
#pragma unroll
      for (int o = 0; o < CH_MAX_NODES * (int)sizeof(ChainNode); o += 256) {
        if (o < nbytes) {                                       // (uniform)
          cptr q = np + o / 4;
          uint32_t t0, t1, t2, t3;
          asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(q) : "memory");
        }
      }
    }
#pragma unroll
    for (int j = 0; j < WB; j++) { const int i = tid + j * CHAIN_THREADS; if (i < wvecs) reinterpret_cast<u32x4*>(wreg)[i] = wv[j]; }
    for (int i = tid + WB * CHAIN_THREADS; i < wvecs; i += CHAIN_THREADS) reinterpret_cast<u32x4*>(wreg)[i] = *wsrc_of(i);   // (more nodes / wider maps)
#pragma unroll
    for (int e = 0; e < CH_MAX_EXT; e++) {
      const ChainExt& x = a.ext[e];
      if (e >= a.next || cg >= CG) continue;
      T* dst = slots + x.off;
      const T* src = reinterpret_cast<const T*>(x.src) + (int64_t)b * x.sh * x.sw * C;
      if (x.kind != SRC_DOWN) {
        if (e < EPF && prow < x.h * x.w) ev[e < EPF ? e : 0].store(dst + (int64_t)prow * C + cg * 8);
        for (int p = prow + (e < EPF ? pstride : 0); p < x.h * x.w; p += pstride) R8::load(src + (int64_t)p * C + cg * 8).store(dst + (int64_t)p * C + cg * 8);
      } else {       // pooled while it is loaded (zero-padded 3x3 / 2 max-pool); optionally also a map of its own (p6_in)
        for (int p = prow; p < x.h * x.w; p += pstride) {
          float m[8];
          slot_pool<BF16>(src, x.sh, x.sw, C, x.pool_pad, p / x.w, p % x.w, cg * 8, m);
          V::store(dst, (int64_t)p * C + cg * 8, m);                                    // exact: the values are session-dtype numbers already
          if (x.store) V::store(x.store, ((int64_t)b * x.h * x.w + p) * C + cg * 8, m);
        }
      }
    }
  }
  __syncthreads();
  CSTAMP();

  // ---- 2. the nodes, out of LDS (descriptors included); pixel
  //         indices are split with a reciprocal multiply (a run-time integer division is ~30 instructions on the one
  //         item a lane has per phase). ----
#pragma unroll 1      // a real loop: the body runs once per node, unrolled it would be fetched cold every time (measured +20 %)
  for (int n = 0; n < a.nnodes; n++) {
    // the descriptor is copied to registers once and made wave-uniform (scalar registers): read field by field from LDS
    // inside the phases it cost a load + wait per use (the gather phase took 2 us of a 3 us node)
    // (the descriptor arrives as a few wide scalar loads from constant memory; copied from LDS and made uniform dword by
    //  dword it was a chain of ~25 LDS round trips at the head of every node, paid by all 16 waves: per-wave stamps showed
    //  0.5-1.0 us per node in waves that had no item at all)
    ChainNode nd;
    {
      typedef const __attribute__((address_space(4))) uint32_t* cptr;
      cptr sp = (cptr)(a.nodes + n);
      uint32_t* dp = reinterpret_cast<uint32_t*>(&nd);
#pragma unroll
      for (int i = 0; i < (int)(sizeof(ChainNode) / 4); i++) dp[i] = sp[i];
    }
    const int h = nd.h, w = nd.w, hw = h * w;
    const uint32_t w_rcp = nd.w_rcp, hs_rcp = nd.hs_rcp;     // (host-made reciprocals: the 64-bit divisions were part of a ~0.6 us fixed cost per node)
    T* oslot = slots + nd.out_off;
    if (nd.pool_only) {          // p7_in = pool(p6_in): a map of its own, no convolution
      const ChainSrc& s0 = nd.src[0];
      if (cg < CG)
        for (int p = prow; p < hw; p += pstride) {
          float m[8];
          const int py = (int)__umulhi((uint32_t)p, w_rcp);
          slot_pool<BF16>(slots + s0.off, s0.sh, s0.sw, C, nd.pool_pad, py, p - py * w, cg * 8, m);
          V::store(oslot, (int64_t)p * C + cg * 8, m);
          V::store(nd.out, ((int64_t)b * hw + p) * C + cg * 8, m);
        }
      __syncthreads();
      continue;
    }
    const float* wdw_s = reinterpret_cast<const float*>(wreg + (WGLOBAL ? (size_t)nd.widx * (10 * C * 4) : (size_t)(STREAM ? (nd.widx & 1) : nd.widx) * a.wnode_bytes));
    const float* bias_s = wdw_s + 9 * C;
    const T* wpw_s = reinterpret_cast<const T*>(bias_s + C);                       // (resident / streamed forms)
    const T* wpw_g = reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(a.wblob) + (size_t)nd.widx * a.wnode_bytes + (size_t)10 * C * 4);
    // streamed weights: the NEXT node's go straight from global memory into the other LDS buffer (LDS-DMA, no registers: held in
    // registers under the node they spilled); that buffer was last read by the previous node, which every wave has left
    // (its closing barrier).  1 KB per wave instruction; the host pads a node's weights to whole KB.
    if constexpr (STREAM) {
      if (nd.widx + 1 < a.nconv) {
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wblob) + (size_t)(nd.widx + 1) * a.wnode_bytes;
        unsigned char* wdst = wreg + (size_t)((nd.widx + 1) & 1) * a.wnode_bytes;
        for (int c = wave; c * 1024 < a.wnode_bytes; c += CHAIN_WAVES)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + c * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(wdst + c * 1024), 16, 0, 0);
      }
    }
    const int HS = w + 2;
    // fused + swished input with its one-pixel zero halo
    if (cg < CG)
      for (int pos = prow; pos < (h + 2) * HS; pos += pstride) {
        const int hy = (int)__umulhi((uint32_t)pos, hs_rcp);
        const int y = hy - 1, x = pos - hy * HS - 1;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (y >= 0 && y < h && x >= 0 && x < w) {
#pragma unroll
          for (int i = 0; i < HEP_MAX_SRC; i++) {
            if (i >= nd.nsrc) break;
            const ChainSrc& s = nd.src[i];
            float t[8];
            if (s.kind == SRC_DOWN) slot_pool<BF16>(slots + s.off, s.sh, s.sw, C, nd.pool_pad, y, x, cg * 8, t);
            else if (s.kind == SRC_UP) slot_load<BF16>(slots + s.off, s.sh, s.sw, C, y >> 1, x >> 1, cg * 8, t);
            else slot_load<BF16>(slots + s.off, s.sh, s.sw, C, y, x, cg * 8, t);
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = fmaf(s.fw, t[c], v[c]);
          }
          swish_n<BF16, 8>(v);
        }
        V::store(halo, (int64_t)pos * CH + cg * 8, v);
      }
    CSTAMP();
    if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the node's weights (requested one node ago) has landed
    __syncthreads();
    CSTAMP();
    // depthwise 3x3 -> MFMA operand tile [pixels][C]; rows beyond the map are zeroed (their products are discarded)
    if (cg < CG)
      for (int p = prow; p < ((hw + 15) & ~15); p += pstride) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p < hw) {
          const int py = (int)__umulhi((uint32_t)p, w_rcp), px = p - py * w;
#pragma unroll
          for (int q = 0; q < 9; q++) {
            float hv[8];
            V::load(halo, (int64_t)((py + q / 3) * HS + px + q % 3) * CH + cg * 8, hv);
            const f32x4* wp = reinterpret_cast<const f32x4*>(wdw_s + q * C + cg * 8);
            const f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
            for (int c = 0; c < 4; c++) { acc[c] = fmaf(hv[c], w0[c], acc[c]); acc[4 + c] = fmaf(hv[4 + c], w1[c], acc[4 + c]); }
          }
        }
        V::store(atile, (int64_t)p * CH + cg * 8, acc);
      }
    CSTAMP();
    __syncthreads();
    CSTAMP();
    // pointwise: D[n, pixel] = W[n, :] . tile[pixel, :]; (m-tile, n-tile) pairs dealt to the waves; lane ends with 4
    // consecutive channels of one pixel -> its slot in LDS
    const int mt_n = (hw + 15) >> 4, nt_n = (C + 15) >> 4, ksteps = (C + KSTEP - 1) / KSTEP;      // (widths like 88: the last n-tile is half used)
    for (int pair = wave; pair < mt_n * nt_n; pair += CHAIN_WAVES) {
      const int mt = udiv_rcp(pair, a.nt_rcp), nt = pair - mt * nt_n;      // (host-made reciprocal: a 64-bit division sat here)
      const int m = mt * 16 + r;
      // (WGLOBAL: at widths that are no multiple of 16 the last n-tile's rows past C would lie behind the node's [C][C + pad] weights
      //  in the blob - for the last node behind the blob itself; their products are never stored: clamp the row)
      const T* wrow = (WGLOBAL ? wpw_g : wpw_s) + (int64_t)(WGLOBAL ? min(nt * 16 + r, C - 1) : nt * 16 + r) * CH + KLANE * g;
      const T* arow = atile + (int64_t)m * CH + KLANE * g;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (WGLOBAL) {
        // every k-step's weight fragment of the pair is requested before the first MFMA (one L2 round trip per pair, not per
        // k-step); unconditional loads on clamped offsets, k >= C meets a zeroed activation fragment
        constexpr int KSG = 6;                                                   // C <= 192 (bf16) / 96 (fp32): host-checked
        frag_t wfr[KSG];
#pragma unroll
        for (int ks = 0; ks < KSG; ks++) wfr[ks] = *reinterpret_cast<const frag_t*>(wrow + min(ks * KSTEP, C - KLANE - KLANE * g));
#pragma unroll
        for (int ks = 0; ks < KSG; ks++) {
          if (ks < ksteps) {
            frag_t xa = {};
            if (ks * KSTEP + KLANE * g < C) xa = *reinterpret_cast<const frag_t*>(arow + ks * KSTEP);
            if constexpr (BF16) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[ks]), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
            else {
#pragma unroll
              for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[ks][q], xa[q], acc, 0, 0, 0);
            }
          }
        }
      } else
      for (int ks = 0; ks < ksteps; ks++) {
        frag_t wf = {}, xa = {};
        if (ks * KSTEP + KLANE * g < C) { wf = *reinterpret_cast<const frag_t*>(wrow + ks * KSTEP); xa = *reinterpret_cast<const frag_t*>(arow + ks * KSTEP); }
        if constexpr (BF16) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
        else {
#pragma unroll
          for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xa[q], acc, 0, 0, 0);      // exact fp32; lane group g supplies k = 16 ks + 4 g + q
        }
      }
      const int nn = nt * 16 + 4 * g;
      if (m < hw && nn < C) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + nn);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = acc[q] + bias[q];
        V::store4(oslot, (int64_t)m * C + nn, v);
      }
    }
    CSTAMP();
    __syncthreads();
    CSTAMP();
    // the finished map leaves as full 16-byte vectors; nobody in this launch reads it from global memory
    if (cg < CG)
      for (int p = prow; p < hw; p += pstride) R8::load(oslot + (int64_t)p * C + cg * 8).store(reinterpret_cast<T*>(nd.out) + ((int64_t)b * hw + p) * C + cg * 8);
    CSTAMP();
  }
#ifdef HEP_MBF_TRACE
  if (g_chain_trace && lane == 0) { unsigned long long* o = g_chain_trace + ((size_t)b * CHAIN_WAVES + wave) * 64; for (int i = 0; i < 60; i++) o[i] = i < nst ? st_buf[i] : 0; o[63] = (unsigned long long)nst; o[62] = __builtin_amdgcn_s_memtime() - clk0; }
#endif
}

#ifdef HEP_MBF_TRACE
extern "C" int hep_dbg_chain_trace(unsigned long long* host, int nblocks, int enable) {
  static unsigned long long* buf = nullptr;
  if (!buf) { if (hipMalloc((void**)&buf, (size_t)4096 * CHAIN_WAVES * 64 * 8) != hipSuccess) return -1; hipMemset(buf, 0, (size_t)4096 * CHAIN_WAVES * 64 * 8); }
  unsigned long long* p = enable ? buf : nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &p, sizeof p);
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)nblocks * CHAIN_WAVES * 64 * 8, hipMemcpyDeviceToHost); }
  return 0;
}
#endif

template <bool BF16, int WMODE> static int chain_prep_one() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<BF16, WMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess ? 0 : -1;
}
int chain_prepare(void) {
  return chain_prep_one<true, 0>() | chain_prep_one<true, 1>() | chain_prep_one<true, 2>() | chain_prep_one<false, 0>() | chain_prep_one<false, 1>() | chain_prep_one<false, 2>();
}

void launch_chain(const ChainArgs& a_, hipStream_t s) {
  ChainArgs a = a_;
  a.nt_rcp = rcp_u32((uint32_t)((a.C + 15) >> 4));
  const dim3 g(a.B), b(CHAIN_THREADS);
  if (a.bf16) {
    if (a.stream_w == 2) hipLaunchKernelGGL((chain_kernel<true, 2>), g, b, a.lds_bytes, s, a);
    else if (a.stream_w == 1) hipLaunchKernelGGL((chain_kernel<true, 1>), g, b, a.lds_bytes, s, a);
    else hipLaunchKernelGGL((chain_kernel<true, 0>), g, b, a.lds_bytes, s, a);
  } else {
    if (a.stream_w == 2) hipLaunchKernelGGL((chain_kernel<false, 2>), g, b, a.lds_bytes, s, a);
    else if (a.stream_w == 1) hipLaunchKernelGGL((chain_kernel<false, 1>), g, b, a.lds_bytes, s, a);
    else hipLaunchKernelGGL((chain_kernel<false, 0>), g, b, a.lds_bytes, s, a);
  }
}
