// k_heads.hip - the five head towers DEPTH-FIRST: one launch for `D x [depthwise 3x3 -> pointwise 1x1 + per-level BatchNorm ->
// swish] -> header depthwise 3x3 -> header pointwise (+ sigmoid for the class scores) -> [B, N_anchors, K]` of all five nets on all
// five pyramid levels (reference efficientdet/model.py:361-417 Regressor / Classifier, hmdegopose/model.py:55-228 RotationNet /
// TranslationNet / HandNet; iter 0).  bf16 sessions, BiFPN width 64.
//
// Launch by launch (k_tower.hip) the towers are D launches that each read and write every map of every net (28 MB per layer at
// phi 0 b16) plus one header launch: waves live 8.4 us of which 5.4 wait for their 6x6 halo from HBM (NOTEBOOK.md section 2,
// "Round 5").  Here a workgroup (16 waves) owns one 16x16 OUTPUT tile of one (net, level, image): the input region with a halo
// of D + 1 pixels (24x24 at D = 3) is loaded ONCE into LDS and the layers run in place - a wave computes whole m-tiles (16 pixels:
// depthwise taps from LDS straight into the MFMA operand layout, the 64x64 pointwise weights of the layer as eight fragments in
// registers, fetched in operand order from a host-packed blob), keeps its results in registers across a barrier and writes them
// back over its inputs; only pixels INSIDE the image are computed and the rest of the buffer stays zero, which is the zero
// padding of every layer.  The tower activations never reach HBM; the header outputs leave as in k_tower.hip.
// Arithmetic and its order are k_tower.hip's (depthwise: zero + nine fused multiply-adds in tap order, rounded to bf16; pointwise:
// bias + two MFMA k-steps; swish; bf16): the two plans agree bit for bit (tests/test_gpu_parity.py).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "hep_dev.h"
#include "hep_internal.h"

// per-wave phase stamps (make trace: -DHEP_HEADS_TRACE), read back with hep_dbg_heads_trace() (tools/trace_heads.py): 0 start, 1 region loaded,
// 2 + i layer i done (its results written back), 6 header operands parked, 7 end; 8: item id
#ifdef HEP_HEADS_TRACE
__device__ unsigned long long* g_heads_trace = nullptr;
#define HSTAMP(i) do { if (g_heads_trace && lane == 0 && blockIdx.x == 0) g_heads_trace[((size_t)blockIdx.y * 16 + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HSTAMP(i)
#endif

namespace {
typedef __attribute__((ext_vector_type(2))) float hf2;
constexpr int HF_C = 64, HF_P = HF_C + 8;          // channels, LDS pixel pitch (bf16 elements: 144 bytes)
constexpr int HF_T = 16;                           // output tile side
constexpr int HF_MTW = 3;                          // m-tiles per wave and layer (26 x 26 region at D = 4: 36 m-tiles on 16 waves)

struct HAcc { hf2 a[4]; };                          // channels (0,2) (1,3) (4,6) (5,7) of a lane's eight: k_tower.hip's Frag<true>
__device__ __forceinline__ void hf_fma_tap(HAcc& c, const u32x4& x, const float* w /* 8 swizzled weights in LDS */) {
  const f32x4 wa = reinterpret_cast<const f32x4*>(w)[0], wb = reinterpret_cast<const f32x4*>(w)[1];
  const hf2 e0 = {__uint_as_float(x[0] << 16), __uint_as_float(x[1] << 16)}, o0 = {__uint_as_float(x[0] & 0xffff0000u), __uint_as_float(x[1] & 0xffff0000u)};
  const hf2 e1 = {__uint_as_float(x[2] << 16), __uint_as_float(x[3] << 16)}, o1 = {__uint_as_float(x[2] & 0xffff0000u), __uint_as_float(x[3] & 0xffff0000u)};
  c.a[0] = __builtin_elementwise_fma(e0, (hf2){wa[0], wa[1]}, c.a[0]);
  c.a[1] = __builtin_elementwise_fma(o0, (hf2){wa[2], wa[3]}, c.a[1]);
  c.a[2] = __builtin_elementwise_fma(e1, (hf2){wb[0], wb[1]}, c.a[2]);
  c.a[3] = __builtin_elementwise_fma(o1, (hf2){wb[2], wb[3]}, c.a[3]);
}
__device__ __forceinline__ u32x4 hf_pack(const HAcc& c) {
  return (u32x4){pack_bf16x2(c.a[0][0], c.a[1][0]), pack_bf16x2(c.a[0][1], c.a[1][1]), pack_bf16x2(c.a[2][0], c.a[3][0]), pack_bf16x2(c.a[2][1], c.a[3][1])};
}
__device__ __forceinline__ f32x4 hf_mma(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
}  // namespace

__global__ __launch_bounds__(1024) void heads_kernel(HeadsArgs a_by_value) {
  // (the item table is indexed at run time: read through the kernel-argument segment, not a by-value copy in scratch - k_late.hip)
  const HeadsArgs& a = *(const HeadsArgs*)(const __attribute__((address_space(4))) HeadsArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)a_by_value;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  HEP_POISON(smem, a.lds_bytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
  // grid = (images, items): the dispatcher hands workgroups out x-fastest, so item 0 of EVERY image starts before item 1 of any - the
  // table is sorted by decreasing work (the hand net's 16x16 tiles first)
  const int b = blockIdx.x;
  const HeadItem& it = *reinterpret_cast<const HeadItem*>(a.blob + (size_t)blockIdx.y * sizeof(HeadItem));
  const int D = a.D, RP = HF_T + 2 * (D + 1), hw = it.hw, lvl = it.level;
  const int by0 = it.y0 - (D + 1), bx0 = it.x0 - (D + 1);            // image coordinates of buffer position (0, 0)
  bf16_t* buf = reinterpret_cast<bf16_t*>(smem);                      // [RP * RP][HF_P]
  float* wdw_s = reinterpret_cast<float*>(smem + a.off_wdw);          // [3][9][64]: the running layer's table, header 0's, header 1's (swizzled)
  const unsigned char* blob = a.blob;

  HSTAMP(0);
  // ---- input region -> LDS (zero outside the image), depthwise tables of layer 0 and of the headers ----
  {
    const bf16_t* src = reinterpret_cast<const bf16_t*>(a.feat[lvl]) + (size_t)b * hw * hw * HF_C;
    const float rp_inv = __builtin_amdgcn_rcpf((float)RP);
    for (int u = tid; u < RP * RP * 8; u += 1024) {
      const int pos = u >> 3, v = u & 7, ry = udiv_f(pos, RP, rp_inv), rx = pos - ry * RP;
      const int iy = by0 + ry, ix = bx0 + rx;
      const bool in = iy >= 0 && iy < hw && ix >= 0 && ix < hw;
      u32x4 x = *reinterpret_cast<const u32x4*>(src + (size_t)((in ? iy : 0) * hw + (in ? ix : 0)) * HF_C + v * 8);
      if (!in) x = (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(buf + pos * HF_P + v * 8) = x;
    }
    if (tid < 144) reinterpret_cast<f32x4*>(wdw_s)[tid] = *reinterpret_cast<const f32x4*>(blob + it.off_layers + 8192 + 256 + (size_t)tid * 16);
    for (int h = 0; h < it.nhdr; h++)
      if (tid >= 256 * (h + 1) && tid < 256 * (h + 1) + 144)
        reinterpret_cast<f32x4*>(wdw_s + (1 + h) * 576)[tid - 256 * (h + 1)] = *reinterpret_cast<const f32x4*>(blob + it.off_hdr[h] + (size_t)(tid - 256 * (h + 1)) * 16);
  }
  __syncthreads();
  HSTAMP(1);

  // ---- tower layers, in place ----
  // the layer's eight weight fragments (n-tile, k-step) and this lane's bias values: operand order, 1 KB per wave instruction;
  // the NEXT layer's are requested as soon as this layer's last MFMA has consumed them (they land under the two barriers)
  u32x4 wf[4][2]; f32x4 bias[4];
  auto load_layer = [&](int i) {
    const unsigned char* lw = blob + it.off_layers + (size_t)i * (8192 + 256 + 2304);
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++) wf[nt][ks] = *reinterpret_cast<const u32x4*>(lw + ((size_t)(nt * 2 + ks) * 64 + lane) * 16);
      bias[nt] = *reinterpret_cast<const f32x4*>(lw + 8192 + (size_t)(nt * 16 + 4 * g) * 4);
    }
  };
  load_layer(0);
  for (int i = 0; i < D; i++) {
    const unsigned char* lw = blob + it.off_layers + (size_t)i * (8192 + 256 + 2304);
    // pixels of this layer's output: the region shrinks by one ring per layer and is cut to the image
    const int r0 = max(i + 1, -by0), r1 = min(RP - i - 1, hw - by0), q0 = max(i + 1, -bx0), q1 = min(RP - i - 1, hw - bx0);
    const int wdt = q1 - q0, npx = (r1 - r0) * wdt, mtiles = (npx + 15) >> 4;
    const float wdt_inv = __builtin_amdgcn_rcpf((float)wdt);
    u32x2 res[HF_MTW][4]; int dst[HF_MTW];
#pragma unroll
    for (int s = 0; s < HF_MTW; s++) {
      const int mt = wave + 16 * s;
      dst[s] = -1;
      if (mt < mtiles) {                                         // (wave-uniform)
        const int p = min(mt * 16 + r, npx - 1), pr = udiv_f(p, wdt, wdt_inv), py = r0 + pr, px = q0 + p - pr * wdt;
        if (mt * 16 + r < npx) dst[s] = (py * RP + px) * HF_P;
        const bf16_t* ctr = buf + (py * RP + px) * HF_P + 8 * g;
        u32x4 xa[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          HAcc acc; for (int e = 0; e < 4; e++) acc.a[e] = (hf2){0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 9; q++)
            hf_fma_tap(acc, *reinterpret_cast<const u32x4*>(ctr + ((q / 3 - 1) * RP + q % 3 - 1) * HF_P + ks * 32), wdw_s + q * 64 + ks * 32 + 8 * g);
          xa[ks] = hf_pack(acc);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
          f32x4 acc = bias[nt];
          acc = hf_mma(wf[nt][0], xa[0], acc);
          acc = hf_mma(wf[nt][1], xa[1], acc);
          float v[4] = {acc[0], acc[1], acc[2], acc[3]};
          swish_n<true, 4>(v);
          res[s][nt] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      }
    }
    if (i + 1 < D) load_layer(i + 1);
    __syncthreads();                                               // every tap of this layer has been read
#pragma unroll
    for (int s = 0; s < HF_MTW; s++)
      if (dst[s] >= 0) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) *reinterpret_cast<u32x2*>(buf + dst[s] + nt * 16 + 4 * g) = res[s][nt];
      }
    if (i + 1 < D && tid < 144) reinterpret_cast<f32x4*>(wdw_s)[tid] = *reinterpret_cast<const f32x4*>(lw + (8192 + 256 + 2304) + 8192 + 256 + (size_t)tid * 16);
    __syncthreads();
    HSTAMP(2 + i);
  }

  // ---- headers.  (a) wave = m-tile of the output tile: the header's depthwise conv -> operand fragments -> LDS;  (b) wave = (n-tile,
  //      quarter of the m-tiles): its two weight fragments come from global memory ONCE per item and meet the parked operands -
  //      with one m-tile per wave and every n-tile streamed by every wave the hand header (36 n-tiles) pulled 1.1 MB of fragments
  //      through the CU's L1 per workgroup ----
  {
    const int r0 = max(D + 1, -by0), r1 = min(D + 1 + HF_T, hw - by0), q0 = max(D + 1, -bx0), q1 = min(D + 1 + HF_T, hw - bx0);
    const int wdt = q1 - q0, npx = (r1 - r0) * wdt, mtiles = (npx + 15) >> 4;
    const float wdt_inv = __builtin_amdgcn_rcpf((float)wdt);
    u32x4* xa_s = reinterpret_cast<u32x4*>(smem + a.off_wdw + 3 * 576 * 4);       // [16 m-tiles][2 k-steps][64 lanes]
    typedef float __attribute__((ext_vector_type(4), aligned(4))) f32x4_u;
    for (int h = 0; h < it.nhdr; h++) {
      if (wave < mtiles) {
        const int p = min(wave * 16 + r, npx - 1), pr = udiv_f(p, wdt, wdt_inv), py = r0 + pr, px = q0 + p - pr * wdt;
        const bf16_t* ctr = buf + (py * RP + px) * HF_P + 8 * g;
        const float* wd = wdw_s + (1 + h) * 576;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          HAcc acc; for (int e = 0; e < 4; e++) acc.a[e] = (hf2){0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 9; q++)
            hf_fma_tap(acc, *reinterpret_cast<const u32x4*>(ctr + ((q / 3 - 1) * RP + q % 3 - 1) * HF_P + ks * 32), wd + q * 64 + ks * 32 + 8 * g);
          xa_s[(wave * 2 + ks) * 64 + lane] = hf_pack(acc);
        }
      }
      __syncthreads();
      if (h == 0) HSTAMP(6);
      const int ntl = it.hdr_ntiles[h], N = it.hdr_N[h], kin = it.hdr_kin[h], kout = it.hdr_kout[h], coff = it.hdr_off[h], act = it.hdr_act[h];
      const unsigned char* hb = blob + it.off_hdr[h] + 2304;                    // bias [ntl * 16] f32, then the fragments [ntl][2][64][8] bf16
      const unsigned char* hwf = hb + (size_t)ntl * 64;
      float* Ob = a.out[it.hdr_out[h]] + ((size_t)b * a.num_anchors + a.level_off[lvl]) * kout;
      const bool contiguous = kin == kout && coff == 0;
      const int mq = (mtiles + 3) >> 2, nitems = ntl * 4;                       // m-tiles per quarter
      for (int item = wave; item < nitems; item += 16) {
        const int nt = item >> 2, mt0 = (item & 3) * mq, mt1 = min(mt0 + mq, mtiles);
        if (mt0 >= mt1) continue;
        const u32x4 w0 = *reinterpret_cast<const u32x4*>(hwf + ((size_t)(nt * 2) * 64 + lane) * 16), w1 = *reinterpret_cast<const u32x4*>(hwf + ((size_t)(nt * 2 + 1) * 64 + lane) * 16);
        const f32x4 bs = *reinterpret_cast<const f32x4*>(hb + (size_t)(nt * 16 + 4 * g) * 4);
        const int n = nt * 16 + 4 * g;
        for (int mt = mt0; mt < mt1; mt++) {
          f32x4 acc = bs;
          acc = hf_mma(w0, xa_s[(mt * 2) * 64 + lane], acc);
          acc = hf_mma(w1, xa_s[(mt * 2 + 1) * 64 + lane], acc);
          const int pp = mt * 16 + r;
          if (pp < npx && n < N) {
            const int pr = udiv_f(pp, wdt, wdt_inv), y = by0 + r0 + pr, x = bx0 + q0 + pp - pr * wdt;
            float* O = Ob + ((size_t)y * hw + x) * 9 * kout;
            if (act == ACT_SIGMOID) {
#pragma unroll
              for (int q = 0; q < 4; q++) acc[q] = sigmoid_t<true>(acc[q]);
            } else if (act == ACT_SWISH) {
#pragma unroll
              for (int q = 0; q < 4; q++) acc[q] = swish_t<true>(acc[q]);
            }
            if (contiguous && n + 4 <= N) *reinterpret_cast<f32x4_u*>(O + n) = acc;
            else {
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const int nn = n + q;
                if (nn < N) O[(nn / kin) * kout + nn % kin + coff] = acc[q];
              }
            }
          }
        }
      }
      if (h + 1 < it.nhdr) __syncthreads();                                      // the parked operands are rewritten by the next header
    }
  }
  HSTAMP(7);
}

#ifdef HEP_HEADS_TRACE
extern "C" int hep_dbg_heads_trace(unsigned long long* host, int max_words, int enable) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)256 * 16 * 16;
  if (!buf) { if (hipMalloc((void**)&buf, cap * 8) != hipSuccess) return -1; }
  if (enable) hipMemset(buf, 0, cap * 8);
  unsigned long long* p = enable ? buf : nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(g_heads_trace), &p, sizeof p);
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)std::min<size_t>(max_words, cap) * 8, hipMemcpyDeviceToHost); }
  return (int)cap;
}
#endif

int heads_fused_supported(int C, int depth, int bf16) { return bf16 == 1 && C == HF_C && depth >= 1 && depth <= 4; }

int heads_lds_bytes(int depth, int* off_wdw) {
  const int rp = HF_T + 2 * (depth + 1);
  const int o = (rp * rp * HF_P * 2 + 15) & ~15;
  if (off_wdw) *off_wdw = o;
  return o + 3 * 576 * 4 + 16 * 2 * 1024;      // + the header operand fragments of 16 m-tiles
}

int heads_prepare(void) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(heads_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : -1;
}

void launch_heads(const HeadsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(heads_kernel, dim3(a.B, a.nitems), dim3(1024), (size_t)a.lds_bytes, s, a);
}
