// k_pw.hip - dispatch of the pointwise GEMM (kernels: k_pw_impl.h, instantiated per precision in
// k_pw_f32.hip / k_pw_bf16.hip / k_pw_fp8.hip) and the grouped launch of the BiFPN lateral convs.
#include "hep_dev.h"
#include "hep_internal.h"

template <bool BF16> struct Frag;
template <> struct Frag<true> { typedef u32x4 raw; static constexpr int KSTEP = 32, KLANE = 8; };
template <> struct Frag<false> { typedef f32x4 raw; static constexpr int KSTEP = 16, KLANE = 4; };

template <int PREC> void launch_pw_prec(const PwArgs&, hipStream_t);
extern template void launch_pw_prec<0>(const PwArgs&, hipStream_t);
extern template void launch_pw_prec<1>(const PwArgs&, hipStream_t);
extern template void launch_pw_prec<2>(const PwArgs&, hipStream_t);

// squeeze-excite prologue variant of a launch (also names the device function, hep_kernel_symbol)
int pw_se_variant(const PwArgs& a) {
  if (a.sq <= 0 || a.act == ACT_SWISH) return 0;          // (sq is set by the planner for project convs only)
  if (a.se_from_tensor) return 3;
  return (a.K > 256 || a.sqp / (a.bf16 ? 8 : 4) > 2) ? 2 : 1;
}

#ifdef HEP_PW_TRACE
unsigned long long* g_pw_trace_host = nullptr;
extern "C" int hep_dbg_pw_trace(unsigned long long* host, int max_waves, int enable) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)1 << 20;
  if (!buf) { if (hipMalloc((void**)&buf, cap * 8) != hipSuccess) return -1; hipMemset(buf, 0, cap * 8); }
  g_pw_trace_host = enable ? buf : nullptr;
  if (host) { hipDeviceSynchronize(); hipMemcpy(host, buf, (size_t)max_waves * 64, hipMemcpyDeviceToHost); }
  return (int)(cap / 8);
}
#endif

void launch_pw(const PwArgs& a, hipStream_t s) {
#ifdef HEP_WITH_FP8
  if (a.fp8) { launch_pw_prec<2>(a, s); return; }
#endif
  if (a.bf16) launch_pw_prec<1>(a, s);
  else launch_pw_prec<0>(a, s);
}

// ------------------------------------------------------------------------------------------------
// Grouped launch for the BiFPN lateral convs of cell 0 (p3/p4/p5_down_channel, their *_2 twins and
// p5_to_p6; reference efficientdet/model.py:107-140): six GEMMs with 1 Ki .. 16 Ki rows each that only
// depend on the backbone taps.  As separate launches they cost six kernel boundaries for ~2 us of work.
// One workgroup = one 16-row strip of one (segment, image); its 4 waves take the n-tiles round-robin
// (fpn width 64 = 4 n-tiles -> one each).  No activation, BN folded into W / bias.
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void pw_group_kernel(PwgArgs a) {
  typedef Vec8<BF16> V;
  typedef typename V::elem T;
  typedef Frag<BF16> F;
  typedef typename F::raw raw_t;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
  int si = 0;                                    // segment of this strip (uniform; nseg <= 8)
#pragma unroll
  for (int i = 1; i < PWG_MAX; i++) if (i < a.nseg && (int)blockIdx.x >= a.seg[i].blk_begin) si = i;
  // (run-time indexing of the by-value table would spill it: select the fields with a uniform loop instead)
  const void* Ap = a.seg[0].A; const void* Wp = a.seg[0].W; const float* bias = a.seg[0].bias; void* out = a.seg[0].out;
  int HW = a.seg[0].HW, K = a.seg[0].K, N = a.seg[0].N, tilesN = a.seg[0].tilesN, blk0 = a.seg[0].blk_begin;
#pragma unroll
  for (int i = 1; i < PWG_MAX; i++)
    if (i == si) { Ap = a.seg[i].A; Wp = a.seg[i].W; bias = a.seg[i].bias; out = a.seg[i].out; HW = a.seg[i].HW; K = a.seg[i].K; N = a.seg[i].N; tilesN = a.seg[i].tilesN; blk0 = a.seg[i].blk_begin; }
  const int b = blockIdx.y;
  const int row = (blockIdx.x - blk0) * 16 + r;          // pixel of this lane's MFMA column
  const bool rok = row < HW;
  const T* A = reinterpret_cast<const T*>(Ap) + ((int64_t)b * HW + (rok ? row : 0)) * K + F::KLANE * g;
  const T* W = reinterpret_cast<const T*>(Wp);
  T* O = reinterpret_cast<T*>(out) + ((int64_t)b * HW + row) * N;
  const int ksteps = (K + F::KSTEP - 1) / F::KSTEP;
  for (int nt = wave; nt < tilesN; nt += 4) {
    const T* Wr = W + (int64_t)(nt * 16 + r) * K + F::KLANE * g;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ks++) {
      raw_t xa = {}, wf = {};
      if (ks * F::KSTEP + F::KLANE * g < K) { wf = *reinterpret_cast<const raw_t*>(Wr + ks * F::KSTEP); if (rok) xa = *reinterpret_cast<const raw_t*>(A + ks * F::KSTEP); }
      if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xa[q], acc, 0, 0, 0);
      }
    }
    const int n = nt * 16 + 4 * g;
    if (rok && n < N) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = acc[q] + bv[q];
      V::store4(O, n, v);
    }
  }
}

void launch_pwg(const PwgArgs& a, hipStream_t s) {
  const dim3 grid(a.strips_per_image, a.B);
  if (a.bf16) hipLaunchKernelGGL(pw_group_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(pw_group_kernel<false>, grid, dim3(256), 0, s, a);
}
