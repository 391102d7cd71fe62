#!/bin/bash
# TIMING-ONLY build (results are WRONG): prices an image-resident late-backbone kernel for the THROUGHPUT regime before it is written
# (VERDICT r04 item 2).  In the captured graph every launch whose name starts with one of the prefixes in HEP_EXP_SKIP
# (e.g. "b11.,b12.,b13.,b14.,b15.") is dropped, and in place of the first one ONE stand-in launch runs: HEP_EXP_DUMMY="G,T" =
# G workgroups of 1024 threads that stream the session's weights through their CU (fragment-order 16-byte loads + MFMAs) until T us
# have passed.  G = 0 / unset: nothing in their place (the upper bound: the blocks cost nothing).
#   tools/exp/late_pricing_build.sh && HEP_LIB=$PWD/hmd_ego_pose_amd/libhep_latex.so HEP_EXP_SKIP=b11.,b12.,b13.,b14.,b15. HEP_EXP_DUMMY=16,150 python bench.py ...
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
T="$R/hmd_ego_pose_amd/csrc_latex"; rm -rf "$T"; cp -r "$R/hmd_ego_pose_amd/csrc" "$T"; rm -rf "$T"/build*
python3 - "$T" <<'PY'
import sys
p = sys.argv[1] + "/hep_api.cpp"
s = open(p).read()
old = "        for (size_t i = 1; i < s->ops.size(); i++) launch_op(*s, s->lane_ops[l][i], s->lane_count(batch, l), s->stream, nullptr, nullptr);\n"
assert old in s
new = """        {
          const char* sk = getenv("HEP_EXP_SKIP"); const char* dm = getenv("HEP_EXP_DUMMY");
          int G = 0, Tus = 0; if (dm) sscanf(dm, "%d,%d", &G, &Tus);
          bool placed = false;
          for (size_t i = 1; i < s->ops.size(); i++) {
            bool skip = false;
            if (sk) { std::string list = sk; size_t a = 0; while (a < list.size()) { size_t b = list.find(',', a); if (b == std::string::npos) b = list.size(); const std::string pre = list.substr(a, b - a); if (!pre.empty() && s->ops[i].name.compare(0, pre.size(), pre) == 0) skip = true; a = b + 1; } }
            if (!skip) { launch_op(*s, s->lane_ops[l][i], s->lane_count(batch, l), s->stream, nullptr, nullptr); continue; }
            if (!placed && G > 0) hipLaunchKernelGGL(exp_dummy_kernel, dim3(G), dim3(1024), 0, s->stream, (const exp_u32x4*)s->d_weights, (long)(4608), (float*)s->d_arena, (unsigned long long)Tus * 100ull);
            placed = true;
          }
        }
"""
s = s.replace(old, new)
kern = """
typedef __attribute__((ext_vector_type(8))) __bf16 exp_bf16x8;
typedef __attribute__((ext_vector_type(4))) float exp_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned exp_u32x4;
__global__ __launch_bounds__(1024) void exp_dummy_kernel(const exp_u32x4* __restrict__ w, long nchunks, float* out, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, W = blockDim.x >> 6;
  exp_f32x4 acc[4]; exp_bf16x8 a[4];
  for (int i = 0; i < 4; i++) { acc[i] = (exp_f32x4){0.f, 0.f, 0.f, 0.f}; for (int j = 0; j < 8; j++) a[i][j] = (__bf16)(float)(lane + i + j); }
  const long nb = nchunks / 8;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    for (long b = wv; b < nb; b += W) {
      exp_u32x4 c[8];
      for (int u = 0; u < 8; u++) c[u] = w[(b * 8 + u) * 64 + lane];
      for (int u = 0; u < 8; u++) { const exp_bf16x8 f = __builtin_bit_cast(exp_bf16x8, c[u]); for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, a[m], acc[m], 0, 0, 0); }
      if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
    }
  }
  float sum = 0.f; for (int i = 0; i < 4; i++) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sum == 12345.678f) out[threadIdx.x] = sum;
}
"""
s = s.replace("namespace hep {\n", kern + "\nnamespace hep {\n", 1)
open(p, "w").write(s)
PY
make -C "$T" -j8 OUT="$R/hmd_ego_pose_amd/libhep_latex.so" OBJDIR="$T/build" ROOT="$R" > /dev/null
rm -rf "$T"; ls -la "$R/hmd_ego_pose_amd/libhep_latex.so"
