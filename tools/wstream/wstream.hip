// wstream.hip - step 0 of the image-resident late-backbone kernel: how fast can ONE CU pull a weight
// stream that every workgroup of the launch reads (blocks 11-15 of EfficientNet-B0: 4.7 MB of bf16 weights), with the
// fragments going straight to registers in MFMA operand order (host-packed: one wave instruction = 1 KB contiguous) and
// MM MFMAs (16x16x32 bf16) per fragment riding on them?
//   hipcc -O3 --offload-arch=gfx950 wstream.hip -o wstream;  ./wstream            (sweep)
// Prints GB/s per workgroup (= per CU: one workgroup per CU) for G workgroups x W waves x U loads in flight per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// chunks of 1 KB (64 lanes x 16 B); wave w of W takes batches of U consecutive chunks round-robin.  The next batch is
// requested before the current one is consumed (two batches of registers).
template <int U, int MM>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ w, long nchunks, float* out, int per_wg_offset) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, W = blockDim.x >> 6;
  const u32x4* base = w + (long)per_wg_offset * blockIdx.x * nchunks * 64;
  const long nb = nchunks / U;            // batches
  f32x4 acc[MM > 0 ? MM : 1];
  for (int i = 0; i < (MM > 0 ? MM : 1); i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a[MM > 0 ? MM : 1];
  for (int i = 0; i < (MM > 0 ? MM : 1); i++) for (int j = 0; j < 8; j++) a[i][j] = (__bf16)(float)(lane + i + j);
  unsigned x = 0;
  u32x4 cur[U], nxt[U];
  long b = wv;
  if (b < nb) for (int u = 0; u < U; u++) cur[u] = __builtin_nontemporal_load(base + (b * U + u) * 64 + lane);
  for (; b < nb; b += W) {
    const long bn = b + W < nb ? b + W : b;       // (the last batch is re-read: unconditional loads)
    for (int u = 0; u < U; u++) nxt[u] = __builtin_nontemporal_load(base + (bn * U + u) * 64 + lane);
    __builtin_amdgcn_sched_barrier(0);      // (without it the compiler sinks every load to its use: one round trip per fragment)
    for (int u = 0; u < U; u++) {
      if (MM > 0) {
        const bf16x8 f = __builtin_bit_cast(bf16x8, cur[u]);
        for (int m = 0; m < MM; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, a[m], acc[m], 0, 0, 0);
      } else x ^= cur[u][0] ^ cur[u][1] ^ cur[u][2] ^ cur[u][3];
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
  }
  float s = (float)x;
  for (int i = 0; i < (MM > 0 ? MM : 1); i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int U, int MM> static float run(const u32x4* w, long nchunks, float* out, int G, int Wv, int per_wg, hipStream_t st) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((stream_kernel<U, MM>), dim3(G), dim3(Wv * 64), 0, st, w, nchunks, out, per_wg);
  const int R = 10;
  CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < R; i++) hipLaunchKernelGGL((stream_kernel<U, MM>), dim3(G), dim3(Wv * 64), 0, st, w, nchunks, out, per_wg);
  CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / R * 1e3f;    // us per launch
}

int main(int argc, char** argv) {
  const long bytes = argc > 1 ? atol(argv[1]) : 4718592;       // 4.5 MiB: the five blocks' weights
  const long nchunks = bytes / 1024 / 16 * 16;
  u32x4* w; float* out;
  const int maxG = 256;
  CHECK(hipMalloc(&w, (size_t)nchunks * 1024 * maxG)); CHECK(hipMemset(w, 1, (size_t)nchunks * 1024 * maxG));
  CHECK(hipMalloc(&out, 4096));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  printf("weight stream %ld KB per workgroup; GB/s per workgroup (one workgroup per CU)\n", nchunks);
  printf("%-28s %4s %3s %3s %3s %9s %9s %9s\n", "source", "G", "W", "U", "MM", "us", "GB/s/CU", "TB/s chip");
  for (int per_wg = 0; per_wg < 2; per_wg++)
    for (int G : {1, 8, 16, 32, 64, 256})
      for (int Wv : {4, 8, 16})
        for (int cfg = 0; cfg < 6; cfg++) {
          float us = 0; int U = 0, MM = 0;
          switch (cfg) {
            case 0: U = 4; MM = 0; us = run<4, 0>(w, nchunks, out, G, Wv, per_wg, st); break;
            case 1: U = 8; MM = 0; us = run<8, 0>(w, nchunks, out, G, Wv, per_wg, st); break;
            case 2: U = 16; MM = 0; us = run<16, 0>(w, nchunks, out, G, Wv, per_wg, st); break;
            case 3: U = 8; MM = 4; us = run<8, 4>(w, nchunks, out, G, Wv, per_wg, st); break;
            case 4: U = 8; MM = 8; us = run<8, 8>(w, nchunks, out, G, Wv, per_wg, st); break;
            case 5: U = 16; MM = 4; us = run<16, 4>(w, nchunks, out, G, Wv, per_wg, st); break;
          }
          const double gbs = (double)nchunks * 1024 / (us * 1e-6) / 1e9;
          printf("%-28s %4d %3d %3d %3d %9.1f %9.1f %9.2f\n", per_wg ? "own buffer per workgroup" : "one buffer for all", G, Wv, U, MM, us, gbs, gbs * G / 1e3);
        }
  return 0;
}
