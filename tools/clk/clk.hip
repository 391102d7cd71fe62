// shader-clock probe: ns per dependent v_fma_f32 of one wave (s_memrealtime, 100 MHz), alone / after load / beside load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void calib(float* out, unsigned long long* t, int iters) {
  float x = out[threadIdx.x];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int q = 0; q < 16; q++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { t[0] = t0; t[1] = t1; }
}
__global__ void burn(float* out, int iters) {
  float x = out[threadIdx.x + blockIdx.x * blockDim.x], y = x + 1.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int q = 0; q < 16; q++) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(y)); }
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = x + y;
}
static double run_calib(float* d, unsigned long long* dt, int iters, hipStream_t s) {
  hipLaunchKernelGGL(calib, dim3(1), dim3(64), 0, s, d, dt, iters);
  hipStreamSynchronize(s);
  unsigned long long h[2]; hipMemcpy(h, dt, 16, hipMemcpyDeviceToHost);
  return (double)(h[1] - h[0]) * 10.0 / ((double)iters * 16.0);       // ns per dependent fma
}
int main() {
  float* d; unsigned long long* dt; hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24); hipMalloc(&dt, 16);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  const int iters = 1 << 14;       // 262144 fmas ~ 0.5 ms
  printf("first launch          : %.3f ns per dependent v_fma_f32\n", run_calib(d, dt, iters, s1));
  sleep(2);
  printf("after 2 s idle        : %.3f ns\n", run_calib(d, dt, iters, s1));
  printf("again at once         : %.3f ns\n", run_calib(d, dt, iters, s1));
  hipLaunchKernelGGL(burn, dim3(4096), dim3(256), 0, s2, d + 4096, 20000); hipStreamSynchronize(s2);
  printf("after a 4096x256 burn : %.3f ns\n", run_calib(d, dt, iters, s1));
  hipLaunchKernelGGL(burn, dim3(1024), dim3(256), 0, s2, d + 4096, 200000);
  usleep(2000);
  printf("beside a running burn : %.3f ns\n", run_calib(d, dt, iters, s1));
  hipStreamSynchronize(s2);
  for (int k = 0; k < 3; k++) { usleep(300); printf("short gaps (300 us)   : %.3f ns\n", run_calib(d, dt, 1 << 10, s1)); }
  return 0;
}
