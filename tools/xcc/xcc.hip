// xcc.hip - which XCD does workgroup id L land on?  (hipcc --offload-arch=gfx950 xcc.hip -o xcc; ./xcc)  HW_REG_XCC_ID (hwreg 20) per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 4); unsigned h[4096];
  for (int threads : {64, 1024}) for (int n : {9, 48, 64}) {
    hipLaunchKernelGGL(k, dim3(n), dim3(threads), threads == 1024 ? 150 * 1024 : 0, 0, d);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    printf("%d workgroups x %d threads: raw XCC_ID register per workgroup id:", n, threads);
    for (int i = 0; i < n; i++) printf(" %x", h[i]);
    printf("\n");
  }
  return 0;
}
