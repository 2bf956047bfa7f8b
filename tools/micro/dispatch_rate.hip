// Micro-benchmark: how long the GPU takes just to dispatch N workgroups of 256 threads with a given LDS allocation
// (empty kernel) -- the floor under the per-tile conv kernels.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_rate.hip -o tools/micro/bin/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void empty_kernel(float* out, int flag) {
  extern __shared__ char smem[];
  if (flag) out[blockIdx.x] = smem[threadIdx.x];
}
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds[] = {0, 18 * 1024, 37 * 1024, 64 * 1024};
  const int wgs[] = {1024, 8192, 16384, 32768};
  for (int l : lds) {
    hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int n : wgs) {
      for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(empty_kernel, dim3(n), dim3(256), l, 0, out, 0);
      hipEventRecord(e0);
      for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(empty_kernel, dim3(n), dim3(256), l, 0, out, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("lds %2d KB  %5d workgroups: %6.1f us per launch  (%.2f ns per workgroup)\n", l / 1024, n, ms * 1e3 / 20, ms * 1e6 / 20 / n);
    }
  }
  return 0;
}
