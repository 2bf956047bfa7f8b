// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_16x16x16_bf16 (one wave per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void k(float* out, int iters) {
  f32x4 acc[4] = {};
  bf16x8 a8, b8;
  bf16x4 a4, b4;
  for (int i = 0; i < 8; i++) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
  for (int i = 0; i < 4; i++) { a4[i] = a8[i]; b4[i] = b8[i]; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[j], 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int kind = 0; kind < 2; kind++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) {
        double n = 256.0 * 4 * iters * 4;  // MFMAs chip-wide (one wave per SIMD)
        double flop = n * 2.0 * 16 * 16 * (kind == 0 ? 32 : 16);
        printf("%s: %.3f ms, %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n", kind == 0 ? "16x16x32" : "16x16x16", ms,
               ms * 1e6 / (iters * 4.0), flop / ms / 1e9);
      }
    }
  }
  return 0;
}
