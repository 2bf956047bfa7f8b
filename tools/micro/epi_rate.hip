// Micro-benchmark: cost of the pieces of a P2 epilogue granule (csrc/conv_p2.h) per wave, one or two waves per SIMD:
//   0  BN + ReLU only (2 packed-ish FMAs + 4 max per float4)
//   1  + p2_split (4 x cvt f32->f16, cvt back, sub, cvt)
//   2  + the two v_permlane32_swap
//   3  + ds_write_b128 of the granule
//   4  2 + raw_buffer_store_b128 (+ s_nop 1)
//   5  p2_join of a residual granule (2 swaps + 8 cvt + 4 add) on top of 2
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/epi_rate.hip -o /tmp/epi_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned* sink, int iters) {
  __shared__ u32x4 lds[256 * 4];
  f32x4 acc[4];
  for (int j = 0; j < 4; j++) acc[j] = (f32x4){threadIdx.x * 0.01f + j, 1.f + j, 2.f, 3.f};
  const f32x4 sc = {1.0001f, 0.9999f, 1.0002f, 0.9998f}, sh = {0.1f, -0.1f, 0.2f, -0.2f};
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(sink, 0, 1u << 26, 0x00020000);
  unsigned keep = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f32x4 v = acc[j] * sc + sh;
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      if (KIND == 5) {
        const u32x4 g = lds[(threadIdx.x + j * 256) & 1023];
        const auto r0 = __builtin_amdgcn_permlane32_swap(g.x, g.z, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(g.y, g.w, false, false);
        const u32x2 rh = {r0[0], r1[0]}, rl = {r0[1], r1[1]};
        v += __builtin_convertvector(__builtin_bit_cast(f16x4, rh), f32x4) + __builtin_convertvector(__builtin_bit_cast(f16x4, rl), f32x4);
      }
      if (KIND >= 1) {
        const f16x4 h = __builtin_convertvector(v, f16x4);
        const f16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
        const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
        if (KIND >= 2) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
          const u32x4 g = {s0[0], s1[0], s0[1], s1[1]};
          if (KIND == 3) lds[(threadIdx.x + j * 256) & 1023] = g;
          else if (KIND == 4) {
            __builtin_amdgcn_raw_buffer_store_b128(g, br, ((blockIdx.x * 256 + threadIdx.x) * 4 + j) * 16u, 0, 0);
            asm volatile("s_nop 1");
          } else keep ^= g.x ^ g.y ^ g.z ^ g.w;
        } else keep ^= hu.x ^ hu.y ^ lu.x ^ lu.y;
      }
      acc[j] = v * 0.5f + acc[j] * 0.25f;  // (keeps the chain data-dependent across iterations)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)keep;
}

template <int KIND>
static void run(const char* name, float* out, unsigned* sink, int wgs) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, out, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-44s %d wave(s)/SIMD: %7.1f ns per granule per wave (%5.0f cycles at 2.4 GHz)\n", name, wgs / 256, ms * 1e6 / (iters * 4.0), ms * 1e6 / (iters * 4.0) * 2.4);
  }
}

int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  unsigned* sink; hipMalloc(&sink, 1u << 26);
  for (int wgs = 256; wgs <= 512; wgs += 256) {
    run<0>("BN + ReLU", out, sink, wgs);
    run<1>("+ split (h, l)", out, sink, wgs);
    run<2>("+ 2 x v_permlane32_swap", out, sink, wgs);
    run<3>("+ ds_write_b128", out, sink, wgs);
    run<4>("split + swaps + buffer_store_b128 + s_nop", out, sink, wgs);
    run<5>("residual granule: LDS read, 2 swaps, join", out, sink, wgs);
  }
  return 0;
}
