// Shared helpers for the libmval_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mval_hip.h"

#define MVAL_WAVE 64

void mval_set_error(const char* fmt, ...);

#define MVAL_CHECK_LAUNCH(name)                                                        \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      mval_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return -2;                                                                       \
    }                                                                                  \
  } while (0)

#define MVAL_REQUIRE(cond, ...)                                                        \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      mval_set_error(__VA_ARGS__);                                                     \
      return -1;                                                                       \
    }                                                                                  \
  } while (0)

static inline hipStream_t mval_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- wave-level reductions (64 lanes) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
