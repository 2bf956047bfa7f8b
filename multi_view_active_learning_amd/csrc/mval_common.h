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

// ---- arg-max keys (decode from the heat-map layer's epilogue) ---------------------------------------------------------
// One 64-bit key per heat-map, kept with atomicMax by the workgroups that STORE the map (csrc/conv_p2.hip,
// conv_common.h): high word = the value as an order-preserving unsigned (NaN above +inf, -0 == +0: torch.argmax's
// order, utils/evaluation.py:13-30), low word = ~flat index, so among equal values the FIRST index wins.  The keys are
// zeroed before the layer runs (every stored value maps to a key > 0); mval_argmax_from_keys turns them into key-points.
__device__ __forceinline__ unsigned long long mval_argmax_key(float v, unsigned flat_index) {
  const unsigned b = __float_as_uint(v);
  const unsigned o = (v != v) ? 0xffffffffu : (b == 0x80000000u) ? b : (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - flat_index);
}
// max over the 2^steps-lane groups of a wave (xor butterfly from `first` down to 1)
__device__ __forceinline__ unsigned long long mval_key_group_max(unsigned long long k, int first) {
  for (int o = first; o > 0; o >>= 1) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)k, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(k >> 32), o, 64);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    k = other > k ? other : k;
  }
  return k;
}
