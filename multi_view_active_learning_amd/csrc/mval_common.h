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
// 64-bit keys of a heat-map, kept by the waves that STORE the map (csrc/conv_p2.hip, conv_common.h): high word = the
// value as an order-preserving unsigned (NaN above +inf, -0 == +0: torch.argmax's order, utils/evaluation.py:13-30),
// low word = ~flat index, so among equal values the FIRST index wins.  A map has MVAL_ARGMAX_SLOTS partial keys,
// zeroed before the layer runs (every stored value maps to a key > 0); a wave stores the best key of its part of the map
// in its own slot -- plain stores: device-scope atomics on one address cost ~0.2 us each, 311 k of them tripled the
// layer's 29 us -- and folds with atomicMax only when a map has more partials than slots.  mval_argmax_from_keys reduces
// the rows and turns them into key-points.
// Layout [image][slot][joint]: the keys a wave leaves for one tile (its couts) sit next to each other -- [image][joint][slot]
// made every one of them its own 8-byte write transaction, half as many again as the layer's own stores.
__device__ __forceinline__ void mval_argmax_key_put(unsigned long long* keys, int image, int joint, int joints, int slot, int total,
                                                    unsigned long long key) {
  unsigned long long* p = keys + ((int64_t)image * MVAL_ARGMAX_SLOTS + (total <= MVAL_ARGMAX_SLOTS ? slot : slot % MVAL_ARGMAX_SLOTS)) * joints + joint;
  if (total <= MVAL_ARGMAX_SLOTS) *p = key;
  else atomicMax(p, key);
}
__device__ __forceinline__ unsigned long long mval_argmax_key(float v, unsigned flat_index) {
  const unsigned b = __float_as_uint(v);
  const unsigned o = (v != v) ? 0xffffffffu : (b == 0x80000000u) ? b : (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - flat_index);
}
// max over each 16-lane row of the wave, in every lane of the row: four DPP steps (xor 1, xor 2 inside the quads, mirror
// inside the half rows, mirror inside the row) -- register traffic only; the ds_bpermute butterfly is four dependent LDS
// round trips, ~1 us per tile in the heat-map layer's epilogue
template <int CTRL>
__device__ __forceinline__ unsigned long long mval_key_dpp(unsigned long long k) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xf, 0xf, false);
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long mval_key_row16_max(unsigned long long k) {
  k = mval_key_dpp<0xB1>(k);   // quad_perm [1, 0, 3, 2]
  k = mval_key_dpp<0x4E>(k);   // quad_perm [2, 3, 0, 1]
  k = mval_key_dpp<0x141>(k);  // row_half_mirror
  k = mval_key_dpp<0x140>(k);  // row_mirror
  return k;
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

// ReLU that hands a NaN on, as torch.relu does (fmaxf returns the OTHER operand for a NaN and would turn a diverged model's
// NaNs into zeros): IEEE 754-2019 maximum, one v_maximum3_f32 on gfx950.
__device__ __forceinline__ float mval_relu(float v) { return __builtin_elementwise_maximum(v, 0.f); }
