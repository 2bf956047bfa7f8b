// Keypoint decode kernels: hard arg-max (K9) and soft-arg-max (K10).
// HBM-bound: each heat-map is read exactly once (J*hh*wh*4 bytes per frame x view).
// One 64-lane wave per map, float4 loads, wave reduction with first-index tie-break.
#include "mval_common.h"

// order: NaN is the maximum (torch.argmax semantics); ties -> lowest flat index
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
  bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (vn) return i < bi;
  return (v > bv) || (v == bv && i < bi);
}

__global__ __launch_bounds__(256) void argmax_decode_kernel(
    const float* __restrict__ hm, const uint8_t* __restrict__ valid, int64_t* __restrict__ kp2d,
    int64_t n_maps, int V, int J, int npix, int stride, int split_width) {
  const int lane = threadIdx.x & 63;
  const int64_t map = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (map >= n_maps) return;
  const int j = (int)(map % J);
  const int64_t b = map / ((int64_t)V * J);
  if (valid && !valid[b * J + j]) {
    if (lane == 0) { kp2d[map * 2] = 0; kp2d[map * 2 + 1] = 0; }
    return;
  }
  const float* p = hm + map * (int64_t)npix;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  if ((npix & 3) == 0 && ((uintptr_t)p & 15) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = lane; i < (npix >> 2); i += 64) {
      float4 q = p4[i];
      int base = i << 2;
      if (better(q.x, base, bv, bi)) { bv = q.x; bi = base; }
      if (better(q.y, base + 1, bv, bi)) { bv = q.y; bi = base + 1; }
      if (better(q.z, base + 2, bv, bi)) { bv = q.z; bi = base + 2; }
      if (better(q.w, base + 3, bv, bi)) { bv = q.w; bi = base + 3; }
    }
  } else {
    for (int i = lane; i < npix; i += 64) {
      float q = p[i];
      if (better(q, i, bv, bi)) { bv = q; bi = i; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(bv, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) {
    if (bi == 0x7fffffff) bi = 0;  // all -inf: first element
    kp2d[map * 2] = (int64_t)(bi % split_width) * stride;
    kp2d[map * 2 + 1] = (int64_t)(bi / split_width) * stride;
  }
}

extern "C" int mval_argmax_decode(const float* heatmaps, const uint8_t* valid, int64_t* kp2d, int B, int V,
                                  int J, int hh, int wh, int stride, int split_width, void* stream) {
  MVAL_REQUIRE(B >= 0 && V > 0 && J > 0 && hh > 0 && wh > 0 && split_width > 0, "mval_argmax_decode: bad dims");
  int64_t n_maps = (int64_t)B * V * J;
  if (n_maps == 0) return 0;
  // an all -inf map must still pick index 0: handled by better() never firing -> bi stays max; fix below
  dim3 grid((unsigned)((n_maps + 3) / 4));
  hipLaunchKernelGGL(argmax_decode_kernel, grid, dim3(256), 0, mval_stream(stream), heatmaps, valid, kp2d, n_maps, V,
                     J, hh * wh, stride, split_width);
  MVAL_CHECK_LAUNCH("mval_argmax_decode");
  return 0;
}

// ---- key-points from the arg-max keys the heat-map layer's epilogue kept (mval_common.h) ------------------------------------
// one wave per map: its MVAL_ARGMAX_SLOTS partial keys -> the largest
__global__ __launch_bounds__(256) void argmax_from_keys_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                               int64_t* __restrict__ kp2d, int64_t n_maps, int V, int J, int stride,
                                                               int split_width) {
  const int lane = threadIdx.x & 63;
  const int64_t map = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (map >= n_maps) return;
  const int j = (int)(map % J);
  const int64_t b = map / ((int64_t)V * J);
  unsigned long long k = 0ull;
  static_assert(MVAL_ARGMAX_SLOTS % 64 == 0, "whole waves over a row");
#pragma unroll
  for (int i = 0; i < MVAL_ARGMAX_SLOTS / 64; i++) {
    const unsigned long long v = keys[((map / J) * MVAL_ARGMAX_SLOTS + i * 64 + lane) * J + j];  // [image][slot][joint]
    k = v > k ? v : k;
  }
  k = mval_key_group_max(k, 32);
  if (lane != 0) return;
  const bool inval = valid && !valid[b * J + j];
  const unsigned bi = k == 0ull ? 0u : 0xffffffffu - (unsigned)k;
  kp2d[map * 2] = inval ? 0 : (int64_t)(bi % (unsigned)split_width) * stride;
  kp2d[map * 2 + 1] = inval ? 0 : (int64_t)(bi / (unsigned)split_width) * stride;
}

extern "C" int mval_argmax_from_keys(const uint64_t* keys, const uint8_t* valid, int64_t* kp2d, int B, int V, int J, int stride,
                                     int split_width, void* stream) {
  MVAL_REQUIRE(keys && kp2d && B >= 0 && V > 0 && J > 0 && split_width > 0, "mval_argmax_from_keys: bad arguments");
  const int64_t n_maps = (int64_t)B * V * J;
  if (n_maps == 0) return 0;
  hipLaunchKernelGGL(argmax_from_keys_kernel, dim3((unsigned)((n_maps + 3) / 4)), dim3(256), 0, mval_stream(stream),
                     reinterpret_cast<const unsigned long long*>(keys), valid, kp2d, n_maps, V, J, stride, split_width);
  MVAL_CHECK_LAUNCH("mval_argmax_from_keys");
  return 0;
}

// ---- soft-argmax ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void soft_argmax_kernel(const float* __restrict__ hm, float* __restrict__ out,
                                                          int64_t n_maps, int hh, int wh, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t map = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (map >= n_maps) return;
  const int npix = hh * wh;
  const float* p = hm + map * (int64_t)npix;
  float m = -INFINITY;
  for (int i = lane; i < npix; i += 64) m = fmaxf(m, p[i]);
  m = wave_max(m);
  float s = 0.f, sx = 0.f, sy = 0.f;
  for (int i = lane; i < npix; i += 64) {
    float e = expf(p[i] - m);
    int y = i / wh, x = i - y * wh;
    s += e;
    sx += e * (float)x;
    sy += e * (float)y;
  }
  s = wave_sum(s);
  sx = wave_sum(sx);
  sy = wave_sum(sy);
  if (lane == 0) {
    out[map * 2] = (sx / s) * scale;
    out[map * 2 + 1] = (sy / s) * scale;
  }
}

extern "C" int mval_soft_argmax(const float* heatmaps, float* kp2d, int64_t n_maps, int hh, int wh, float scale,
                                void* stream) {
  MVAL_REQUIRE(n_maps >= 0 && hh > 0 && wh > 0, "mval_soft_argmax: bad dims");
  if (n_maps == 0) return 0;
  dim3 grid((unsigned)((n_maps + 3) / 4));
  hipLaunchKernelGGL(soft_argmax_kernel, grid, dim3(256), 0, mval_stream(stream), heatmaps, kp2d, n_maps, hh, wh,
                     scale);
  MVAL_CHECK_LAUNCH("mval_soft_argmax");
  return 0;
}
