// Shared definitions of the fused conv operators (direct and MFMA kernels).
#pragma once
#include "mval_common.h"

struct ConvArgs {
  const float* in;
  const float* w;      // packed weights
  const float* scale;  // [cout] folded BN scale (or 1)
  const float* shift;  // [cout] folded BN shift (or bias)
  const float* res1;   // optional residuals at output resolution (after upsample), NHWC
  const float* res2;
  float* out;
  int N, Hin, Win, Cin, Hout, Wout, Cout;  // Hout/Wout before the fused upsample
  int k, stride, pad;
  int up, relu, in_nchw, out_nchw;
  int dil;  // 1, or 2: the input is read as if zero-dilated by 2 (data gradient of a stride-2 conv)
  // Views (all 0 = plain conv; ConvArgs is zero-initialised by its creators):
  //   org_dy / org_dx  extra shift of the input window on top of -pad (asymmetric windows);
  //   os_log2, ooy, oox  the (Hout, Wout) grid is written to pixel (y << os_log2) + ooy, (x << os_log2) + oox
  //                      of an (Hout << os_log2) x (Wout << os_log2) tensor (parity planes of a transposed conv)
  //   in_sub_log2  the input is read at pixels (y << in_sub_log2, x << in_sub_log2): a stride-2 1x1 conv is
  //                a stride-1 1x1 conv on that subsampled view (nothing else is staged)
  int org_dy, org_dx, os_log2, ooy, oox, in_sub_log2;
  // split-bf16 2x2 parity kernels: != 0 = all four parities of a transposed conv in ONE launch (blockIdx.z = parity
  // sets org_dy/dx and ooy/oox and advances w by parity * par_w_stride floats)
  int par_w_stride;
  int precise;  // split-bf16 kernel: separate accumulator for the correction products (training plans)
  // 16-bit split kernels (conv_mfma_split.hip): planes = 3 bf16x3 (six products), 2 fp16x2 (three products).  The
  // fp16 split needs the power-of-two scales: in_amax[n][MVAL_AMAX_ROW] = the max |x| row of image n of the input
  // tensor (kept by its producer; PER IMAGE, so a frame's result does not depend on what else is in the batch;
  // conv_amax_read gives the value), w_unscale = the factor that undoes the weight scale (trailer of the packed
  // weights).
  int planes;
  int in_amax_stride;  // dwords between the rows of consecutive images (MVAL_AMAX_ROW); 0 = ONE row for the whole tensor
  const unsigned* in_amax;
  const float* w_unscale;
  // != nullptr: every workgroup that writes part of image n of `out` leaves max |value| of what it stored in its
  // slot of the row out_amax[n][MVAL_AMAX_ROW] (see conv_amax_put).  amax_tiles: workgroups per image and cout
  // group (filled by the launcher)
  unsigned* out_amax;
  int amax_tiles;
  // NCHW heat-map layer (1x1, stride 1, out_nchw, up == 0) on conv_mfma.hip's kernels (KEYS instantiations of
  // conv_tile_store): != nullptr = also keep the arg-max keys of every map, [N][MVAL_ARGMAX_SLOTS][Cout] (mval_common.h: decode from the epilogue)
  unsigned long long* argmax_keys;
  // Training forward (raw z out, no residual / ReLU / upsample; net_train.hip): != nullptr = every workgroup also leaves
  // the per-channel (sum, sum of squares) of the pixels it stored, float64 [cout][bn_tiles][2] -- train-mode BatchNorm's
  // batch statistics (hrnet.py:16 nn.BatchNorm2d under model.train()) without a second pass over z.  The launcher
  // fills bn_tiles (workgroups along x) and reports it in *bn_tiles_host; a kernel / tile shape that cannot keep them
  // leaves *bn_tiles_host untouched (the caller then runs the separate statistics pass).  bn_part_cap: doubles at bn_part
  double* bn_part;
  int64_t bn_part_cap;
  int* bn_tiles_host;
  int bn_tiles;
  // MFMA tiling (filled by the launcher)
  int th, tw, tn, tw_log2, thw_log2;
  int tiles_x, tiles_y;
  unsigned pw_magic, ph_magic;  // ceil(2^20 / PW), ceil(2^20 / PH): patch index -> (row, column) without a divide
  unsigned tw_magic;            // != 0: odd tile (tw x th pixels, neither a power of two, tn = 1): ceil(2^20 / tw)
  int G_total, NS_total;
};

// out = act(((v + res1) + res2)), written to the 2^up x 2^up replicated positions.
__device__ __forceinline__ void conv_store(const ConvArgs& a, int n, int y, int x, int c, float v) {
  const int Ho = a.Hout << a.up, Wo = a.Wout << a.up;
  const int rep = 1 << a.up;
  for (int dy = 0; dy < rep; dy++) {
    for (int dx = 0; dx < rep; dx++) {
      const int Y = (y << a.up) + dy, X = (x << a.up) + dx;
      const int64_t o = (((int64_t)n * Ho + Y) * Wo + X) * a.Cout + c;
      float r = v;
      if (a.res1) r += a.res1[o];
      if (a.res2) r += a.res2[o];
      if (a.relu) r = mval_relu(r);
      if (a.out_nchw)
        a.out[(((int64_t)n * a.Cout + c) * Ho + Y) * Wo + X] = r;
      else
        a.out[o] = r;
    }
  }
}

// Row of one (activation, image): [count, partial maxima ...] (MVAL_AMAX_ROW dwords, include/mval_hip.h).  Every
// WAVE that writes part of the image stores the maximum of what it wrote (bits of a non-negative float) in its own
// slot and the number of such waves in the header -- plain stores, fire and forget: no atomics (device-scope
// atomics on one cache line cost ~300 ns each: 20 us on a 60 us conv), no workgroup barrier at the end of the kernel
// (a reduction over the waves first: 12 us on the same conv, the workgroup keeps its LDS / registers while it
// waits), nothing to zero between forwards; the consumer takes the maximum of the first `count` slots.  More than
// MVAL_AMAX_ROW - 1 partials per image (inputs above ~512 x 512): the launcher zeroes the rows and the waves fold
// into slot % (MVAL_AMAX_ROW - 1) with atomicMax.
__device__ __forceinline__ void conv_amax_put(unsigned* row, int slot, int count, unsigned bits) {
  if (count <= MVAL_AMAX_ROW - 1) {
    row[0] = (unsigned)count;
    row[1 + slot] = bits;
  } else {
    row[0] = MVAL_AMAX_ROW - 1;
    atomicMax(row + 1 + slot % (MVAL_AMAX_ROW - 1), bits);
  }
}
// One image per workgroup: every wave leaves its own partial (workgroup slot `slot` of `count`).  All lanes of the
// wave must call it.
__device__ __forceinline__ void conv_amax_commit(unsigned* row, int slot, int count, float m) {
  m = wave_max(m);
  const int nw = (int)(blockDim.x >> 6);
  if ((threadIdx.x & 63) == 0) conv_amax_put(row, slot * nw + (int)(threadIdx.x >> 6), count * nw, __float_as_uint(m));
}
// Consumer side: max |x| bits of an image.  Lanes read the partial slots, one wave reduction (uniform result).
__device__ __forceinline__ unsigned conv_amax_read(const unsigned* row) {
  const int count = min((int)row[0], MVAL_AMAX_ROW - 1);  // (never past the row, whatever the header holds)
  unsigned m = 0;
  for (int i = (int)(threadIdx.x & 63); i < count; i += 64) m = max(m, row[1 + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  return m;
}
__device__ __forceinline__ float conv_amax4(float m, const float x, const float y, const float z, const float w) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(x), fabsf(y))), fmaxf(fabsf(z), fabsf(w)));
}

// floor(v / d) for v < 4096 and d < 256 with magic = ceil(2^20 / d) (exact: v * (d - 1) < 2^20)
__device__ __forceinline__ int conv_div20(int v, unsigned magic) { return (int)(((unsigned)v * magic) >> 20); }

// Tile-local pixel slot -> (image of the tile, row, column).  Power-of-two tiles decode with shifts; the odd tiles of
// the split kernel (a.tw_magic != 0: th x tw pixels, one image per tile, slots >= th * tw are padding -> false) with
// a magic divide.
__device__ __forceinline__ bool conv_tile_decode(const ConvArgs& a, int p, int& tni, int& ty, int& tx) {
  if (a.tw_magic) {
    ty = conv_div20(p, a.tw_magic);
    tx = p - ty * a.tw;
    tni = 0;
    return ty < a.th;
  }
  tni = p >> a.thw_log2;
  const int rem = p & ((1 << a.thw_log2) - 1);
  ty = rem >> a.tw_log2;
  tx = rem & ((1 << a.tw_log2) - 1);
  return true;
}

// Second half of the MFMA kernels' epilogue: the BN'd output tile sits in LDS as
// ot[pixel][NTILE + 4] (followed by MT + 8 dwords of scratch for the max |x| reduction: the launchers size the LDS
// for it, CONV_OTILE_FLOATS); add the residual(s), ReLU and store with float4 lanes along channels.
// All residual loads of a thread are issued before the first store (MT*NTILE/1024 float4 loads
// in flight per thread) -- with one load per iteration the memory-bound layers (1x1 convs,
// 32-channel 3x3) sat at 2.8 TB/s, latency- rather than bandwidth-bound.
typedef float conv_f32x4 __attribute__((ext_vector_type(4)));

#define CONV_OTILE_FLOATS(MT, NTILE) ((MT) * ((NTILE) + 4) + (MT) + 8)

// Leave the workgroup's max |stored value| in the rows of its image(s).  tn == 1: `amax` holds every thread's
// running maximum; tn > 1: the threads already folded theirs into scratch[image of the tile] (LDS atomics).
template <int MT>
__device__ __forceinline__ void conv_tile_amax(const ConvArgs& a, unsigned* scratch, float amax, int n0) {
  if (!a.out_amax) return;
  const int slot = (int)(blockIdx.x % (unsigned)a.amax_tiles) + a.amax_tiles * (int)(blockIdx.y + gridDim.y * blockIdx.z);
  const int count = a.amax_tiles * (int)(gridDim.y * gridDim.z);
  if (a.tn == 1) {
    conv_amax_commit(a.out_amax + (int64_t)n0 * MVAL_AMAX_ROW, slot, count, amax);
    return;
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < a.tn && n0 + t < a.N) conv_amax_put(a.out_amax + (int64_t)(n0 + t) * MVAL_AMAX_ROW, slot, count, scratch[t]);
}

template <int MT, int NTILE, int NTH = 256, bool KEYS = false>
__device__ __forceinline__ void conv_tile_store(const ConvArgs& a, const float* ot, int tid, int n0, int oy0, int ox0,
                                                int cbase) {
  constexpr int LDW = NTILE + 4;
  unsigned* scratch = reinterpret_cast<unsigned*>(const_cast<float*>(ot)) + MT * LDW;
  if (a.out_amax && a.tn != 1) {  // rare (maps under 8 rows): per-image maxima of the tile, zeroed here
    for (int i = tid; i < a.tn; i += NTH) scratch[i] = 0u;
    __syncthreads();
  }
  const int Ho = a.Hout << a.up, Wo = a.Wout << a.up, rep = 1 << a.up;
  if (!a.out_nchw && (a.Cout & 3) == 0) {
    constexpr int Q = NTILE / 4;
    constexpr int IT = (MT * Q + NTH - 1) / NTH;
    if (a.up == 0) {
      int64_t off[IT];
      conv_f32x4 r1[IT], r2[IT];
#pragma unroll
      for (int i = 0; i < IT; i++) {
        const int e = tid + NTH * i;
        const int p = e / Q, c4 = e % Q;
        const int c = cbase + c4 * 4;
        int tni, ty, tx;
        const bool inside = conv_tile_decode(a, p, tni, ty, tx);
        const int y = oy0 + ty, x = ox0 + tx;
        const int n = n0 + tni;
        const bool ok = inside && e < MT * Q && c < a.Cout && n < a.N && y < a.Hout && x < a.Wout;
        off[i] = ok ? (((int64_t)n * (a.Hout << a.os_log2) + (y << a.os_log2) + a.ooy) * (a.Wout << a.os_log2) +
                       (x << a.os_log2) + a.oox) * a.Cout + c
                    : -1;
        r1[i] = (ok && a.res1) ? *reinterpret_cast<const conv_f32x4*>(a.res1 + off[i]) : (conv_f32x4){0.f, 0.f, 0.f, 0.f};
        r2[i] = (ok && a.res2) ? *reinterpret_cast<const conv_f32x4*>(a.res2 + off[i]) : (conv_f32x4){0.f, 0.f, 0.f, 0.f};
      }
      float amax = 0.f;
      conv_f32x4 bsum = (conv_f32x4){0.f, 0.f, 0.f, 0.f}, bsq = (conv_f32x4){0.f, 0.f, 0.f, 0.f};  // (a.bn_part)
#pragma unroll
      for (int i = 0; i < IT; i++) {
        if (off[i] < 0) continue;
        const int e = tid + NTH * i;
        conv_f32x4 r = *reinterpret_cast<const conv_f32x4*>(ot + (e / Q) * LDW + (e % Q) * 4);
        if constexpr (NTH % Q == 0) {  // a thread's granules are all of ONE channel quad (e % Q == tid % Q)
          if (a.bn_part) {
            bsum += r;
            bsq += r * r;
          }
        }
        if (a.res1) r += r1[i];
        if (a.res2) r += r2[i];
        if (a.relu) {
          r.x = mval_relu(r.x); r.y = mval_relu(r.y); r.z = mval_relu(r.z); r.w = mval_relu(r.w);
        }
        *reinterpret_cast<conv_f32x4*>(a.out + off[i]) = r;
        amax = conv_amax4(amax, r.x, r.y, r.z, r.w);
        if (a.out_amax && a.tn != 1) {  // several images per tile (maps under 8 rows): per image through LDS
          atomicMax(scratch + ((e / Q) >> a.thw_log2), __float_as_uint(amax));
          amax = 0.f;
        }
      }
      if constexpr (NTH % Q == 0 && MT * LDW >= NTH * 8) {
        if (a.bn_part) {
          // batch-statistics partials of the tile: thread sums (a few pixels of one channel quad, float32) -> LDS ->
          // one thread per channel adds the NTH / Q partials of its channel in float64, fixed order (deterministic)
          __syncthreads();  // every thread is done with the tile in `ot`
          float* sc = const_cast<float*>(ot);
          *reinterpret_cast<conv_f32x4*>(sc + tid * 8) = bsum;
          *reinterpret_cast<conv_f32x4*>(sc + tid * 8 + 4) = bsq;
          __syncthreads();
          if (tid < NTILE && cbase + tid < a.Cout) {
            const int c4 = tid >> 2, k = tid & 3;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
            for (int j = 0; j < NTH / Q; j++) {
              s1 += (double)sc[(j * Q + c4) * 8 + k];
              s2 += (double)sc[(j * Q + c4) * 8 + 4 + k];
            }
            double* dst = a.bn_part + ((int64_t)(cbase + tid) * a.bn_tiles + blockIdx.x) * 2;
            dst[0] = s1;
            dst[1] = s2;
          }
        }
      }
      conv_tile_amax<MT>(a, scratch, amax, n0);
      return;
    }
    float amax = 0.f;
    for (int e = tid; e < MT * Q; e += NTH) {  // fused nearest upsample: 2^up x 2^up replicas
      const int p = e / Q, c4 = e % Q;
      const int c = cbase + c4 * 4;
      if (c >= a.Cout) continue;
      int tni, ty, tx;
      if (!conv_tile_decode(a, p, tni, ty, tx)) continue;
      const int y = oy0 + ty, x = ox0 + tx;
      const int n = n0 + tni;
      if (n >= a.N || y >= a.Hout || x >= a.Wout) continue;
      const conv_f32x4 v = *reinterpret_cast<const conv_f32x4*>(ot + p * LDW + c4 * 4);
      // four replicas at a time (rep * rep is 4, 16 or 64), their residual loads issued before the first store
      for (int i0 = 0; i0 < rep * rep; i0 += 4) {
        int64_t o[4];
        conv_f32x4 r1[4], r2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int dy = (i0 + u) >> a.up, dx = (i0 + u) & (rep - 1);
          o[u] = (((int64_t)n * Ho + (y << a.up) + dy) * Wo + (x << a.up) + dx) * a.Cout + c;
          r1[u] = a.res1 ? *reinterpret_cast<const conv_f32x4*>(a.res1 + o[u]) : (conv_f32x4){0.f, 0.f, 0.f, 0.f};
          r2[u] = a.res2 ? *reinterpret_cast<const conv_f32x4*>(a.res2 + o[u]) : (conv_f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          conv_f32x4 r = v + r1[u] + r2[u];
          if (a.relu) {
            r.x = mval_relu(r.x); r.y = mval_relu(r.y); r.z = mval_relu(r.z); r.w = mval_relu(r.w);
          }
          *reinterpret_cast<conv_f32x4*>(a.out + o[u]) = r;
          amax = conv_amax4(amax, r.x, r.y, r.z, r.w);
        }
      }
      if (a.out_amax && a.tn != 1) {
        atomicMax(scratch + tni, __float_as_uint(amax));
        amax = 0.f;
      }
    }
    conv_tile_amax<MT>(a, scratch, amax, n0);
    return;
  }
  // scalar path (NCHW heat-map output, odd channel counts): pixel-fastest so that NCHW rows are
  // written in contiguous runs
  if constexpr (KEYS) if (a.argmax_keys && a.out_nchw && a.up == 0) {
    // the heat-map layer with decode from the epilogue (hrnet.py:344-350,500 -> utils/evaluation.py:13-30): every wave
    // folds the keys of what it stores -- a wave's 64 elements are 64 pixels of ONE map when MT is a multiple of 64 and the
    // tile holds one image; otherwise per lane
    static_assert(NTH % 64 == 0, "whole waves");
    for (int e0 = 0; e0 < MT * NTILE; e0 += NTH) {  // (uniform trip count: the shuffles below need every lane)
      const int e = e0 + tid;
      const int cl = e / MT, p = e % MT;
      const int c = cbase + cl;
      int tni = 0, ty = 0, tx = 0;
      const bool inside = conv_tile_decode(a, p, tni, ty, tx);
      const int y = oy0 + ty, x = ox0 + tx;
      const int n = n0 + tni;
      const bool ok = e < MT * NTILE && inside && c < a.Cout && n < a.N && y < a.Hout && x < a.Wout;
      unsigned long long key = 0ull;
      const int64_t map = (int64_t)n * a.Cout + c;
      if (ok) {
        float r = ot[p * LDW + cl];
        const int64_t o = (((int64_t)n * a.Hout + y) * a.Wout + x) * a.Cout + c;
        if (a.res1) r += a.res1[o];
        if (a.res2) r += a.res2[o];
        if (a.relu) r = mval_relu(r);
        a.out[(map * a.Hout + y) * a.Wout + x] = r;
        key = mval_argmax_key(r, (unsigned)(y * a.Wout + x));
      }
      if ((MT & 63) == 0 && a.tn == 1) {  // the wave's lanes share (n, c): its key goes to its slot of the map's row
        key = mval_key_row16_max(key);
        {  // rows 0..3 -> lane 0 (two xor steps across the rows)
          const unsigned long long k16 = ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(key >> 32), 16, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)key, 16, 64);
          key = k16 > key ? k16 : key;
          const unsigned long long k32 = ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(key >> 32), 32, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)key, 32, 64);
          key = k32 > key ? k32 : key;
        }
        if ((tid & 63) == 0 && c < a.Cout && n < a.N)
          mval_argmax_key_put(a.argmax_keys, n, c, a.Cout, (int)(blockIdx.x % (unsigned)a.amax_tiles) * (MT / 64) + p / 64,
                              a.amax_tiles * (MT / 64), key);
      } else if (key) {  // (several images per tile -- maps under 8 rows -- or 16-pixel groups: a few atomics per map)
        atomicMax(a.argmax_keys + (int64_t)n * MVAL_ARGMAX_SLOTS * a.Cout + c, key);
      }
    }
    return;
  }
  for (int e = tid; e < MT * NTILE; e += NTH) {
    const int cl = e / MT, p = e % MT;
    const int c = cbase + cl;
    if (c >= a.Cout) continue;
    int tni, ty, tx;
    if (!conv_tile_decode(a, p, tni, ty, tx)) continue;
    const int y = oy0 + ty, x = ox0 + tx;
    const int n = n0 + tni;
    if (n >= a.N || y >= a.Hout || x >= a.Wout) continue;
    conv_store(a, n, y, x, c, ot[p * LDW + cl]);
  }
}

// Launcher side of the max |x| rows: sets a.amax_tiles; more workgroups per image than a row has slots -> the rows are
// zeroed here and the kernel folds with atomics (conv_amax_put).
void mval_launch_zero_rows(unsigned* rows, int64_t n_dwords, hipStream_t s);  // net.hip (a kernel: graph-capturable anywhere)
static inline void conv_amax_prepare(ConvArgs& a, int tiles_per_image, int groups, hipStream_t s) {
  a.amax_tiles = tiles_per_image;
  if (a.out_amax && (int64_t)tiles_per_image * groups * 4 > MVAL_AMAX_ROW - 1)  // (up to 4 waves per workgroup)
    mval_launch_zero_rows(a.out_amax, (int64_t)a.N * MVAL_AMAX_ROW, s);
}

// Launcher side of the batch-statistics partials (a.bn_part): keeps them when the tile shape can (conv_tile_store's
// float4 path with NTH % (NTILE / 4) == 0 and room for the NTH x 8 float scratch in the output tile) and the buffer
// holds cout x tiles x 2 doubles; otherwise switches them off for this launch.
template <int MT, int NTILE, int NTH>
static inline void conv_bn_part_prepare(ConvArgs& a, unsigned grid_x, unsigned grid_z) {
  if (!a.bn_part) return;
  constexpr int Q = NTILE / 4;
  const bool ok = NTH % Q == 0 && MT * (NTILE + 4) >= NTH * 8 && grid_z == 1 && a.up == 0 && !a.out_nchw && (a.Cout & 3) == 0 &&
                  !a.res1 && !a.res2 && !a.relu && (int64_t)a.Cout * grid_x * 2 <= a.bn_part_cap;
  if (!ok) {
    a.bn_part = nullptr;
    return;
  }
  a.bn_tiles = (int)grid_x;
  if (a.bn_tiles_host) *a.bn_tiles_host = (int)grid_x;
}

int mval_launch_conv_mfma(const ConvArgs& a, hipStream_t s);  // conv_mfma.hip; returns 1 if unsupported
int mval_conv_mfma_supported(const ConvArgs& a);            // same selection logic, no launch
int mval_launch_conv_split(const ConvArgs& a, hipStream_t s);  // conv_mfma_split.hip (a.planes); returns 1 if unsupported
int mval_conv_split_supported(const ConvArgs& a);
int mval_pack_bf3(int mode, const float* w, float* packed, int cout, int cin, int k, hipStream_t s);
int mval_pack_h2(int mode, const float* w, float* packed, int cout, int cin, int k, hipStream_t s);
// net.hip: rows[i][*] = max(rows[i][*], max |x| over image i) for n_images images of per_image floats each
// ---- branch concurrency (net.hip, net_train.hip) ---------------------------------------------------------------
// Ops carry (phase, lane) hints: ops of one phase on different lanes are independent (HRNet branches / fuse outputs).
// Lanes > 0 run on per-device side streams that fork from / join into the caller's stream with events at every phase
// change (no host synchronisation; capturable).  The streams and events are created once per device and never
// destroyed (stream churn around hipGraph captures crashed a later replay inside the HIP runtime).
#define MVAL_MAX_LANES 4
struct MvalLanes {
  hipStream_t side[MVAL_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr;
  hipEvent_t join_ev[MVAL_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  bool ready = false;
};
MvalLanes* mval_device_lanes();  // net.hip; nullptr on failure

// One pass over an op list: stream_for() per op in list order, finish() at the end.
struct MvalLaneWalk {
  MvalLanes* L;
  hipStream_t main_s;
  bool used[MVAL_MAX_LANES] = {false, false, false, false};
  bool forked = false, started = false;
  int phase = 0;
  MvalLaneWalk(MvalLanes* lanes, hipStream_t s) : L(lanes), main_s(s) {}
  void join() {  // side streams -> main
    for (int l = 1; l < MVAL_MAX_LANES; l++)
      if (used[l]) {
        (void)hipEventRecord(L->join_ev[l], L->side[l]);
        (void)hipStreamWaitEvent(main_s, L->join_ev[l], 0);
        used[l] = false;
      }
  }
  hipStream_t stream_for(int op_phase, int op_lane) {
    if (!L) return main_s;
    if (!started || op_phase != phase) {
      if (started) join();
      phase = op_phase;
      started = true;
      forked = false;
    }
    if (!forked) {
      // recorded at the START of the phase, before its lane-0 ops are enqueued: the side lanes then wait for the
      // previous phases only, not for this phase's (longest) lane-0 chain as well
      (void)hipEventRecord(L->fork_ev, main_s);
      forked = true;
    }
    const int lane = (op_lane > 0 && op_lane < MVAL_MAX_LANES) ? op_lane : 0;
    if (lane == 0) return main_s;
    if (!used[lane]) {
      (void)hipStreamWaitEvent(L->side[lane], L->fork_ev, 0);
      used[lane] = true;
    }
    return L->side[lane];
  }
  void finish() {
    if (L) join();
  }
  ~MvalLaneWalk() { finish(); }  // (an error return in the middle of a pass still joins the side streams)
};

// conv_block.hip: a whole BasicBlock (two 3x3 convs + BNs + residual + ReLUs) in one launch; returns 1 if unsupported
int mval_conv_block_supported(int C, int N, int H, int W);
int mval_launch_conv_block(int C, const float* in, float* out, const float* w1, const float* scale1, const float* shift1,
                           const float* w1_unscale, const float* w2, const float* scale2, const float* shift2,
                           const float* w2_unscale, const unsigned* in_amax, unsigned* out_amax, int N, int H, int W,
                           hipStream_t s);
int mval_launch_amax(const float* x, int64_t per_image, int n_images, unsigned* rows, hipStream_t s);
int mval_pack_bf3_batch(const void* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, hipStream_t s);
int mval_pack_split_batch(const void* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, hipStream_t s);
int mval_launch_conv_direct(const ConvArgs& a, int kind, hipStream_t s);
int mval_launch_conv_stem(const ConvArgs& a, hipStream_t s);  // conv_stem.hip; returns 1 if unsupported
