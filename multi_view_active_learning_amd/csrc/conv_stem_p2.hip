// Fused HRNet stem over a P2 output (conv_p2.h), hrnet.py:303-310 / 469-474 in eval mode:
//
//     y1  = relu(bn1(conv3x3 s2 (x)))      x:  fp32 NCHW image, 3 channels            (H x W)
//     out = relu(bn2(conv3x3 s2 (y1)))     y1: 64 channels (H/2 x W/2), never leaves the CU;  out: 64 channels, P2 planes (H/4 x W/4)
//
// Launched op by op the pair costs 263 us (VALU stem, 537 MB written) + 287 us (fp32 -> P2 format change) + ~215 us
// (the 64 -> 64 stride-2 conv reading those 537 MB again) per 128 images; fused, the only HBM traffic is the image
// (100 MB) and the output planes (134 MB).
//
// Workgroup = 4 waves on a 2 x 16 tile of `out`, persistent over an XCD-contiguous range of tiles; 51 KB of LDS, three
// workgroups per CU:
//   P   the 11 x 67 input patch (zero outside the image) as the (h, l) fp16 planes of x * 2^sx, sx from the image's max |x|;
//       the next tile's travels in registers;
//   1.  conv1 on the matrix cores too: K = 27 taps padded to ONE 32-deep MFMA step.  The B fragment of sixteen intermediate
//       pixels is an im2col gather (a lane reads its eight taps' halves out of the patch), the A fragments (4 sub-tiles x
//       (h, l) of the scaled weights) are built once per workgroup from the [27][64] table.  11 fragments cover the 5 x 33
//       intermediate; a wave takes every fourth.  BN1 + ReLU + zero outside y1 -> scaled by the per-IMAGE bound A1 max|x| +
//       B1, split, 16-byte granules into Y1 = [chunk][plane h,l][8-ch block][column parity, row, column / 2][16 B].
//       (The first form did conv1 on the vector ALUs in exact fp32 -- 594 packed FMAs per lane and tile, 41 % of the
//       kernel: 388 us against 297 us now; -DST_VALU_CONV1 still builds it.)
//   2.  conv2 on the matrix cores from Y1 (conv_p2.hip arithmetic: three fp16 MFMA products per fp32 product): wave =
//       16 output channels x 2 rows, a row fragment = 16 consecutive slots of one column parity;
//       BN2 + ReLU + max |x| + split in registers, 16-byte stores into the output planes.
// max |x| of every image comes from a small pass over the input first (image_amax_rows_kernel, P2 rows).
#include <stdlib.h>

#include "conv_p2.h"
#ifndef P2_STEM_AUX
#define P2_STEM_AUX 2  // cache policy of the image loads: non-temporal (every image is read by this launch only); with P2_RES_AUX: C2 10.08 -> 10.01 ms
#endif

#ifndef P2_VALU_PRIO
#define P2_VALU_PRIO 2
#endif
#ifndef ST_VALU_CONV1  // -DST_VALU_CONV1: conv1 on the vector ALUs in exact fp32 (the first form of this kernel: 388 vs 297 us)
#define ST_MFMA1 1
#endif

typedef p2_f32x4 f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef p2_f16x8 f16x8;
typedef p2_f16x4 f16x4;
typedef p2_u32x4 u32x4;
typedef p2_u32x2 u32x2;

#ifdef P2_STAMP
#define ST_T0 unsigned long long bp_t = wall_clock64(), bp_t00 = bp_t; unsigned long long bp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define ST_ACC(k)                                 \
  do {                                            \
    const unsigned long long t_ = wall_clock64(); \
    bp_acc[k] += t_ - bp_t;                       \
    bp_t = t_;                                    \
  } while (0)
#define ST_FLUSH                                                                                     \
  do {                                                                                               \
    if (g_dbg && lane == 0) {                                                                        \
      unsigned long long* d_ = g_dbg + ((int64_t)blockIdx.x * 4 + wave) * 16;                        \
      d_[0] = bp_t00; d_[4] = wall_clock64(); d_[1] = d_[0];                                         \
      for (int k_ = 0; k_ < 8; k_++) d_[8 + k_] = bp_acc[k_];                                        \
    }                                                                                                \
  } while (0)
extern unsigned long long* g_p2_dbg_shared;
#else
#define ST_T0
#define ST_ACC(k)
#define ST_FLUSH
#endif

struct StemP2Args {
  const float* in;   // [N][3][H][W]
  _Float16* out;     // P2 planes [N][2][8][H2][W2][8]
  const float* w1;   // [27][64] (tap-major, cin, cout: MVAL_PACK_HWIO)
  const float *scale1, *shift1, *bound1;
  const float *w2, *w2_unscale, *scale2, *shift2, *bound2;  // w2: MVAL_PACK_MFMA16_H2
  const unsigned* in_row;  // P2 rows of the input image (its max |x|)
  unsigned* out_row;
  int N, H, W, H1, W1, H2, W2;
  int tiles_x, tiles_y, tiles_total, wgs_x;
  unsigned tiles_img_magic, tiles_x_magic;
  unsigned long long* dbg;  // diagnostic builds (-DP2_STAMP)
};

__device__ __forceinline__ f32x4 st_mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int st_fresh(int v) {  // (conv_bneck_p2.hip: keeps per-phase address plans out of the tile loop's registers)
  asm volatile("" : "+v"(v));
  return v;
}

// max |x| of every image -> slots 0 .. gridDim.x - 1 of its P2 row (the other slots stay zero, the scale slot is unused)
__global__ __launch_bounds__(256) void image_amax_rows_kernel(const float* __restrict__ x, int64_t per_image, unsigned* __restrict__ rows) {
  const int n = blockIdx.y;
  const f32x4* p = reinterpret_cast<const f32x4*>(x + (int64_t)n * per_image);
  const int64_t n4 = per_image >> 2;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = p[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < per_image; i += 256) m = fmaxf(m, fabsf(x[(int64_t)n * per_image + i]));
  __shared__ unsigned red;
  if (threadIdx.x == 0) red = 0u;
  __syncthreads();
  const unsigned b = p2_wave_umax(__float_as_uint(m));
  if ((threadIdx.x & 63) == 0) atomicMax(&red, b);
  __syncthreads();
  if (threadIdx.x == 0) rows[(int64_t)n * P2_ROW + blockIdx.x] = red;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void conv_stem_p2_kernel(StemP2Args a) {
  constexpr int TH = 2, TW = 16;                       // tile of `out`
  constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1;      // 5 x 33 pixels of y1
  constexpr int PWh = (PW + 1) / 2, SL = 2 * PH * PWh;  // 170 slots per 8-channel block: [column parity][row][column / 2]
  constexpr int YPL = 4 * SL * 16, YCH = 8 * SL * 16, YB = 2 * YCH;  // 43 520 bytes
  constexpr int IH = 2 * PH + 1, IW = 2 * PW + 1, IWP = 68;          // 11 x 67 input pixels, row stride 68 floats
  constexpr int PB = 3 * IH * IWP * 4;                               // 8 976 bytes
  constexpr int P0 = YB, W0 = P0 + PB;                               // patch, workgroup reduction words
  constexpr int NP = (3 * IH * IW + 255) / 256;                      // patch elements per thread: 9
  constexpr int RUN = 11;                                            // y1 pixels per lane: a third of a row
  static_assert(PW == 3 * RUN, "three runs per intermediate row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* wgred = reinterpret_cast<unsigned*>(smem + W0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- tile walk (as conv_p2.hip) ------------------------------------------------------------------------------------
  const int X = a.wgs_x >= 8 ? 8 : 1;
  const int per = (a.tiles_total + X - 1) / X, wgx = a.wgs_x / X;
  const int xg = (int)blockIdx.x % X;
  int tile = xg * per + (int)blockIdx.x / X;
  const int tile_end = min(a.tiles_total, (xg + 1) * per);
  if (tile >= tile_end) return;
  const int tiles_img = a.tiles_x * a.tiles_y;
  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    n = a.tiles_img_magic ? (int)__umulhi((unsigned)t, a.tiles_img_magic) : t;
    const int r = t - n * tiles_img;
    const int tyi = a.tiles_x_magic ? (int)__umulhi((unsigned)r, a.tiles_x_magic) : r;
    oy0 = tyi * TH;
    ox0 = (r - tyi * a.tiles_x) * TW;
  };

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (unsigned)min((int64_t)0x7fffffff, (int64_t)a.N * 3 * a.H * a.W * 4), 0x00020000);
  const unsigned hw16 = (unsigned)(a.H2 * a.W2) * 16u, out_img = 2u * 8u * hw16, out_plane = 8u * hw16;
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.N * out_img, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w2), 0, 0x7fffffff, 0x00020000);

  // ---- the input patch of a tile: element e = tid + 256 i -> (plane c, patch row, patch column) --------------------------
  float pre[NP];
  auto load_patch = [&](int n, int oy0, int ox0) {
    const int T = st_fresh(tid);
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int e = T + 256 * i;
      const int c = e / (IH * IW), r = e - c * (IH * IW);
      const int py = r / IW, px = r - py * IW;
      const int iy = 4 * oy0 - 3 + py, ix = 4 * ox0 - 3 + px;
      const bool ok = e < 3 * IH * IW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? (unsigned)(((n * 3 + c) * a.H + iy) * a.W + ix) * 4u : 0x80000000u, 0, P2_STEM_AUX));
    }
  };
#ifdef ST_MFMA1
  // conv1 on the matrix cores: the patch is kept as the (h, l) fp16 planes of x * 2^sx (sx from the image's max |x|), 2 bytes each
  constexpr int PLB = 3 * IH * IWP * 2;
  auto store_patch = [&](float xmul) {
    const int T = st_fresh(tid);
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int e = T + 256 * i;
      const int c = e / (IH * IW), r = e - c * (IH * IW);
      const int py = r / IW, px = r - py * IW;
      if (e < 3 * IH * IW) {
        const float xs = pre[i] * xmul;
        const _Float16 h = (_Float16)xs, l = (_Float16)(xs - (float)h);
        *reinterpret_cast<_Float16*>(smem + P0 + ((c * IH + py) * IWP + px) * 2) = h;
        *reinterpret_cast<_Float16*>(smem + P0 + PLB + ((c * IH + py) * IWP + px) * 2) = l;
      }
    }
  };
#else
  auto store_patch = [&](float) {
    const int T = st_fresh(tid);
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int e = T + 256 * i;
      const int c = e / (IH * IW), r = e - c * (IH * IW);
      const int py = r / IW, px = r - py * IW;
      if (e < 3 * IH * IW) *reinterpret_cast<float*>(smem + P0 + ((c * IH + py) * IWP + px) * 4) = pre[i];
    }
  };
#endif

  // ---- prologue ----------------------------------------------------------------------------------------------------------
  int tn, toy, tox;
  decode(tile, tn, toy, tox);
  load_patch(tn, toy, tox);
  if (tid == 0) wgred[0] = wgred[1] = 0u;
  const float b1a = a.bound1[0], b1b = a.bound1[1], b2a = a.bound2[0], b2b = a.bound2[1], w2u = *a.w2_unscale;
  // BN vectors of the lane's channels (conv1: cout quad tid & 15; conv2: the wave's sub-tile), the first image's row
  const f32x4 sc1 = *reinterpret_cast<const f32x4*>(a.scale1 + (tid & 15) * 4), sh1 = *reinterpret_cast<const f32x4*>(a.shift1 + (tid & 15) * 4);
  const int c0_2 = wave * 16 + ((lane >> 4) & 1) * 8 + (lane >> 5) * 4;
  const f32x4 sc2 = *reinterpret_cast<const f32x4*>(a.scale2 + c0_2), sh2v = *reinterpret_cast<const f32x4*>(a.shift2 + c0_2);
  P2RowRegs row_in;
  p2_row_request(a.in_row, tn, row_in);
  float x_amax_cur = p2_row_amax(row_in), x_mul_cur, x_inv_cur;
  p2_scale_of(x_amax_cur, x_mul_cur, x_inv_cur);
#ifdef ST_MFMA1
  // conv1's weights as A fragments [cout sub-tile][plane]: K = 27 taps (k = tap * 3 + c, the [27][64] table's rows) padded to one
  // 32-deep step; rows 4..7 <-> 8..11 permuted as in conv_p2.hip; scaled by a power of two from max |w1|, split (h, l)
  float* bn1 = reinterpret_cast<float*>(smem + W0 + 16);  // scale1[64], shift1[64]
  if (tid < 64) { bn1[tid] = a.scale1[tid]; bn1[64 + tid] = a.shift1[tid]; }
  if (tid == 0) wgred[2] = 0u;
  __syncthreads();
  {
    float wm = 0.f;
    for (int i = tid; i < 27 * 64; i += 256) wm = fmaxf(wm, fabsf(a.w1[i]));
    const unsigned wb = p2_wave_umax(__float_as_uint(wm));
    if (lane == 0) atomicMax(&wgred[2], wb);
  }
  __syncthreads();
  float w1_mul, w1_inv;
  p2_scale_of(__uint_as_float(wgred[2]), w1_mul, w1_inv);
  u32x4 WA[4][2];
  int toff[8];  // byte offset (inside a plane of the patch) of the lane's eight taps: k octet lane >> 4
  {
    const int wr_ = lane & 15, prow = (wr_ & 3) | ((wr_ & 4) << 1) | ((wr_ & 8) >> 1), g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int k = g * 8 + i, tap = k / 3, c = k - tap * 3, ky = tap / 3, kx = tap - ky * 3;
      toff[i] = k < 27 ? ((c * IH + ky) * IWP + kx) * 2 : 0;  // (k >= 27: any finite value, its weight is zero)
    }
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      _Float16 h[8], l[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int k = g * 8 + i;
        const float w = k < 27 ? a.w1[k * 64 + ct * 16 + prow] * w1_mul : 0.f;
        h[i] = (_Float16)w;
        l[i] = (_Float16)(w - (float)h[i]);
      }
      WA[ct][0] = __builtin_bit_cast(u32x4, *reinterpret_cast<f16x8*>(h));
      WA[ct][1] = __builtin_bit_cast(u32x4, *reinterpret_cast<f16x8*>(l));
    }
  }
#endif
  store_patch(x_mul_cur);
  __syncthreads();
#ifdef P2_STAMP
  unsigned long long* g_dbg = a.dbg;
#endif
  ST_T0;

  for (;;) {
    const int n = tn, oy0 = toy, ox0 = tox;
    const int next_tile = tile + wgx;
    const bool have_next = next_tile < tile_end;
    if (have_next) decode(next_tile, tn, toy, tox);
    const float x_amax = x_amax_cur, x_inv = x_inv_cur;
    (void)x_inv;
    if (have_next) p2_row_request(a.in_row, tn, row_in);  // (the next tile's image: requested a tile ahead)
    const float y1_bound = b1a * x_amax + b1b;
    float m1_mul, m1_inv, out_mul, out_inv;
    p2_scale_of(y1_bound, m1_mul, m1_inv);
    p2_scale_of(b2a * y1_bound + b2b, out_mul, out_inv);
    if (oy0 == 0 && ox0 == 0 && tid == 0) a.out_row[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);

#ifdef ST_MFMA1
    // ---- 1. conv1 on the matrix cores: one 32-deep step per sixteen-pixel fragment of the 5 x 33 intermediate (11 fragments, a
    // wave takes every fourth) x four 16-channel sub-tiles; the B fragment is an im2col gather of the lane's eight taps ------------
    {
      const int L = st_fresh(lane);
      const float k1 = x_inv * w1_inv * m1_mul;  // (powers of two: exact)
      const int cq1 = ((L >> 4) & 1) * 8 + (L >> 5) * 4;
#pragma unroll
      for (int fi = 0; fi < 3; fi++) {
        const int f = wave + 4 * fi;
        if (f >= 11) break;
        const int praw = f * 16 + (L & 15), p = praw < PH * PW ? praw : PH * PW - 1;
        const int py = p / PW, px = p - py * PW;
        const char* b0 = smem + P0 + ((2 * py) * IWP + 2 * px) * 2;
        u32x4 xh, xl;
#pragma unroll
        for (int i2 = 0; i2 < 4; i2++) {
          const unsigned h0 = *reinterpret_cast<const unsigned short*>(b0 + toff[2 * i2]), h1 = *reinterpret_cast<const unsigned short*>(b0 + toff[2 * i2 + 1]);
          const unsigned l0 = *reinterpret_cast<const unsigned short*>(b0 + PLB + toff[2 * i2]), l1 = *reinterpret_cast<const unsigned short*>(b0 + PLB + toff[2 * i2 + 1]);
          xh[i2] = h0 | (h1 << 16);
          xl[i2] = l0 | (l1 << 16);
        }
        const int yy = 2 * oy0 - 1 + py, xx = 2 * ox0 - 1 + px;
        const bool inside = praw < PH * PW && yy >= 0 && yy < a.H1 && xx >= 0 && xx < a.W1;
        const int slot = ((px & 1) * PH + py) * PWh + (px >> 1);
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = st_mfma(WA[ct][1], xh, acc);
          acc = st_mfma(WA[ct][0], xl, acc);
          acc = st_mfma(WA[ct][0], xh, acc);
          const int c0 = ct * 16 + cq1;
          const f32x4 s1 = *reinterpret_cast<const f32x4*>(bn1 + c0) * k1, h1 = *reinterpret_cast<const f32x4*>(bn1 + 64 + c0) * m1_mul;
          f32x4 v = acc * s1 + h1;
          v.x = p2_max_nan(v.x, 0.f); v.y = p2_max_nan(v.y, 0.f); v.z = p2_max_nan(v.z, 0.f); v.w = p2_max_nan(v.w, 0.f);
          if (!inside) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          f16x4 h, l;
          p2_split(v, h, l);
          const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
          const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
          const auto s1_ = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
          if (praw < PH * PW)
            *reinterpret_cast<u32x4*>(smem + (ct >> 1) * YCH + (L >> 5) * YPL + ((ct & 1) * 2 + ((L >> 4) & 1)) * SL * 16 + slot * 16) =
                (u32x4){s0[0], s1_[0], s0[1], s1_[1]};
        }
      }
    }
#else
    // ---- 1. conv1 on the vector ALUs: lane = (cout quad q, run g of 11 pixels of intermediate row g / 3) -----------------------
    __builtin_amdgcn_s_setprio(P2_VALU_PRIO);
    {
      const int T = st_fresh(tid);
      const int q = T & 15, g = T >> 4;
      if (g < 3 * PH) {
        const int py = g / 3, px0 = (g - py * 3) * RUN;
        f32x4 acc[RUN];
#pragma unroll
        for (int p = 0; p < RUN; p++) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef ST_SKIP1
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            // 23 consecutive input pixels feed the row's 11 outputs at all three kx
            float xv[2 * RUN + 2];
            const char* row = smem + P0 + ((c * IH + 2 * py + ky) * IWP + 2 * px0) * 4;
#pragma unroll
            for (int j = 0; j < RUN + 1; j++) {
              const f32x2 v = *reinterpret_cast<const f32x2*>(row + j * 8);
              xv[2 * j] = v.x;
              xv[2 * j + 1] = v.y;
            }
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
              const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.w1 + ((ky * 3 + kx) * 3 + c) * 64 + q * 4);  // (6.9 KB shared by every lane: L1 hits)
#pragma unroll
              for (int p = 0; p < RUN; p++) {
                const f32x2 vv = {xv[2 * p + kx], xv[2 * p + kx]};
                const f32x2 lo = __builtin_elementwise_fma(vv, (f32x2){w4.x, w4.y}, (f32x2){acc[p].x, acc[p].y});
                const f32x2 hi = __builtin_elementwise_fma(vv, (f32x2){w4.z, w4.w}, (f32x2){acc[p].z, acc[p].w});
                acc[p] = (f32x4){lo.x, lo.y, hi.x, hi.y};
              }
            }
          }
        }
#endif
        ST_ACC(0);
        // BN1 + ReLU + zero outside y1 -> scaled, split, half-granules (4 channels x 2 bytes) of both planes
        const f32x4 s1 = sc1 * m1_mul, h1 = sh1 * m1_mul;
        const int yy = 2 * oy0 - 1 + py;
        const bool row_in_img = yy >= 0 && yy < a.H1;
        const int gb = (q >> 3) * YCH + ((q >> 1) & 3) * SL * 16 + (q & 1) * 8;
#pragma unroll
        for (int p = 0; p < RUN; p++) {
          const int px = px0 + p, xx = 2 * ox0 - 1 + px;
          f32x4 v = acc[p] * s1 + h1;
          v.x = p2_max_nan(v.x, 0.f); v.y = p2_max_nan(v.y, 0.f); v.z = p2_max_nan(v.z, 0.f); v.w = p2_max_nan(v.w, 0.f);
          if (!(row_in_img && xx >= 0 && xx < a.W1)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          f16x4 h, l;
          p2_split(v, h, l);
          const int slot = ((px & 1) * PH + py) * PWh + (px >> 1);
          *reinterpret_cast<u32x2*>(smem + gb + slot * 16) = __builtin_bit_cast(u32x2, h);
          *reinterpret_cast<u32x2*>(smem + gb + YPL + slot * 16) = __builtin_bit_cast(u32x2, l);
        }
      }
    }
#endif
    ST_ACC(1);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();  // Y1 is complete, the patch is free
    ST_ACC(2);
    if (have_next) load_patch(tn, toy, tox);  // the next tile's patch travels during the matrix phase

    // ---- 2. conv2 on the matrix cores: wave = 16 output channels x 2 rows; step = (tap, chunk) ------------------------------------
    f32x4 acc2[TH];
    acc2[0] = acc2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int L = st_fresh(lane);
    const int wrow = L & 15, wsrc = (L & 48) | ((wrow & 3) | ((wrow & 4) << 1) | ((wrow & 8) >> 1));
    const int cq = ((L >> 4) & 1) * 8 + (L >> 5) * 4;
    {
      const int wv = (wave * 128 + wsrc) * 16;
      const int yb = ((L >> 4) * SL + (L & 15)) * 16;
      constexpr int STEPS = 18;
      auto wf = [&](int step, int p) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(w2r, wv + p * 1024, step * (4 * 2048), 0); };
      auto yoff = [&](int step, int ms) {  // step = tap * 2 + chunk
        const int tap = step >> 1, ch = step & 1, ky = tap / 3, kx = tap % 3;
        return ch * YCH + (((kx & 1) * PH + 2 * ms + ky) * PWh + (kx >> 1)) * 16;
      };
      u32x4 B[3][2];
      B[0][0] = wf(0, 0); B[0][1] = wf(0, 1);
      B[1][0] = wf(1, 0); B[1][1] = wf(1, 1);
#ifndef ST_SKIP2
#pragma unroll
      for (int step = 0; step < STEPS; step++) {
        if (step + 2 < STEPS) { B[(step + 2) % 3][0] = wf(step + 2, 0); B[(step + 2) % 3][1] = wf(step + 2, 1); }
        u32x4 Xf[TH][2];
#pragma unroll
        for (int ms = 0; ms < TH; ms++) {
          Xf[ms][0] = *reinterpret_cast<const u32x4*>(smem + yb + yoff(step, ms));
          Xf[ms][1] = *reinterpret_cast<const u32x4*>(smem + yb + yoff(step, ms) + YPL);
        }
#pragma unroll
        for (int ms = 0; ms < TH; ms++) {
          f32x4 c = acc2[ms];
          c = st_mfma(B[step % 3][1], Xf[ms][0], c);
          c = st_mfma(B[step % 3][0], Xf[ms][1], c);
          acc2[ms] = st_mfma(B[step % 3][0], Xf[ms][0], c);
        }
      }
#endif
    }
    ST_ACC(3);
    __builtin_amdgcn_s_setprio(P2_VALU_PRIO);
    // ---- BN2 + ReLU + max |x| + split, 16-byte stores ------------------------------------------------------------------------------
    float amax = 0.f;
    {
      const int c0 = wave * 16 + cq;
      const f32x4 s2u = sc2 * (m1_inv * w2u), sh2 = sh2v;
      const int xo = ox0 + (L & 15);
      const unsigned vb = xo < a.W2 ? (unsigned)n * out_img + ((L >> 5) ? out_plane : 0u) + (unsigned)(((c0 >> 3) * a.H2 + oy0) * a.W2 + xo) * 16u : 0x80000000u;
#pragma unroll
      for (int ms = 0; ms < TH; ms++) {
        const bool ok = oy0 + ms < a.H2;  // (uniform)
        f32x4 r = acc2[ms] * s2u + sh2;
        r.x = p2_max_nan(r.x, 0.f); r.y = p2_max_nan(r.y, 0.f); r.z = p2_max_nan(r.z, 0.f); r.w = p2_max_nan(r.w, 0.f);
        if (ok && xo < a.W2) amax = conv_amax4(amax, r.x, r.y, r.z, r.w);
        f16x4 h, l;
        p2_split(r * out_mul, h, l);
        const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
        const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
        // (the row offset goes into the vector offset: conv_p2.hip on the x4-store / SGPR-soffset hazard)
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){s0[0], s1[0], s0[1], s1[1]}, orr,
                                               __builtin_elementwise_add_sat(vb, ok ? (unsigned)(ms * a.W2) * 16u : 0x80000000u), 0, 0);
        asm volatile("s_nop 1");
      }
    }
    {
      const unsigned amax_bits = p2_wave_umax(__float_as_uint(amax));
      if (lane == 0) {
        atomicMax(&wgred[0], amax_bits);
        if (atomicAdd(&wgred[1], 1u) == 3u) {
          const unsigned m = atomicExch(&wgred[0], 0u);
          wgred[1] = 0u;
          p2_slot_put(a.out_row + (int64_t)n * P2_ROW, (oy0 / TH) * a.tiles_x + ox0 / TW, tiles_img, m);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    ST_ACC(4);
    if (!have_next) break;
    x_amax_cur = p2_row_amax(row_in);  // (the next tile's image)
    p2_scale_of(x_amax_cur, x_mul_cur, x_inv_cur);
    store_patch(x_mul_cur);  // (every wave passed the barrier behind conv1: the patch buffer is free)
    ST_ACC(5);
    __syncthreads();  // every wave is done with Y1, the next patch is visible
    ST_ACC(6);
    tile = next_tile;
  }
  ST_FLUSH;
}

int mval_conv_stem_p2_supported(int N, int H, int W) {
  if (H < 16 || W < 64 || (H & 3) || (W & 3)) return 0;
  if ((int64_t)N * 3 * H * W * 4 >= ((int64_t)1 << 31)) return 0;  // byte offsets into the image below 2^31
  return 1;
}

int mval_launch_conv_stem_p2(const float* in, void* out, const float* w1, const float* scale1, const float* shift1, const float* bound1,
                             const float* w2, const float* w2_unscale, const float* scale2, const float* shift2, const float* bound2,
                             unsigned* in_row, unsigned* out_row, int N, int H, int W, hipStream_t s) {
  if (!mval_conv_stem_p2_supported(N, H, W)) return 1;
  StemP2Args a = {};
  a.in = in; a.out = reinterpret_cast<_Float16*>(out);
  a.w1 = w1; a.scale1 = scale1; a.shift1 = shift1; a.bound1 = bound1;
  a.w2 = w2; a.w2_unscale = w2_unscale; a.scale2 = scale2; a.shift2 = shift2; a.bound2 = bound2;
  a.in_row = in_row; a.out_row = out_row;
  a.N = N; a.H = H; a.W = W;
  a.H1 = (H - 1) / 2 + 1; a.W1 = (W - 1) / 2 + 1;
  a.H2 = (a.H1 - 1) / 2 + 1; a.W2 = (a.W1 - 1) / 2 + 1;
  a.tiles_x = (a.W2 + 15) / 16;
  a.tiles_y = (a.H2 + 1) / 2;
  const int tiles_img = a.tiles_x * a.tiles_y;
  a.tiles_total = tiles_img * N;
  a.tiles_img_magic = tiles_img > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)tiles_img + 1) : 0u;
  a.tiles_x_magic = a.tiles_x > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)a.tiles_x + 1) : 0u;
  // max |x| per image: 64 partial slots of the input's rows
  hipLaunchKernelGGL(image_amax_rows_kernel, dim3(64, (unsigned)N), dim3(256), 0, s, in, (int64_t)3 * H * W, in_row);
  constexpr size_t smem = 2 * 8 * 170 * 16 + 3 * 11 * 68 * 4 + 16 + 512;
  static std::atomic<int> occ{0};
  int per_cu = p2_resident_wgs(&conv_stem_p2_kernel, occ, smem, 4);
#ifdef P2_TUNE  // (measurement builds only: workgroups per CU)
  const char* pe = getenv("MVAL_P2_WGS_BS");
  if (pe && atoi(pe) > 0) per_cu = atoi(pe);
#endif
  int wgs = mval_cu_count() * per_cu;
  if (wgs >= a.tiles_total) wgs = a.tiles_total;
  else {
    const int per = (a.tiles_total + 7) / 8, rounds = (per + wgs / 8 - 1) / (wgs / 8);
    wgs = 8 * ((per + rounds - 1) / rounds);
  }
#ifdef P2_STAMP
  a.dbg = g_p2_dbg_shared;
#endif
  a.wgs_x = wgs;
  if (tiles_img > P2_SLOTS) mval_launch_zero_rows(out_row, (int64_t)N * P2_ROW, s);
  hipLaunchKernelGGL(conv_stem_p2_kernel, dim3((unsigned)wgs), dim3(256), smem, s, a);
  return 0;
}
