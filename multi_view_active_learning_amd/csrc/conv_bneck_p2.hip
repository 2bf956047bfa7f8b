// Fused Bottleneck over P2 activations (conv_p2.h), hrnet.py:75-95 in eval mode, planes = 64 (HRNet's layer1):
//
//     out = relu( bn3(conv1x1( relu(bn2(conv3x3( relu(bn1(conv1x1(x))) ))) )) + r )
//           x: CIN = 64 or 256 channels, r: the 256-channel residual (x itself, or the block's downsample branch), out: 256
//
// One launch per block.  Launched op by op the three convs of a 256 -> 64 -> 64 -> 256 block move 2.15 GB for 128
// 64x64 maps (x 537 MB in, twice; the 64-channel intermediates four times; out 537 MB) and their two 1x1 convs sit at
// 4.0 / 5.2 TB/s -- at the HBM roofline, 511 us together.  Fused, the intermediates never leave the CU: x is read
// once (with a 1-pixel halo; neighbouring tiles are walked by the same XCD, so the halo comes out of its L2), the
// residual tile is read again where it is added (out of the 256 MB infinity cache) and out is written once.
//
// Arithmetic = conv_p2.hip (three fp16 MFMA products per fp32 product, fp32 accumulate, power-of-two scales undone in
// the epilogues).  Both intermediates are scaled by per-IMAGE bounds (A1 max|x| + B1, then A2 (that) + B2), known
// before the first MFMA, so the result does not depend on the tiling.
//
// Workgroup = 4 waves on an 8 x 16 output tile, persistent over an XCD-contiguous range of tiles; 80 KB of LDS, two
// workgroups per CU:
//   M1  the 10 x 18 first intermediate (1-pixel halo of the 3x3), 64 channels as [chunk][plane h,l][8-ch block][184 slots][16 B]
//   U   phase 1: ONE 32-channel chunk of the 10 x 18 input patch (same layout), staged by copy, the next chunk
//       travelling in registers meanwhile;  phases 2-3: the 8 x 16 second intermediate M2 (128 slots per block)
//   1.  conv1 (1x1, K = CIN streamed): wave = 32 output channels x 6 of the 12 sixteen-pixel sub-tiles;
//       BN1 + ReLU + zero outside the image -> scaled, split, 16-byte granules into M1;
//   2.  conv2 (3x3 over M1 with row sharing, as conv_block_p2.hip): wave = 16 output channels x 8 rows; BN2 + ReLU
//       -> M2;
//   3.  conv3 (1x1 over M2) in two halves of 128 output channels: wave = 32 channels x 8 rows; BN3 + residual +
//       ReLU + max |x| + split in registers, 16-byte stores into the output planes; residual granules are requested
//       four rows ahead, the next tile's first input chunk during the second half.
#include <stdlib.h>

#include "conv_p2.h"

#ifndef BN_ORDER
#define BN_ORDER 1  // 0: conv1's patch chunk requested before the next chunk's weights (rounds 4-5)
#endif


#ifndef P2_VALU_PRIO
#define P2_VALU_PRIO 2
#endif

typedef p2_f32x4 f32x4;
typedef p2_f16x8 f16x8;
typedef p2_f16x4 f16x4;
typedef p2_u32x4 u32x4;
typedef p2_u32x2 u32x2;

#ifdef P2_STAMP
#define BN_T0 unsigned long long bp_t = wall_clock64(), bp_t00 = bp_t; unsigned long long bp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BN_ACC(k)                                 \
  do {                                            \
    const unsigned long long t_ = wall_clock64(); \
    bp_acc[k] += t_ - bp_t;                       \
    bp_t = t_;                                    \
  } while (0)
#define BN_FLUSH                                                                                     \
  do {                                                                                               \
    if (a.dbg && lane == 0) {                                                                        \
      unsigned long long* d_ = a.dbg + ((int64_t)blockIdx.x * 4 + wave) * 16;                        \
      d_[0] = bp_t00; d_[4] = wall_clock64(); d_[1] = d_[0];                                         \
      for (int k_ = 0; k_ < 8; k_++) d_[8 + k_] = bp_acc[k_];                                        \
    }                                                                                                \
  } while (0)
extern unsigned long long* g_p2_dbg_shared;
#else
#define BN_T0
#define BN_ACC(k)
#define BN_FLUSH
#endif

__device__ __forceinline__ f32x4 bn_mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the three products of one fp32 product, small terms first
__device__ __forceinline__ f32x4 bn_mfma3(const u32x4 wh, const u32x4 wl, const u32x4 xh, const u32x4 xl, f32x4 c) {
  c = bn_mfma(wl, xh, c);
  c = bn_mfma(wh, xl, c);
  return bn_mfma(wh, xh, c);
}
__device__ __forceinline__ f32x4 bn_relu(f32x4 v) {
  v.x = p2_max_nan(v.x, 0.f); v.y = p2_max_nan(v.y, 0.f); v.z = p2_max_nan(v.z, 0.f); v.w = p2_max_nan(v.w, 0.f);
  return v;
}
// scaled fp32 x4 of a lane's four channels -> the 16-byte granule the lane pair (l, l + 32) stores: lanes below 32 get the
// h plane of 8 channels, the others the l plane
__device__ __forceinline__ u32x4 bn_granule(const f32x4 v) {
  f16x4 h, l;
  p2_split(v, h, l);
  const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
  const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
  const auto s1 = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
  return (u32x4){s0[0], s1[0], s0[1], s1[1]};
}
// ... and back: the granule a lane loaded -> its four channels' (h, l) halves joined
__device__ __forceinline__ f32x4 bn_ungranule(const u32x4 g) {
  const auto r0 = __builtin_amdgcn_permlane32_swap(g.x, g.z, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(g.y, g.w, false, false);
  const u32x2 rh = {r0[0], r1[0]}, rl = {r0[1], r1[1]};
  return p2_join(__builtin_bit_cast(f16x4, rh), __builtin_bit_cast(f16x4, rl));
}

struct P2BneckArgs {
  const _Float16* in;
  const _Float16* res;
  _Float16* out;
  const float* params;  // everything below: byte offsets from here, [conv1, conv2, conv3]
  unsigned w[3], w_unscale[3], scale[3], shift[3], bound[3];
  const unsigned* in_row;
  const unsigned* res_row;
  unsigned* out_row;
  int N, H, W;
  int tiles_x, tiles_y, tiles_total, wgs_x;
  unsigned tiles_img_magic, tiles_x_magic;
  unsigned long long* dbg;  // diagnostic builds (-DP2_STAMP)
};

// The compiler may not move what is computed from a laundered value out of the tile loop: per-phase addresses stay
// per-phase registers (hoisted, the address plans of all phases together cost more registers than the kernel has).
__device__ __forceinline__ int bn_fresh(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// the lane SUPPLIES MFMA row wsrc (rows 4..7 <-> 8..11 swapped, conv_p2.hip), so lanes l / l + 32 own the two halves of one
// 8-channel granule
__device__ __forceinline__ int bn_wsrc(int lane) {
  const int wrow = lane & 15;
  return (lane & 48) | ((wrow & 3) | ((wrow & 4) << 1) | ((wrow & 8) >> 1));
}
// first of the lane's four channels inside a 16-channel sub-tile
__device__ __forceinline__ int bn_cq(int lane) { return ((lane >> 4) & 1) * 8 + (lane >> 5) * 4; }

template <int CIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void conv_bneck_p2_kernel(P2BneckArgs a) {
  static_assert(CIN == 64 || CIN == 256, "fused P2 Bottleneck: 64 or 256 input channels, 64 planes");
  constexpr int COUT = 256, NCH1 = CIN / 32, C8I = CIN / 8, C8O = COUT / 8;
  constexpr int TH = 8, TW = 16, MW = TW + 2, MH = TH + 2, MPX = MH * MW;  // 180 intermediate pixels
  constexpr int SL = MPX;                                                  // slots per 8-channel block of X and M1
  constexpr int XPL = 4 * SL * 16;                                         // plane l behind plane h (X chunk, M1 chunk)
  constexpr int M1CH = 8 * SL * 16, M1B = 2 * M1CH;                        // 46 080 bytes
  constexpr int M2SL = TH * TW, M2PL = 4 * M2SL * 16, M2CH = 8 * M2SL * 16;  // M2: 32 768 bytes
  constexpr int U0 = M1B, BN3 = U0 + 2 * M2CH;                             // BN3: scale[256], shift[256] of bn3
  constexpr int XCB = M1CH;  // one staged chunk; chunk c of the input patch lives in region c % 3: M1's halves, then U
  constexpr int NE = (MPX * 8 + 255) / 256;  // staged granules per thread and chunk: 6
  constexpr int SB = 0;
  constexpr int RA = 4;  // residual rows in flight
  static_assert(XCB + 16 * 16 <= 2 * M2CH, "a chunk (and the fragment reads past its last slots) fits U");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* wgred = reinterpret_cast<unsigned*>(smem + BN3 + 2048);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto region = [&](int c) { return (c % 3) == 0 ? 0 : (c % 3) == 1 ? XCB : U0; };

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

  // ---- staging of one 32-channel chunk of the input patch: granule e -> (patch row, block sp = plane*4 + c8, column) ----
  const unsigned hw16 = (unsigned)(a.H * a.W) * 16u;
  const unsigned in_img = 2u * C8I * hw16, out_img = 2u * C8O * hw16, out_plane = C8O * hw16;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, (unsigned)a.N * in_img, 0x00020000);
#ifdef BNECK_NO_RES  // measurement build only (tools/run: what the residual re-read costs): an empty range -- every residual load returns 0 without a memory access
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res), 0, 0u, 0x00020000);
#else
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res), 0, (unsigned)a.N * out_img, 0x00020000);
#endif
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.N * out_img, 0x00020000);
  // ONE descriptor over the parameter buffer: weights, BN vectors and bounds are byte offsets into it
  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.params), 0, 0x7fffffff, 0x00020000);
  unsigned lp[NE];    // LDS byte offset in the chunk << 16 | sp << 12 | py << 7 | px   (px = 127: no granule)
  unsigned goff[NE];  // byte offset of the granule of the tile being staged, chunk 0 (2^31: zero padding)
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e = tid + 256 * i;
    const int r = e / MW, px = e - r * MW;
    const int sp = r & 7, py = r >> 3;
    lp[i] = py < MH ? ((unsigned)((sp * SL + py * MW + px) * 16) << 16) | (sp << 12) | (py << 7) | px : 127u;
  }
  auto plan_tile = [&](int n, int oy0, int ox0) {  // once per tile: where its granules of chunk 0 are
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const unsigned l = (unsigned)bn_fresh((int)lp[i]);
      const int px = l & 127, py = (l >> 7) & 31, sp = (l >> 12) & 7;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      const bool inb = px != 127 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      goff[i] = inb ? (unsigned)n * in_img + (unsigned)((sp >> 2) * C8I + (sp & 3)) * hw16 + (unsigned)(iy * a.W + ix) * 16u : 0x80000000u;
    }
  };
  u32x4 stage[2][NE];  // two chunks in flight (the second set only inside phase 1)
  auto load_x = [&](int ch, int st) {
#pragma unroll
    for (int i = 0; i < NE; i++) stage[st][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, __builtin_elementwise_add_sat(goff[i], (unsigned)ch * 4u * hw16), 0, 0);
  };
  auto store_x = [&](int ro, int st) {
#pragma unroll
    for (int i = 0; i < NE; i++)
      if ((lp[i] & 127u) != 127u) *reinterpret_cast<u32x4*>(smem + ro + (lp[i] >> 16)) = stage[st][i];
  };
  auto wld = [&](unsigned voff, int soff) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(pr, voff, soff, 0); };
  auto pf4 = [&](unsigned off, int c0) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, c0 * 4, off, 0)); };
  const int wn1 = wave & 1, wm1 = wave >> 1;  // conv1: 32 couts (2 sub-tiles) x 6 pixel sub-tiles

  // uniform parameters: scalar loads, once
  auto ps = [&](unsigned off) -> float { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.params) + off); };
  const float b1a = ps(a.bound[0]), b1b = ps(a.bound[0] + 4), b2a = ps(a.bound[1]), b2b = ps(a.bound[1] + 4), b3a = ps(a.bound[2]), b3b = ps(a.bound[2] + 4);
  const float w1u = ps(a.w_unscale[0]), w2u = ps(a.w_unscale[1]), w3u = ps(a.w_unscale[2]);

  // ---- prologue: bn3's vectors into LDS; the first tile's chunks 0 and 1 into their regions, chunk 2 into the staging registers --
  reinterpret_cast<float*>(smem + BN3)[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, tid * 4, a.scale[2], 0));
  reinterpret_cast<float*>(smem + BN3 + 1024)[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pr, tid * 4, a.shift[2], 0));
  int tn, toy, tox;
  decode(tile, tn, toy, tox);
  plan_tile(tn, toy, tox);
  load_x(0, 0);
  if (tid == 0) wgred[0] = wgred[1] = 0u;
  u32x4 W1[2][2][2];  // conv1 weights [chunk parity][sub-tile][plane], a chunk ahead
  {
    const unsigned wv = a.w[0] + (unsigned)((wn1 * 2 * 128 + bn_wsrc(lane)) * 16);
#pragma unroll
    for (int nt = 0; nt < 2; nt++) { W1[0][nt][0] = wld(wv + nt * 2048, 0); W1[0][nt][1] = wld(wv + nt * 2048 + 1024, 0); }
  }
  store_x(region(0), 0);
  load_x(1, 0);
  store_x(region(1), 0);
  if (NCH1 > 2) load_x(2, 0);
  __syncthreads();
  BN_T0;

  for (;;) {
    const int n = tn, oy0 = toy, ox0 = tox;
    const int next_tile = tile + wgx;
    const bool have_next = next_tile < tile_end;
    if (have_next) decode(next_tile, tn, toy, tox);
    P2RowRegs row_in, row_res;
    p2_row_request(a.in_row, n, row_in);
    p2_row_request(a.res_row, n, row_res);

    // ---- 1. conv1: K = CIN in 32-channel chunks; chunk c + 2 is stored and chunk c + 3 requested while chunk c is multiplied --
    f32x4 acc1[6][2];
#pragma unroll
    for (int ms = 0; ms < 6; ms++) acc1[ms][0] = acc1[ms][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 B2[2][3][2];  // conv2 weights [column parity][row tap][plane]
    f32x4 sc1[2], sh1[2];
    {
      const int L = bn_fresh(lane);
      const unsigned wv = a.w[0] + (unsigned)((wn1 * 2 * 128 + bn_wsrc(L)) * 16);
      const unsigned wv2 = a.w[1] + (unsigned)((wave * 128 + bn_wsrc(L)) * 16);
      const int xb1 = ((L >> 4) * SL + wm1 * 96 + (L & 15)) * 16;  // sub-tile ms at + ms * 256 (+ region)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) { sc1[nt] = pf4(a.scale[0], (wn1 * 2 + nt) * 16 + bn_cq(L)); sh1[nt] = pf4(a.shift[0], (wn1 * 2 + nt) * 16 + bn_cq(L)); }
#pragma unroll
      for (int ch = 0; ch < NCH1; ch++) {
        // chunk k: requested at iteration k - 4 into set k & 1, stored at k - 2, multiplied at k (chunks 0 .. 2 of a tile
        // come from the previous tile's phase 3, chunk 3 is requested here)
        if (ch + 2 < NCH1) store_x(region(ch + 2), ch & 1);
        // (round 6, BN_ORDER 1: loads return in order -- the next chunk's weight fragments (L2) are requested BEFORE the patch chunk four
        // ahead (HBM / MALL), so the wait for them at the next chunk does not include that chunk's round trip; conv_p2.hip P2_ORDER)
        if (BN_ORDER == 0) {
          if (ch == 0 && 3 < NCH1) load_x(3, 1);
          if (ch + 4 < NCH1) load_x(ch + 4, ch & 1);
        }
        if (ch + 1 < NCH1) {
#pragma unroll
          for (int nt = 0; nt < 2; nt++) { W1[(ch + 1) & 1][nt][0] = wld(wv + nt * 2048, (ch + 1) * (4 * 2048)); W1[(ch + 1) & 1][nt][1] = wld(wv + nt * 2048 + 1024, (ch + 1) * (4 * 2048)); }
        } else {  // the first column of conv2's weights behind the last chunk
#pragma unroll
          for (int ky = 0; ky < 3; ky++) { B2[0][ky][0] = wld(wv2, (ky * 3 * 2) * (4 * 2048)); B2[0][ky][1] = wld(wv2 + 1024, (ky * 3 * 2) * (4 * 2048)); }
        }
        if (BN_ORDER != 0) {
          if (ch == 0 && 3 < NCH1) load_x(3, 1);
          if (ch + 4 < NCH1) load_x(ch + 4, ch & 1);
        }
        const int xb = xb1 + region(ch);
        u32x4 Xf[2][2];
        Xf[0][0] = *reinterpret_cast<const u32x4*>(smem + xb);
        Xf[0][1] = *reinterpret_cast<const u32x4*>(smem + xb + XPL);
#pragma unroll
        for (int ms = 0; ms < 6; ms++) {
          if (ms + 1 < 6) {
            Xf[(ms + 1) & 1][0] = *reinterpret_cast<const u32x4*>(smem + xb + (ms + 1) * 256);
            Xf[(ms + 1) & 1][1] = *reinterpret_cast<const u32x4*>(smem + xb + (ms + 1) * 256 + XPL);
          }
          __builtin_amdgcn_sched_barrier(SB);
#pragma unroll
          for (int nt = 0; nt < 2; nt++) acc1[ms][nt] = bn_mfma3(W1[ch & 1][nt][0], W1[ch & 1][nt][1], Xf[ms & 1][0], Xf[ms & 1][1], acc1[ms][nt]);
          __builtin_amdgcn_sched_barrier(SB);
        }
        __syncthreads();  // chunk ch + 2 is visible, chunk ch's region is free; after the last chunk: M1's halves are free
      }
    }
    BN_ACC(0);
    __builtin_amdgcn_s_setprio(P2_VALU_PRIO);  // the vector phases win issue arbitration against the partner wave's MFMA stream

    // ---- scales of this image ---------------------------------------------------------------------------------------------
    const float in_inv = __uint_as_float(row_in.inv), x_amax = p2_row_amax(row_in);
    const float res_inv = __uint_as_float(row_res.inv), r_amax = p2_row_amax(row_res);
    const float m1_bound = b1a * x_amax + b1b, m2_bound = b2a * m1_bound + b2b;
    float m1_mul, m1_inv, m2_mul, m2_inv, out_mul, out_inv;
    p2_scale_of(m1_bound, m1_mul, m1_inv);
    p2_scale_of(m2_bound, m2_mul, m2_inv);
    p2_scale_of(b3a * m2_bound + b3b + r_amax, out_mul, out_inv);
    if (oy0 == 0 && ox0 == 0 && tid == 0) a.out_row[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);

    // ---- BN1 + ReLU + zero outside the image -> scaled, split, granules into M1 ---------------------------------------------
    {
      const int L = bn_fresh(lane);
      const float k1 = in_inv * w1u * m1_mul;  // (powers of two: exact)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const f32x4 s1u = sc1[nt] * k1, h1u = sh1[nt] * m1_mul;
        const int gbase = wn1 * M1CH + (L >> 5) * XPL + (nt * 2 + ((L >> 4) & 1)) * SL * 16;
#pragma unroll
        for (int ms = 0; ms < 6; ms++) {
          const int p = (wm1 * 6 + ms) * 16 + (L & 15);
          const int my = p / MW, mx = p - my * MW;
          const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
          const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          f32x4 v = bn_relu(acc1[ms][nt] * s1u + h1u);
          if (!inside) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          const u32x4 g = bn_granule(v);
          if (p < MPX) *reinterpret_cast<u32x4*>(smem + gbase + p * 16) = g;
          __builtin_amdgcn_sched_barrier(SB);
        }
      }
    }
    BN_ACC(1);
    __syncthreads();  // M1 is complete
    BN_ACC(2);
    __builtin_amdgcn_s_setprio(0);

    // ---- 2. conv2 with row sharing over M1: wave = 16 output channels x 8 rows ----------------------------------------------
    f32x4 acc2[TH];
#pragma unroll
    for (int ms = 0; ms < TH; ms++) acc2[ms] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 W3[2][2][2];  // conv3 weights of a half [chunk][sub-tile][plane]
    {
      const int L = bn_fresh(lane);
      const unsigned wv2 = a.w[1] + (unsigned)((wave * 128 + bn_wsrc(L)) * 16);
      const unsigned wv3 = a.w[2] + (unsigned)((wave * 2 * 128 + bn_wsrc(L)) * 16);
      const int mb2 = ((L >> 4) * SL + (L & 15)) * 16;  // row fragments of M1
      const f32x4 sc2 = pf4(a.scale[1], wave * 16 + bn_cq(L)), sh2 = pf4(a.shift[1], wave * 16 + bn_cq(L));
      constexpr int COLS = 3 * 2, QR = TH + 2, Q = COLS * QR;  // column = (chunk, kx)
      auto moff = [&](int q) {
        const int col = q / QR, pr = q % QR;
        return (col / 3) * M1CH + (pr * MW + col % 3) * 16;
      };
      u32x4 Xf[2][2];
      Xf[0][0] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(0));
      Xf[0][1] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(0) + XPL);
#pragma unroll
      for (int col = 0; col < COLS; col++)
#pragma unroll
      for (int pr = 0; pr < QR; pr++) {
        const int q = col * QR + pr;
        if (pr == 0 && col + 1 < COLS) {
          const int c1 = col + 1;
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            B2[c1 & 1][ky][0] = wld(wv2, ((ky * 3 + c1 % 3) * 2 + c1 / 3) * (4 * 2048));
            B2[c1 & 1][ky][1] = wld(wv2 + 1024, ((ky * 3 + c1 % 3) * 2 + c1 / 3) * (4 * 2048));
          }
        }
        if (q + 1 < Q) {
          Xf[(q + 1) & 1][0] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(q + 1));
          Xf[(q + 1) & 1][1] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(q + 1) + XPL);
        }
        __builtin_amdgcn_sched_barrier(SB);
        const u32x4 xh = Xf[q & 1][0], xl = Xf[q & 1][1];
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++) {
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            const int ms = pr - ky;
            if (ms < 0 || ms >= TH) continue;
            const u32x4* wv = B2[col & 1][ky];
            acc2[ms] = t3 == 0 ? bn_mfma(wv[1], xh, acc2[ms]) : t3 == 1 ? bn_mfma(wv[0], xl, acc2[ms]) : bn_mfma(wv[0], xh, acc2[ms]);
          }
        }
        __builtin_amdgcn_sched_barrier(SB);
      }
      BN_ACC(3);
      // conv3 weights of the first half travel during the vector phase
#pragma unroll
      for (int ch = 0; ch < 2; ch++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) { W3[ch][nt][0] = wld(wv3 + nt * 2048, ch * (16 * 2048)); W3[ch][nt][1] = wld(wv3 + nt * 2048 + 1024, ch * (16 * 2048)); }
      __builtin_amdgcn_s_setprio(P2_VALU_PRIO);
      // ---- BN2 + ReLU -> M2 (in U) ------------------------------------------------------------------------------------------
      const f32x4 s2u = sc2 * (m1_inv * w2u * m2_mul), h2u = sh2 * m2_mul;
      const int gbase = U0 + (wave >> 1) * M2CH + (L >> 5) * M2PL + ((wave & 1) * 2 + ((L >> 4) & 1)) * M2SL * 16 + (L & 15) * 16;
#pragma unroll
      for (int ms = 0; ms < TH; ms++) {
        *reinterpret_cast<u32x4*>(smem + gbase + ms * 256) = bn_granule(bn_relu(acc2[ms] * s2u + h2u));
        __builtin_amdgcn_sched_barrier(SB);
      }
    }
    BN_ACC(4);
    __syncthreads();  // M2 is complete, M1's halves are free: they take the next tile's chunks 0 and 1 during phase 3
    __builtin_amdgcn_s_setprio(0);

    // ---- 3. conv3 over M2, two halves of 128 output channels: wave = 32 channels x 8 rows -------------------------------------
    float amax = 0.f;
    const int rows_ok = min(TH, a.H - oy0);
    if (have_next) plan_tile(tn, toy, tox);
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      const int L = bn_fresh(lane);
      const int xb3 = U0 + ((L >> 4) * M2SL + (L & 15)) * 16;  // row r of M2 at + r * 256
      const int xo = ox0 + (L & 15);
      // this lane's granules of the residual / output: 8-channel block ((hf * 8 + wave * 2 + nt) * 2 + gsel), plane by lane half
      unsigned vb[2];
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const int cblk = ((hf * 8 + wave * 2 + nt) * 2 + ((L >> 4) & 1));
        vb[nt] = xo < a.W ? (unsigned)n * out_img + ((L >> 5) ? out_plane : 0u) + (unsigned)((cblk * a.H + oy0) * a.W + xo) * 16u : 0x80000000u;
      }
      // rows below the image: a UNIFORM row offset of 2^31; lanes right of it: vb = 2^31 -- the saturating sum stays out of the
      // buffer's range (no lane masks to keep)
      auto roff = [&](int r) -> unsigned { return r < rows_ok ? (unsigned)(r * a.W) * 16u : 0x80000000u; };
      u32x4 RES[RA][2];  // residual granules, RA rows ahead: the first RA rows travel during the matrix phase
#pragma unroll
      for (int r = 0; r < RA; r++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) RES[r][nt] = __builtin_amdgcn_raw_buffer_load_b128(rr, __builtin_elementwise_add_sat(vb[nt], roff(r)), 0, 0);
      if (have_next) load_x(hf, 0);  // the next tile's chunk hf travels during this half's matrix phase

      f32x4 acc3[TH][2];
#pragma unroll
      for (int r = 0; r < TH; r++) acc3[r][0] = acc3[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      {
        u32x4 Xf[2][2];
        Xf[0][0] = *reinterpret_cast<const u32x4*>(smem + xb3);
        Xf[0][1] = *reinterpret_cast<const u32x4*>(smem + xb3 + M2PL);
#pragma unroll
        for (int q = 0; q < 2 * TH; q++) {
          const int ch = q / TH, r = q % TH;
          if (q + 1 < 2 * TH) {
            const int o1 = ((q + 1) / TH) * M2CH + ((q + 1) % TH) * 256;
            Xf[(q + 1) & 1][0] = *reinterpret_cast<const u32x4*>(smem + xb3 + o1);
            Xf[(q + 1) & 1][1] = *reinterpret_cast<const u32x4*>(smem + xb3 + o1 + M2PL);
          }
          __builtin_amdgcn_sched_barrier(SB);
#pragma unroll
          for (int nt = 0; nt < 2; nt++) acc3[r][nt] = bn_mfma3(W3[ch][nt][0], W3[ch][nt][1], Xf[q & 1][0], Xf[q & 1][1], acc3[r][nt]);
          __builtin_amdgcn_sched_barrier(SB);
        }
      }
      BN_ACC(5);
      // the other half's weights (or the next tile's first conv1 weights) behind the matrix phase; the staged chunk goes into
      // its region (a half of M1, free since the barrier before phase 3) and the next one is requested
      if (hf == 0) {
        const unsigned wv3 = a.w[2] + (unsigned)((wave * 2 * 128 + bn_wsrc(L)) * 16);
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) { W3[ch][nt][0] = wld(wv3 + nt * 2048, ch * (16 * 2048) + 8 * 2048); W3[ch][nt][1] = wld(wv3 + nt * 2048 + 1024, ch * (16 * 2048) + 8 * 2048); }
      } else {
        const unsigned wv = a.w[0] + (unsigned)((wn1 * 2 * 128 + bn_wsrc(L)) * 16);
#pragma unroll
        for (int nt = 0; nt < 2; nt++) { W1[0][nt][0] = wld(wv + nt * 2048, 0); W1[0][nt][1] = wld(wv + nt * 2048 + 1024, 0); }
      }
      if (have_next) {
        store_x(region(hf), 0);
        if (hf == 1 && NCH1 > 2) load_x(2, 0);
      }
      __builtin_amdgcn_s_setprio(P2_VALU_PRIO);
      // ---- BN3 + residual + ReLU + max |x| + split, 16-byte stores ------------------------------------------------------------
      {
        const float k3 = m2_inv * w3u;
        f32x4 s3u[2], sh3[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
          const int c0 = (hf * 8 + wave * 2 + nt) * 16 + bn_cq(L);
          s3u[nt] = *reinterpret_cast<const f32x4*>(smem + BN3 + c0 * 4) * k3;
          sh3[nt] = *reinterpret_cast<const f32x4*>(smem + BN3 + 1024 + c0 * 4);
        }
        // two passes: loads and stores share one in-order counter, so a residual load issued behind a store is not usable
        // before that store is acknowledged -- first every granule is finished in registers (loads only), then all stores
        float am = 0.f;
        u32x4 G[TH][2];
#pragma unroll
        for (int r = 0; r < TH; r++) {
#pragma unroll
          for (int nt = 0; nt < 2; nt++) {
            const f32x4 v = bn_relu(acc3[r][nt] * s3u[nt] + sh3[nt] + bn_ungranule(RES[r % RA][nt]) * res_inv);
            if (r < rows_ok) am = conv_amax4(am, v.x, v.y, v.z, v.w);
            G[r][nt] = bn_granule(v * out_mul);
            if (r + RA < TH) RES[r % RA][nt] = __builtin_amdgcn_raw_buffer_load_b128(rr, __builtin_elementwise_add_sat(vb[nt], roff(r + RA)), 0, 0);
          }
          __builtin_amdgcn_sched_barrier(SB);
        }
        if (xo < a.W) amax = fmaxf(amax, am);
#pragma unroll
        for (int r = 0; r < TH; r++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) {
            // (the row offset goes into the vector offset: conv_p2.hip on the x4-store / SGPR-soffset hazard)
            __builtin_amdgcn_raw_buffer_store_b128(G[r][nt], orr, __builtin_elementwise_add_sat(vb[nt], roff(r)), 0, 0);
          }
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(SB);
      }
      BN_ACC(6);
      __builtin_amdgcn_s_setprio(0);
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
    if (!have_next) break;
    __syncthreads();  // every wave is done with M2 (U takes chunk 2), the next tile's chunks 0 and 1 are visible
    BN_ACC(7);
    tile = next_tile;
  }
  BN_FLUSH;
}

static thread_local int g_bn_dry = 0;

template <int CIN>
static int launch_bneck_p2(P2BneckArgs a, hipStream_t s) {
  constexpr size_t smem = 2 * 8 * 180 * 16 + 2 * 8 * 128 * 16 + 2048 + 16;
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + 7) / 8;
  const int tiles_img = a.tiles_x * a.tiles_y;
  a.tiles_total = tiles_img * a.N;
  a.tiles_img_magic = tiles_img > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)tiles_img + 1) : 0u;
  a.tiles_x_magic = a.tiles_x > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)a.tiles_x + 1) : 0u;
  if (g_bn_dry) return 0;
#ifdef P2_STAMP
  a.dbg = g_p2_dbg_shared;
#endif
  static std::atomic<int> occ{0};
  int per_cu = p2_resident_wgs(&conv_bneck_p2_kernel<CIN>, occ, smem, 4);
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
  a.wgs_x = wgs;
  if (tiles_img > P2_SLOTS) mval_launch_zero_rows(a.out_row, (int64_t)a.N * P2_ROW, s);
  hipLaunchKernelGGL((conv_bneck_p2_kernel<CIN>), dim3((unsigned)wgs), dim3(256), smem, s, a);
  return 0;
}

int mval_conv_bneck_p2_supported(int cin, int planes, int N, int H, int W) {
  if (planes != 64 || (cin != 64 && cin != 256)) return 0;
  if (H < 8 || W < 16) return 0;
  if ((int64_t)N * H * W * 256 >= (int64_t)1 << 29) return 0;  // byte offsets into the planes below 2^31
  return 1;
}

int mval_launch_conv_bneck_p2(int cin, const void* in, const void* res, void* out, const float* params, const int64_t* w, const int64_t* w_unscale,
                              const int64_t* scale, const int64_t* shift, const int64_t* bound, const unsigned* in_row,
                              const unsigned* res_row, unsigned* out_row, int N, int H, int W, hipStream_t s) {
  if (!mval_conv_bneck_p2_supported(cin, 64, N, H, W)) return 1;
  P2BneckArgs a = {};
  a.in = reinterpret_cast<const _Float16*>(in);
  a.res = reinterpret_cast<const _Float16*>(res);
  a.out = reinterpret_cast<_Float16*>(out);
  a.params = params;
  for (int i = 0; i < 3; i++) {  // float offsets into the parameter buffer -> byte offsets below 2^31
    if (w[i] < 0 || scale[i] < 0 || shift[i] < 0 || bound[i] < 0 || (w[i] | w_unscale[i] | scale[i] | shift[i] | bound[i]) >= ((int64_t)1 << 28)) return 1;
    a.w[i] = (unsigned)w[i] * 4u; a.w_unscale[i] = (unsigned)w_unscale[i] * 4u; a.scale[i] = (unsigned)scale[i] * 4u;
    a.shift[i] = (unsigned)shift[i] * 4u; a.bound[i] = (unsigned)bound[i] * 4u;
  }
  a.in_row = in_row; a.res_row = res_row; a.out_row = out_row;
  a.N = N; a.H = H; a.W = W;
  if (cin == 64) return launch_bneck_p2<64>(a, s);
  return launch_bneck_p2<256>(a, s);
}
