// Fused BasicBlock over P2 activations (conv_p2.h), hrnet.py:19-52 in eval mode:
//
//     out = relu( bn2(conv3x3(relu(bn1(conv3x3(x))))) + x )          x, out: P2 planes, C = 32 or 64 channels
//
// One launch per block, the intermediate activation never leaves the CU, x is read from HBM ONCE (with its halo) and
// the residual comes out of that LDS copy.  At 46 us a single 32 -> 32 conv on 128 64x64 maps already moves its
// 228 MB at ~83 % of what HBM delivers: the two convs of a block cannot get cheaper one by one, only together.
//
// Arithmetic = conv_p2.hip (three fp16 MFMA products per fp32 product, fp32 accumulate, power-of-two scales undone in
// the epilogues).  The intermediate is scaled by a per-IMAGE bound (A1 max|x| + B1, conv_p2.h), known before the first
// MFMA: no tile-maximum reduction, no barrier for it, and the result does not depend on the tiling.
//
// Workgroup = 4 waves on an 8 x 16 output tile, persistent over an XCD-contiguous range of tiles:
//   X   the 12 x 20 input patch (2-pixel halo), [chunk][plane h,l][8-channel block][slot][16 B], staged by COPY;
//   1.  conv1 over the 10 x 18 intermediate pixels (12 sub-tiles of 16 slots): a wave owns 16 output channels
//       (C = 32: and half of the sub-tiles), so every weight fragment is fetched by as few waves as possible -- the
//       L2 -> register weight stream of the fp32-activation block kernel was as busy as its matrix pipe;
//   2.  BN1 + ReLU + zero outside the image, scaled, split, lanes l / l + 32 exchange halves (v_permlane32_swap) and
//       store 16-byte granules into M (same layout as X);  C = 64: M overlays X (160 KB of LDS hold two workgroups);
//   3.  conv2 over the 8 x 16 output pixels from M with row sharing; BN2 + residual (read from X before it is given
//       up) + ReLU + max |x| + split in registers, 16-byte stores into the output planes;
//   the NEXT tile's patch travels to registers meanwhile and is stored as soon as X is free; weight fragments are
//   requested a step / a column ahead, also from conv1 into conv2 and from conv2 into the next tile's conv1.
#include <stdlib.h>

#include "conv_p2.h"

#ifndef P2_VALU_PRIO
#define P2_VALU_PRIO 2
#endif
#ifndef BP_ORDER
#define BP_ORDER 1      // 0: the next tile's patch granules / the scale row requested at the start of the tile (rounds 3-5); 1: behind weight requests
#endif
#ifndef BP_MFMA_PRIO
#define BP_MFMA_PRIO 1  // s_setprio around every step's MFMA group (conv_p2.hip P2_MFMA_PRIO)
#endif

typedef p2_f32x4 f32x4;
typedef p2_f16x8 f16x8;
typedef p2_f16x4 f16x4;
typedef p2_u32x4 u32x4;
typedef p2_u32x2 u32x2;

#ifdef P2_STAMP
#define BP_T0 unsigned long long bp_t = wall_clock64(), bp_t00 = bp_t; unsigned long long bp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BP_ACC(k)                                 \
  do {                                            \
    const unsigned long long t_ = wall_clock64(); \
    bp_acc[k] += t_ - bp_t;                       \
    bp_t = t_;                                    \
  } while (0)
#define BP_FLUSH                                                                                     \
  do {                                                                                               \
    if (a.dbg && lane == 0) {                                                                        \
      unsigned long long* d_ = a.dbg + ((int64_t)blockIdx.x * 4 + wave) * 16;                        \
      d_[0] = bp_t00; d_[4] = wall_clock64(); d_[1] = d_[0];                                         \
      for (int k_ = 0; k_ < 8; k_++) d_[8 + k_] = bp_acc[k_];                                        \
    }                                                                                                \
  } while (0)
extern unsigned long long* g_p2_dbg_shared;
#else
#define BP_T0
#define BP_ACC(k)
#define BP_FLUSH
#endif

__device__ __forceinline__ f32x4 bp_mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct P2BlockArgs {
  const _Float16* in;
  _Float16* out;
  const float *w1, *w1_unscale, *scale1, *shift1, *bound1;
  const float *w2, *w2_unscale, *scale2, *shift2, *bound2;
  const unsigned* in_row;
  unsigned* out_row;
  int N, H, W;
  int tiles_x, tiles_y, tiles_total, wgs_x;
  unsigned tiles_img_magic, tiles_x_magic;
  unsigned long long* dbg;  // diagnostic builds (-DP2_STAMP)
};

template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void conv_block_p2_kernel(P2BlockArgs a) {
  static_assert(C == 32 || C == 64, "fused P2 BasicBlock: 32 or 64 channels");
  constexpr int NS = C / 16, WN = NS, WM = 4 / WN, NCH = C / 32, C8 = C / 8;
  constexpr int TH = 8, TW = 16, XH = TH + 4, XW = TW + 4, MH = TH + 2, MW = TW + 2;
  constexpr int XS = XH * XW, MPX = MH * MW, MSL = 192;  // 240 patch slots; 180 intermediate pixels in 12 sub-tiles
  static_assert(XS % 16 == 0 && MSL % 16 == 0 && MSL >= MPX, "256-byte aligned 8-channel blocks");
  constexpr int MS1 = (MSL / 16) / WM;  // conv1 sub-tiles per wave: 6 (C = 32), 12 (C = 64)
  // (C = 32 with both 16-channel sub-tiles per wave and 3 pixel sub-tiles -- half the LDS fragment reads, twice the weight
  // stream -- measured 83.6 us against 77: the weight stream costs more than the fragment reads save.  Also measured on C = 32, each
  // without effect on the 68 us kernel span: row sharing for conv1 (10 row fragments + one gathered fragment for columns 16 / 17:
  // 20 instead of 36 fragment reads per column tap; conv1 phase 28.0 -> 25.2 us per wave), one ring of three weight columns for
  // both convs requested two columns ahead (-> 23.7 us), no scheduling barriers in the vector phases.  Per SIMD a tile pair
  // needs 3.6 us of MFMA issue and ~5 us of vector phases that run 3-4x below their instruction count: the phases do not overlap)
  constexpr int MS2 = TH / WM;          // conv2 rows per wave: 4, 8
  constexpr bool OVERLAY = C > 32;      // M overlays X
  constexpr int XB = NCH * 8 * XS * 16, MB = NCH * 8 * MSL * 16;
  constexpr int M0 = OVERLAY ? 0 : XB;
  constexpr int XPL = 4 * XS * 16, XCH = 8 * XS * 16;    // plane l behind h, chunk stride (X)
  constexpr int MPL = 4 * MSL * 16, MCH = 8 * MSL * 16;  // (M)
  constexpr int NE = (XH * NCH * 8 * XW + 255) / 256;    // staged granules per thread: 8 / 15
  constexpr int SB = 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* wgred = reinterpret_cast<unsigned*>(smem + (OVERLAY ? XB : XB + MB));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;

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

  // ---- staging plan of the input patch: granule e -> (patch row, block sp = g*8 + plane*4 + c8, column) ----------------
  const unsigned hw16 = (unsigned)(a.H * a.W) * 16u;
  const unsigned img_bytes = 2u * C8 * hw16;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, (unsigned)a.N * img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.N * img_bytes, 0x00020000);
  // (C = 32 keeps the plan in registers; C = 64 -- 15 granules per thread -- recomputes it per tile: 30 registers)
  auto plan_of = [&](int i, unsigned& lp_i, unsigned& cb_i) {
    const int e = tid + 256 * i;
    const int r = e / XW, px = e - r * XW;
    const int sp = r & (NCH * 8 - 1), py = r / (NCH * 8);
    const int g = sp >> 3, pl = (sp >> 2) & 1, c8 = g * 4 + (sp & 3);
    lp_i = py < XH ? ((unsigned)((sp * XS + py * XW + px) * 16) << 16) | (py << 7) | px : 127u;
    cb_i = (unsigned)(pl * C8 + c8) * hw16;
  };
  unsigned lp[OVERLAY ? 1 : NE], cb[OVERLAY ? 1 : NE];
  if constexpr (!OVERLAY) {
#pragma unroll
    for (int i = 0; i < NE; i++) plan_of(i, lp[i], cb[i]);
  }
  u32x4 stage[NE];
  auto load_patch = [&](int n, int oy0, int ox0) {
    const unsigned nbase = (unsigned)n * img_bytes;
#pragma unroll
    for (int i = 0; i < NE; i++) {
      unsigned lpi, cbi;
      if constexpr (OVERLAY) plan_of(i, lpi, cbi);
      else { lpi = lp[i]; cbi = cb[i]; }
      const int px = lpi & 127, py = (lpi >> 7) & 31;
      const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
      const bool inb = px != 127 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, inb ? nbase + cbi + (unsigned)(iy * a.W + ix) * 16u : 0x80000000u, 0, 0);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NE; i++) {
      unsigned lpi, cbi;
      if constexpr (OVERLAY) plan_of(i, lpi, cbi);
      else { lpi = lp[i]; cbi = cb[i]; }
      (void)cbi;
      if ((lpi & 127) != 127) *reinterpret_cast<u32x4*>(smem + (lpi >> 16)) = stage[i];
    }
  };

  // ---- weights: the lane SUPPLIES MFMA row wsrc (rows 4..7 <-> 8..11 swapped, conv_p2.hip), so lanes l / l + 32 own the
  // two halves of one 8-channel granule.  A wave fetches the fragments of ITS 16 output channels only. -----------------
  const int wrow = lane & 15, wsrc = (lane & 48) | ((wrow & 3) | ((wrow & 4) << 1) | ((wrow & 8) >> 1));
  const int cq = ((lane >> 4) & 1) * 8 + (lane >> 5) * 4;
  const int c0 = wn * 16 + cq;  // first of the lane's four output channels (both convs)
  const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w2), 0, 0x7fffffff, 0x00020000);
  constexpr int BLK = NS * 2048;  // bytes per (tap, chunk) block: NS sub-tiles x 2 planes x 1 KiB
  const int wl = (wn * 128 + wsrc) * 16;
  auto w1f = [&](int tap, int ch, int p) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(w1r, wl + p * 1024, (tap * NCH + ch) * BLK, 0); };
  auto w2f = [&](int tap, int ch, int p) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(w2r, wl + p * 1024, (tap * NCH + ch) * BLK, 0); };
  const float w1u = *a.w1_unscale, w2u = *a.w2_unscale;
  const float b1a = a.bound1[0], b1b = a.bound1[1], b2a = a.bound2[0], b2b = a.bound2[1];
  const f32x4 sc1 = *reinterpret_cast<const f32x4*>(a.scale1 + c0), sh1 = *reinterpret_cast<const f32x4*>(a.shift1 + c0);
  const f32x4 sc2 = *reinterpret_cast<const f32x4*>(a.scale2 + c0), sh2 = *reinterpret_cast<const f32x4*>(a.shift2 + c0);

  // ---- per-lane LDS addresses ------------------------------------------------------------------------------------------
  // conv1: sub-tile ms of the wave = intermediate pixels (wm * MS1 + ms) * 16 + (lane & 15); its tap-(0,0) patch slot
  int xb1[MS1];
#pragma unroll
  for (int ms = 0; ms < MS1; ms++) {
    const int p = (wm * MS1 + ms) * 16 + (lane & 15);
    const int pc = p < MPX ? p : 0;
    const int my = pc / MW, mx = pc - my * MW;
    xb1[ms] = ((lane >> 4) * XS + my * XW + mx) * 16;
  }
  // conv2 (row sharing): rows wm * MS2 .. of the tile, fragment = 16 consecutive slots of an intermediate row
  const int mb2 = M0 + ((lane >> 4) * MSL + wm * MS2 * MW + (lane & 15)) * 16;
  // granule this lane writes into M / reads from X: 8-channel block c0 >> 3, plane by lane half
  const int gch = (c0 >> 3) >> 2, gc8 = (c0 >> 3) & 3, gpl = lane >> 5;
  const int m_gran = M0 + gch * MCH + gpl * MPL + gc8 * MSL * 16;
  const int x_gran = gch * XCH + gpl * XPL + gc8 * XS * 16;

  // ---- prologue ----------------------------------------------------------------------------------------------------------
  int tn, toy, tox;
  decode(tile, tn, toy, tox);
  load_patch(tn, toy, tox);
  if (tid == 0) wgred[0] = wgred[1] = 0u;
  u32x4 B1[3][2];  // conv1 weight ring: [step % 3][plane], two steps ahead
  B1[0][0] = w1f(0, 0, 0); B1[0][1] = w1f(0, 0, 1);
  B1[1][0] = w1f(NCH > 1 ? 0 : 1, NCH > 1 ? 1 : 0, 0); B1[1][1] = w1f(NCH > 1 ? 0 : 1, NCH > 1 ? 1 : 0, 1);
  store_patch();
  __syncthreads();
  BP_T0;

  for (;;) {
    const int n = tn, oy0 = toy, ox0 = tox;
    const int next_tile = tile + wgx;
    const bool have_next = next_tile < tile_end;
    if (have_next) decode(next_tile, tn, toy, tox);
    // (round 6, conv_p2.hip P2_ORDER: loads return in order, so the patch granules -- HBM / MALL latency -- requested HERE sat in front of
    // conv1's weight ring and its third step waited for them.  BP_ORDER 1: they are requested behind conv2's first weight column at the end
    // of conv1, travel during the BN1 phase and conv2's first column, and are stored into X at the end of the tile)
    if (BP_ORDER == 0 && !OVERLAY && have_next) load_patch(tn, toy, tox);  // C = 32: the next patch travels during conv1
    P2RowRegs row_in;
    if (BP_ORDER == 0) p2_row_request(a.in_row, n, row_in);

    // ---- 1. conv1: step = (tap, chunk) in packed order chunk-minor; x fragments three sub-tiles ahead -------------------
    f32x4 acc1[MS1];
#pragma unroll
    for (int ms = 0; ms < MS1; ms++) acc1[ms] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 B2[2][3][2];  // conv2 weights [column parity][row tap][plane]
    {
      constexpr int STEPS = 9 * NCH, Q = STEPS * MS1, RD = 3;  // RD x-fragment pairs in flight
      auto xoff = [&](int q) {
        const int step = q / MS1, tap = step / NCH, ch = step % NCH;
        return ch * XCH + ((tap / 3) * XW + (tap % 3)) * 16;
      };
      u32x4 Xf[RD][2];
#pragma unroll
      for (int q = 0; q < RD - 1 && q < Q; q++) {
        Xf[q][0] = *reinterpret_cast<const u32x4*>(smem + xb1[q % MS1] + xoff(q));
        Xf[q][1] = *reinterpret_cast<const u32x4*>(smem + xb1[q % MS1] + xoff(q) + XPL);
      }
#pragma unroll
      for (int step = 0; step < STEPS; step++)
#pragma unroll
      for (int ms = 0; ms < MS1; ms++) {
        const int q = step * MS1 + ms;
        if (ms == 0) {
          if (step + 2 < STEPS) {
            const int s2 = step + 2;
            B1[s2 % 3][0] = w1f(s2 / NCH, s2 % NCH, 0);
            B1[s2 % 3][1] = w1f(s2 / NCH, s2 % NCH, 1);
            if (BP_ORDER != 0 && step == 0) p2_row_request(a.in_row, n, row_in);  // (read after conv1: behind the ring's first request)
          } else if (step + 2 == STEPS) {  // the first column of conv2's weights behind the last steps of conv1
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
              B2[0][ky][0] = w2f(ky * 3, 0, 0);
              B2[0][ky][1] = w2f(ky * 3, 0, 1);
            }
            if (BP_ORDER != 0 && !OVERLAY && have_next) load_patch(tn, toy, tox);
          }
        }
        if (q + RD - 1 < Q) {
          const int q1 = q + RD - 1;
          Xf[q1 % RD][0] = *reinterpret_cast<const u32x4*>(smem + xb1[q1 % MS1] + xoff(q1));
          Xf[q1 % RD][1] = *reinterpret_cast<const u32x4*>(smem + xb1[q1 % MS1] + xoff(q1) + XPL);
        }
        __builtin_amdgcn_sched_barrier(SB);
        if (BP_MFMA_PRIO) __builtin_amdgcn_s_setprio(BP_MFMA_PRIO);
        f32x4 c = acc1[ms];
        c = bp_mfma(B1[step % 3][1], Xf[q % RD][0], c);
        c = bp_mfma(B1[step % 3][0], Xf[q % RD][1], c);
        acc1[ms] = bp_mfma(B1[step % 3][0], Xf[q % RD][0], c);
        if (BP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(SB);
      }
    }

    BP_ACC(0);
    __builtin_amdgcn_s_setprio(P2_VALU_PRIO);  // the vector phases win issue arbitration against the partner wave's MFMA stream
    // ---- residual: the lane's granules of x (its conv2 rows, its channel block) out of the patch, before X is given up --
    u32x4 RX[MS2];
#pragma unroll
    for (int ms = 0; ms < MS2; ms++)
      RX[ms] = *reinterpret_cast<const u32x4*>(smem + x_gran + ((wm * MS2 + ms + 2) * XW + 2 + (lane & 15)) * 16);

    // ---- scales of this image ---------------------------------------------------------------------------------------------
    const float in_inv = __uint_as_float(row_in.inv), x_amax = p2_row_amax(row_in);
    const float mid_bound = b1a * x_amax + b1b;
    float m_mul, m_inv, out_mul, out_inv;
    p2_scale_of(mid_bound, m_mul, m_inv);
    p2_scale_of(b2a * mid_bound + b2b + x_amax, out_mul, out_inv);
    if (oy0 == 0 && ox0 == 0 && tid == 0) a.out_row[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);

    if (OVERLAY) __syncthreads();  // every wave is done with X: M may overwrite it

    // ---- 2. BN1 + ReLU + zero outside the image -> scaled, split, 16-byte granules into M ---------------------------------
    {
      const f32x4 s1u = sc1 * (in_inv * w1u * m_mul), h1u = sh1 * m_mul;  // (both scales are powers of two: exact)
#pragma unroll
      for (int ms = 0; ms < MS1; ms++) {
        const int p = (wm * MS1 + ms) * 16 + (lane & 15);
        const int my = p / MW, mx = p - my * MW;
        const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
        const bool inside = p < MPX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        f32x4 v = acc1[ms] * s1u + h1u;
        v.x = inside ? p2_max_nan(v.x, 0.f) : 0.f;
        v.y = inside ? p2_max_nan(v.y, 0.f) : 0.f;
        v.z = inside ? p2_max_nan(v.z, 0.f) : 0.f;
        v.w = inside ? p2_max_nan(v.w, 0.f) : 0.f;
        f16x4 h, l;
        p2_split(v, h, l);
        const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
        const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
        *reinterpret_cast<u32x4*>(smem + m_gran + p * 16) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_sched_barrier(SB);
      }
    }
    BP_ACC(1);
    __syncthreads();  // M is complete (C = 32: and every wave is done with X)
    BP_ACC(2);
    if (BP_ORDER == 0 && !OVERLAY && have_next) store_patch();

    __builtin_amdgcn_s_setprio(0);
    // ---- 3. conv2 with row sharing over M -------------------------------------------------------------------------------------
    f32x4 acc2[MS2];
#pragma unroll
    for (int ms = 0; ms < MS2; ms++) acc2[ms] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      constexpr int COLS = 3 * NCH, QR = MS2 + 2, Q = COLS * QR;  // column = (chunk, kx)
      auto moff = [&](int q) {
        const int col = q / QR, pr = q % QR;
        return (col / 3) * MCH + (pr * MW + col % 3) * 16;
      };
      u32x4 Xf[2][2];
      Xf[0][0] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(0));
      Xf[0][1] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(0) + MPL);
#pragma unroll
      for (int col = 0; col < COLS; col++)
#pragma unroll
      for (int pr = 0; pr < QR; pr++) {
        const int q = col * QR + pr;
        if (pr == 0) {
          if (col + 1 < COLS) {
            const int c1 = col + 1;
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
              B2[c1 & 1][ky][0] = w2f(ky * 3 + c1 % 3, c1 / 3, 0);
              B2[c1 & 1][ky][1] = w2f(ky * 3 + c1 % 3, c1 / 3, 1);
            }
          } else {  // the first two steps of the next tile's conv1 (same weights every tile)
            B1[0][0] = w1f(0, 0, 0); B1[0][1] = w1f(0, 0, 1);
            B1[1][0] = w1f(NCH > 1 ? 0 : 1, NCH > 1 ? 1 : 0, 0); B1[1][1] = w1f(NCH > 1 ? 0 : 1, NCH > 1 ? 1 : 0, 1);
          }
        }
        if (q + 1 < Q) {
          Xf[(q + 1) & 1][0] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(q + 1));
          Xf[(q + 1) & 1][1] = *reinterpret_cast<const u32x4*>(smem + mb2 + moff(q + 1) + MPL);
        }
        __builtin_amdgcn_sched_barrier(SB);
        const u32x4 xh = Xf[q & 1][0], xl = Xf[q & 1][1];
        if (BP_MFMA_PRIO) __builtin_amdgcn_s_setprio(BP_MFMA_PRIO);
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++) {
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            const int ms = pr - ky;
            if (ms < 0 || ms >= MS2) continue;
            const u32x4* wv = B2[col & 1][ky];
            acc2[ms] = t3 == 0 ? bp_mfma(wv[1], xh, acc2[ms]) : t3 == 1 ? bp_mfma(wv[0], xl, acc2[ms]) : bp_mfma(wv[0], xh, acc2[ms]);
          }
        }
        if (BP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(SB);
      }
    }

    BP_ACC(3);
    __builtin_amdgcn_s_setprio(P2_VALU_PRIO);
    // ---- epilogue: BN2 + residual + ReLU + max |x| + split, 16-byte stores ---------------------------------------------------
    float amax = 0.f;
    {
      const f32x4 s2u = sc2 * (m_inv * w2u);
      const unsigned plane_bytes = C8 * hw16;
      const int xo = ox0 + (lane & 15);
      const unsigned vb = (xo < a.W) ? (unsigned)n * img_bytes + (lane >= 32 ? plane_bytes : 0u) + (unsigned)(((c0 >> 3) * a.H + oy0 + wm * MS2) * a.W + xo) * 16u
                                     : 0x80000000u;
#pragma unroll
      for (int ms = 0; ms < MS2; ms++) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(RX[ms].x, RX[ms].z, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(RX[ms].y, RX[ms].w, false, false);
        const u32x2 rh = {r0[0], r1[0]}, rl = {r0[1], r1[1]};
        f32x4 r = acc2[ms] * s2u + sh2 + p2_join(__builtin_bit_cast(f16x4, rh), __builtin_bit_cast(f16x4, rl)) * in_inv;
        r.x = p2_max_nan(r.x, 0.f); r.y = p2_max_nan(r.y, 0.f); r.z = p2_max_nan(r.z, 0.f); r.w = p2_max_nan(r.w, 0.f);
        const bool ok = oy0 + wm * MS2 + ms < a.H;
        if (ok) amax = conv_amax4(amax, r.x, r.y, r.z, r.w);
        f16x4 h, l;
        p2_split(r * out_mul, h, l);
        const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
        const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
        // (the row offset goes into the vector offset: conv_p2.hip on the x4-store / SGPR-soffset hazard)
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){s0[0], s1[0], s0[1], s1[1]}, orr,
                                               (ok && vb != 0x80000000u) ? vb + (unsigned)(ms * a.W) * 16u : 0x80000000u, 0, 0);
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(SB);
      }
      if (xo >= a.W) amax = 0.f;
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
    BP_ACC(4);
    __builtin_amdgcn_s_setprio(0);
    if (!have_next) break;
    if (BP_ORDER != 0 && !OVERLAY) store_patch();  // (X has been free since the barrier behind the BN1 phase)
    __syncthreads();  // every wave is done with M (C = 64: X may be written; C = 32: the next patch is in X)
    if (OVERLAY) {  // C = 64: 60 staging registers do not fit beside conv2: the next patch is fetched here (the
      load_patch(tn, toy, tox);  // other workgroup of the CU computes meanwhile)
      store_patch();
      __syncthreads();
    }
    BP_ACC(5);
    tile = next_tile;
  }
  BP_FLUSH;
}

static thread_local int g_bp_dry = 0;

template <int C>
static int launch_block_p2(P2BlockArgs a, hipStream_t s) {
  constexpr int NCH = C / 32;
  constexpr size_t XB = (size_t)NCH * 8 * 240 * 16, MB = (size_t)NCH * 8 * 192 * 16;
  constexpr size_t smem = (C > 32 ? XB : XB + MB) + 16;
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + 7) / 8;
  const int tiles_img = a.tiles_x * a.tiles_y;
  a.tiles_total = tiles_img * a.N;
  a.tiles_img_magic = tiles_img > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)tiles_img + 1) : 0u;
  a.tiles_x_magic = a.tiles_x > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)a.tiles_x + 1) : 0u;
  if (g_bp_dry) return 0;
#ifdef P2_STAMP
  a.dbg = g_p2_dbg_shared;
#endif
  static std::atomic<int> occ{0};
  int per_cu = p2_resident_wgs(&conv_block_p2_kernel<C>, occ, smem, 4);
#ifdef P2_TUNE  // (measurement builds only: workgroups per CU)
  const char* pe = getenv("MVAL_P2_WGS");
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
  hipLaunchKernelGGL((conv_block_p2_kernel<C>), dim3((unsigned)wgs), dim3(256), smem, s, a);
  return 0;
}

int mval_conv_block_p2_supported(int C, int N, int H, int W) {
  if (C != 32 && C != 64) return 0;
  if (H < 8 || W < 16) return 0;
  if ((int64_t)N * H * W * C >= (int64_t)1 << 29) return 0;
  return 1;
}

int mval_launch_conv_block_p2(int C, const void* in, void* out, const float* w1, const float* w1_unscale, const float* scale1,
                              const float* shift1, const float* bound1, const float* w2, const float* w2_unscale, const float* scale2,
                              const float* shift2, const float* bound2, const unsigned* in_row, unsigned* out_row, int N, int H, int W,
                              hipStream_t s) {
  if (!mval_conv_block_p2_supported(C, N, H, W)) return 1;
  P2BlockArgs a = {};
  a.in = reinterpret_cast<const _Float16*>(in);
  a.out = reinterpret_cast<_Float16*>(out);
  a.w1 = w1; a.w1_unscale = w1_unscale; a.scale1 = scale1; a.shift1 = shift1; a.bound1 = bound1;
  a.w2 = w2; a.w2_unscale = w2_unscale; a.scale2 = scale2; a.shift2 = shift2; a.bound2 = bound2;
  a.in_row = in_row; a.out_row = out_row;
  a.N = N; a.H = H; a.W = W;
  if (C == 32) return launch_block_p2<32>(a, s);
  return launch_block_p2<64>(a, s);
}
