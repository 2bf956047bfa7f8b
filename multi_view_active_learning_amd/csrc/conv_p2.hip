// Fused conv on the fp16 matrix cores over "P2" activations (conv_p2.h: every activation kept as the pair of fp16
// planes of the fp16x2 split, channel-blocked [n][plane][C/8][H][W][8]).
//
//     out = act(((conv(x, w) * scale + shift + res1) + res2))   [nearest-upsampled by 2^up on store]
//     (hrnet.py:36-52,75-95,199-287 in eval mode: conv + BN + residual add(s) + ReLU + upsample of the fuse layers)
//
// Arithmetic = conv_mfma_split.hip's PL = 2: three v_mfma_f32_16x16x32_f16 per 32-deep k-step (wl*xh, wh*xl, wh*xh,
// fp32 accumulate), power-of-two scales undone exactly in the epilogue.  What differs is where the work sits:
//   * staging is a COPY: 16-byte granules (8 channels of one pixel of one plane) global -> register -> LDS, no
//     conversion, no scale, no split.  LDS image of a 32-channel chunk: [plane h,l][8-channel block 0..3][slot][16 B];
//     a fragment read (16 consecutive slots of one block per 16-lane group) covers 256 consecutive bytes: conflict-free
//     for every tap offset, no row padding.  Stride-2 convs store the patch columns de-interleaved by parity, so their
//     fragments are consecutive slots too;
//   * two LDS buffers, ONE barrier per chunk: the next chunk's granules travel to registers during the MFMA loop and
//     are stored into the other buffer behind it;
//   * the WEIGHT fragment is the MFMA's first operand: a lane ends up with 4 consecutive output channels of one
//     pixel, so BN / residuals / ReLU / max |x| / the output split all happen in registers and the stores go straight
//     to the output planes (8 bytes per lane and plane; a wave instruction writes four full 128-byte lines).  No LDS
//     round trip, no epilogue barrier;
//   * a wave covers 8 pixel sub-tiles per weight fragment on the large problems (half the L2 -> register weight
//     stream per MFMA of the NHWC kernels: that stream was as busy as the matrix pipe).
#include <stdlib.h>

#include <type_traits>

#include "conv_p2.h"

// register budget: 2 waves per SIMD (256 VGPRs); -DP2_W3: 3 for the light configurations (measurement)
#ifdef P2_W3
#define P2_WAVES(MS, NT, EPI) (((MS) * (NT) <= 4 && (EPI) == 0) ? 3 : 2)
#else
// (EPI 3, the training forward: no residual granules, no output split -- 117 .. 147 registers on the light configurations, so a
// third wave per SIMD fits; MVAL measured below)
#ifdef P2_EPI3_W3
#define P2_WAVES(MS, NT, EPI) (((EPI) == 3 && (MS) * (NT) <= 4) ? 3 : 2)
#else
#define P2_WAVES(MS, NT, EPI) 2
#endif
#endif

#ifndef P2_RES_AUX
#define P2_RES_AUX 2  // cache policy of the FIRST residual's loads: non-temporal (nt) -- the residual of a BasicBlock is its input's last use, that of a
                      // fuse chain the partial sum's -- C2 10.08 -> 10.02 ms, C1x16 6.35 -> 6.31, C4 17.82 -> 17.78 (profiles/r05/p2_nt_res*.log); 0 = default policy
#endif
#ifndef P2_VALU_PRIO
#define P2_VALU_PRIO 2
#endif
// ---- round 6: issue order and prefetch depth of the main loop (tools/micro/p2_loop.hip, profiles/r06/p2_loop_order_prefetch.log) --------
// Vector-memory loads return IN ORDER (one vmcnt counter): the wait for a weight fragment also waits for every load issued before it.  Rounds
// 3-5 requested the next stage's patch granules (and, in a tile's last stage, the epilogue's residual granules and factor rows: HBM / MALL
// latency) at the START of a stage, in front of the next column's weight fragments (L2) -- so the first weight wait of every stage also
// waited for them.  In the loop's skeleton: 2.98-3.09 us per stage -> 2.77 (weights first) / 2.51-2.58 (weights two columns ahead) /
// 2.65 (x fragments two steps ahead) / 2.57-2.62 (s_setprio 1 around the MFMA groups) -> 2.41-2.44 together.
#ifndef P2_ORDER
#define P2_ORDER 1      // 0: patch granules / epilogue operands requested before the stage's MFMA loop (rounds 3-5); 1: behind the first weight
                        // request of the stage; 2: at the start of the stage's second column / step
#endif
#ifndef P2_XD
#define P2_XD 3         // x fragment ring: 2 = one step ahead (rounds 3-5), 3 = two steps ahead (light instantiations only: 8 more registers)
#endif
#ifndef P2_WD_RS
#define P2_WD_RS 3      // row-sharing 3x3 kernels: weight columns in flight + in use: 2 = one column ahead, 3 = two (NT = 1 only: 24 more registers)
#endif
#ifndef P2_WD
#define P2_WD 4         // the other kernels: weight steps in the ring (2 = one step ahead; a step is only MS x NT x 3 MFMAs: 192 cycles at MS x NT = 4)
#endif
#ifndef P2_INZ_MAP
#define P2_INZ_MAP 1    // (INZ staging) 1: a pixel's channel blocks on consecutive lanes; 0: a block's pixels on consecutive lanes (measurement)
#endif
#ifndef P2_MFMA_PRIO
#define P2_MFMA_PRIO 1  // s_setprio around every step's MFMA group (0 = off)
#endif

typedef p2_f32x4 f32x4;
typedef p2_f16x8 f16x8;
typedef p2_f16x4 f16x4;
typedef p2_u32x4 u32x4;
typedef p2_u32x2 u32x2;

#ifdef P2_STAMP
// Diagnostic build only: lane 0 of every wave leaves the 100 MHz wall clock at phase boundaries in a.dbg[wave][16].
#define P2_MARK(k)                                                                                                       \
  do {                                                                                                                   \
    if (a.dbg && lane == 0) a.dbg[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NTH / 64) + wave) * 16 + (k)] = wall_clock64(); \
  } while (0)
// phase timers: P2_T0 starts, P2_ACC(k) adds the time since the last P2_T0 / P2_ACC to dbg slot 8 + k
#define P2_T0 unsigned long long p2_t = wall_clock64(); unsigned long long p2_acc[6] = {0, 0, 0, 0, 0, 0}
#define P2_ACC(k)                              \
  do {                                         \
    const unsigned long long t_ = wall_clock64(); \
    p2_acc[k] += t_ - p2_t;                    \
    p2_t = t_;                                 \
  } while (0)
#define P2_FLUSH                                                                                                         \
  do {                                                                                                                   \
    if (a.dbg && lane == 0)                                                                                              \
      for (int k_ = 0; k_ < 6; k_++)                                                                                     \
        a.dbg[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NTH / 64) + wave) * 16 + 8 + k_] = p2_acc[k_];         \
  } while (0)
static unsigned long long* g_p2_dbg = nullptr;
unsigned long long* g_p2_dbg_shared = nullptr;
extern "C" void mval_p2_debug_buffer(void* p) { g_p2_dbg = g_p2_dbg_shared = reinterpret_cast<unsigned long long*>(p); }
#else
#define P2_MARK(k)
#define P2_T0
#define P2_ACC(k)
#define P2_FLUSH
#endif

__device__ __forceinline__ f32x4 p2_mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// KS: 1 or 3 (pad KS / 2); S: stride; G: 32-channel chunks staged per barrier (1x1 convs: 2 or 4 -- one tap per
// chunk is too little MFMA work per barrier); WN x WM waves (couts x pixels); NT cout sub-tiles and MS pixel
// sub-tiles of 16 per wave; NE staged granules per thread; RS: row sharing (3x3 stride 1, 16-wide tiles: the wave's
// MS sub-tiles are consecutive tile rows, a patch-row fragment feeds the three row taps).
//
// A workgroup is PERSISTENT: it walks tiles (an XCD-contiguous range, so neighbouring tiles' halos meet in one L2)
// and, per tile, the K chunks -- one flat sequence of stages, one barrier each.  In-kernel stamps of the
// one-tile-per-workgroup form showed why (128 -> 128 on 16x16 maps, 35 us): the MFMA phases ran near the rate two
// waves per SIMD can share, but every wave spent 2.9 us waiting for its scale rows before anything else, 6.7 us in an
// epilogue whose residual / BN-factor loads started after the last MFMA, and ~1 us per barrier -- half its life, with
// all workgroups of the launch in the same phase at the same time.  Here every load is requested a stage before its
// use: the next stage's granules (the NEXT TILE's first chunk during a tile's last stage), the next weight blocks
// (also across tiles), and at the start of a tile's last stage everything its epilogue needs (scale rows, residual
// granules, BN factors), so the epilogue is arithmetic and stores.
// EPI: 0 = P2 planes out (residuals at the conv resolution), 1 = P2 planes out through the fused nearest upsample (the 1x1
// convs of the fuse layers), 2 = fp32 NCHW out (the heat-map layer), 3 = raw fp32 NHWC out + BatchNorm batch-statistics
// partials (the TRAINING forward, round 4) -- separate instantiations: one kernel with all three epilogues spilled ~50-100
// registers in every hot instantiation.
// OW > 0 ("odd" tiles, round 4: HRNet-W48's 24 x 18 and 12 x 9 maps): the tile is OW = Wout columns wide and
// floor(16 MS WM / OW) rows high -- full-width rows, pixel slot p -> (p / OW, p % OW), the slots past the last whole row are
// padding.  On such maps the power-of-two tiles compute 1.33x (24 x 18 in 8 x 8 tiles) to 2.4x (12 x 9) the pixels that exist
// (HRNet-W48 forced onto P2: 384 -> 384 on 12 x 9 took 134 us against 65 us for the h2 kernel's odd tiles).
// K48 (row-sharing 3x3 kernels, Cin = 48: HRNet-W48's first branch): the second 32-channel chunk holds 16 channels.  Its stage runs TWO
// column steps instead of three: one PAIRED step whose k-octets 0, 1 are channels 32..47 of column tap 0 and octets 2, 3 the same
// channels of column tap 1 (the lanes of the upper octets read their patch fragment one pixel to the right and their weight fragment from
// the next tap's block: per-lane address constants, the packed weights and the LDS image are unchanged), then column tap 2 as before --
// 15 instead of 18 MFMA column steps per tile.
// The kernel body: `by` / `gy` = the workgroup's cout group and the number of cout groups (blockIdx.y / gridDim.y).  (Round 4 also ran the
// two or three first-level stride-2 convs of a fuse layer as ONE launch through this body -- 112 us of launches less per forward back to
// back, but slower as a step under the multi-stream forward, C2 10.17 vs 10.10 ms: removed in round 5)
template <int KS, int S, int G, int WN, int WM, int NT, int MS, int TW, bool RS, int EPI, int OW = 0, bool K48 = false, bool INZ = false, bool BSUM = false>
__device__ __forceinline__ void conv_p2_body(const P2Args& a, const int by, const int gy) {
  static_assert(!BSUM || (EPI == 3 && KS == 3 && S == 1 && OW == 0 && !K48 && !INZ), "BSUM: the 3x3 stride-1 data gradients");
  static_assert(!K48 || (RS && G == 1), "K48: the row-sharing 3x3 kernels");
  static_assert(!INZ || (EPI == 3 && KS == 3 && S == 1 && OW == 0 && !K48), "INZ: the training forward's 3x3 stride-1 convs");
  constexpr int NTH = 64 * WN * WM, TAPS = KS * KS, SPN = 8 * G, SPN_LOG2 = G == 1 ? 3 : G == 2 ? 4 : 5;
  static_assert(KS == 1 || KS == 2 || KS == 3, "kernel size");
  static_assert(KS != 2 || (S == 1 && !RS && (EPI == 0 || EPI == 3)), "parity convs: stride 1, no row sharing, planes or fp32 NHWC out");
  static_assert(G == 1 || G == 2 || G == 4, "chunks per stage");
  static_assert(!RS || (KS == 3 && S == 1 && G == 1), "row sharing is for 3x3 stride 1");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  // the tile is TH x TW output pixels: every patch / LDS quantity is a compile-time constant (tap offsets become
  // ds_read immediates; with a run-time tile the unrolled loop kept ~60 address registers alive)
  constexpr int TWE = OW ? OW : TW;  // tile width in pixels
  constexpr int TH = 16 * MS * WM / TWE, TW_LOG2 = TW == 8 ? 3 : TW == 16 ? 4 : TW == 32 ? 5 : 6;
  static_assert(OW || (TH * TW == 16 * MS * WM && (TW == 8 || TW == 16 || TW == 32 || TW == 64)), "tile shape");
  static_assert(!RS || TW == 16, "row sharing: 16-wide tiles");
  static_assert(OW == 0 || (!RS && (EPI == 0 || EPI == 3) && TH >= 1), "odd tiles: no fused upsample, no row sharing");
  constexpr int PH = (TH - 1) * S + KS, PW = (TWE - 1) * S + KS, PWh = (PW + 1) >> 1;
  constexpr int slots = S == 1 ? PH * PW : 2 * PH * PWh;
  constexpr int PPX = (slots + 15) & ~15;  // slots per 8-channel block (256-byte aligned blocks)
  constexpr int buf_bytes = SPN * PPX * 16;
  constexpr int NE = (PH * SPN * PW + NTH - 1) / NTH;  // staged granules per thread
  static_assert(PH * SPN * PW < 4096 && PH < 32 && PW < 127, "staging plan packing");
  unsigned* wgred = reinterpret_cast<unsigned*>(smem + 2 * buf_bytes);  // [max |x| of the tile, waves that added]
  float* ztab = reinterpret_cast<float*>(smem + 2 * buf_bytes + 16);    // (INZ) [Cin][alpha * 2^s, beta' * 2^s]
  constexpr int NEZ = INZ ? (PH * 4 * G * PW + NTH - 1) / NTH : 1;      // (INZ) staged items per thread: one (pixel, 8 channels) of z = 32 bytes
  const int ns0 = (by * WN + wn) * NT;
  const bool wave_active = ns0 < a.NS_total;

  // ---- tile walk: workgroup b of this cout group -> XCD group b % X, contiguous tile range per XCD group ----------
  const int X = a.wgs_x >= 8 ? 8 : 1;
  const int per = (a.tiles_total + X - 1) / X, wgx = a.wgs_x / X;
  const int xg = (int)blockIdx.x % X;
  int tile = xg * per + (int)blockIdx.x / X;
  const int tile_end = min(a.tiles_total, (xg + 1) * per);
  // (EPI 3) batch-statistics sums of the lane's four output channels per cout sub-tile, over the workgroup's whole tile walk
  float bsum[EPI == 3 ? NT : 1][4], bsq[EPI == 3 ? NT : 1][4];
  float bgmx[BSUM ? NT : 1][4];  // (BSUM) max |masked gradient| per channel
  auto stats_put = [&]() {
    if constexpr (BSUM) {
      if (!a.bs_part || !wave_active) return;
      const int cqs = ((lane >> 4) & 1) * 8 + (lane >> 5) * 4;
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float s1 = p2_row16_sum(bsum[nt][j]), s2 = p2_row16_sum(bsq[nt][j]), mx = p2_row16_max(bgmx[nt][j]);
          const int c = (ns0 + nt) * 16 + cqs + j;
          if ((lane & 15) == 0 && c < a.Cout) {
            const int64_t e = ((int64_t)blockIdx.x * WM + wm) * a.Cout + c;
            a.bs_part[e * 2] = (double)s1;
            a.bs_part[e * 2 + 1] = (double)s2;
            a.bs_gmax[e] = mx;
          }
        }
    } else if constexpr (EPI == 3) {
      if (!a.bn_part || !wave_active) return;
      const int cqs = ((lane >> 4) & 1) * 8 + (lane >> 5) * 4;
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float s1 = p2_row16_sum(bsum[nt][j]), s2 = p2_row16_sum(bsq[nt][j]);
          const int c = (ns0 + nt) * 16 + cqs + j;
          if ((lane & 15) == 0 && c < a.Cout) {
            double* d = a.bn_part + ((int64_t)c * a.bn_slots + (int64_t)blockIdx.x * WM + wm) * 2;
            d[0] = (double)s1;
            d[1] = (double)s2;
          }
        }
    }
  };
  if constexpr (EPI == 3) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int j = 0; j < 4; j++) bsum[nt][j] = bsq[nt][j] = 0.f;
  }
  // (BSUM) the lane's four output channels' BatchNorm factors per cout sub-tile: r = fma(z, alpha, beta') (train_ops.hip bn_affine),
  // xhat = (z - mean) invstd
  f32x4 bs_al[BSUM ? NT : 1], bs_bp[BSUM ? NT : 1], bs_mu[BSUM ? NT : 1], bs_is[BSUM ? NT : 1];
  if constexpr (BSUM) {
    if (a.bs_bound_slot && blockIdx.x == 0 && by == 0 && threadIdx.x == 0) *a.bs_bound_slot = 0u;
    const int cqs = ((lane >> 4) & 1) * 8 + (lane >> 5) * 4;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int c = min(((by * WN + wn) * NT + nt) * 16 + cqs + j, a.Cout - 1);
        const float al = a.bs_invstd[c] * a.bs_gamma[c];
        bs_al[nt][j] = al;
        bs_bp[nt][j] = __builtin_fmaf(-a.bs_mean[c], al, a.bs_beta[c]);
        bs_mu[nt][j] = a.bs_mean[c];
        bs_is[nt][j] = a.bs_invstd[c];
        bgmx[nt][j] = 0.f;
      }
  }
  if (tile >= tile_end) {
    stats_put();  // (a workgroup without tiles still owns its slots of the partials)
    return;
  }
  const int tiles_img = a.tiles_x * a.tiles_y;
  int tn, toy, tox;  // the tile being computed: image, first output row / column
  auto decode = [&](int t, int& n, int& oy0, int& ox0) {  // (t / d as a multiply: magic = 2^32 / d + 1, t * d < 2^32)
    n = a.tiles_img_magic ? (int)__umulhi((unsigned)t, a.tiles_img_magic) : t;
    const int r = t - n * tiles_img;
    const int tyi = a.tiles_x_magic ? (int)__umulhi((unsigned)r, a.tiles_x_magic) : r;
    oy0 = tyi * TH;
    ox0 = (r - tyi * a.tiles_x) * TWE;
  };

  // ---- staging plan: granule e = tid + NTH * i -> (patch row py, block sp = g*8 + plane*4 + c8, column px) ------
  const int C8 = a.Cin >> 3;
  const unsigned hw = (unsigned)(a.Hin * a.Win), hw16 = hw * 16u;  // bytes of one 8-channel block of one plane
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(a.in), 0, (unsigned)min((int64_t)0xffffffff, (int64_t)a.N * 2 * C8 * hw16), 0x00020000);
  unsigned lp[NE];    // LDS byte offset << 16 | c8 << 12 | py << 7 | px   (px = 127: no granule)
  unsigned cb[NE];    // (plane * C8 + c8) * H * W * 16
  unsigned goff[NE];  // byte offset of the granule of the tile being staged, chunk 0 (0xffffffff: zero padding)
  unsigned lpz[NEZ], goffz[NEZ];  // (INZ) the same two per staged item of z (below)
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e = tid + NTH * i;
    const int r = e / PW;
    const int px = e - r * PW;
    const int sp = r & (SPN - 1), py = r >> SPN_LOG2;
    const int g = sp >> 3, pl = (sp >> 2) & 1, c8 = g * 4 + (sp & 3);
    const int slot = S == 1 ? py * PW + px : ((px & 1) * PH + py) * PWh + (px >> 1);
    lp[i] = py < PH ? ((unsigned)((sp * PPX + slot) * 16) << 16) | (c8 << 12) | (py << 7) | px : 127u;
    cb[i] = (unsigned)(pl * C8 + c8) * hw16;
  }
  const int pad_y = KS == 2 ? a.pad_y : KS / 2, pad_x = KS == 2 ? a.pad_x : KS / 2;  // (KS 2: a parity conv's window origin)
  const int isub = KS == 1 ? a.in_sub : 0;                                             // (KS 1: every 2^isub-th input pixel)
  auto plan = [&](int n, int oy0, int ox0) {
    const int iy0 = oy0 * S - pad_y, ix0 = ox0 * S - pad_x;
    if constexpr (INZ) {
#pragma unroll
      for (int i = 0; i < NEZ; i++) {
        const int px = lpz[i] & 127, py = (lpz[i] >> 7) & 31;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool inb = px != 127 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        goffz[i] = inb ? (unsigned)((n * a.Hin + iy) * a.Win + ix) * (unsigned)(a.Cin * 4) + ((lpz[i] >> 12) & 15) * 32u : 0xffffffffu;
      }
      return;
    }
    const unsigned nbase = (unsigned)n * 2u * (unsigned)C8 * hw16;
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const int px = lp[i] & 127, py = (lp[i] >> 7) & 31;
      const int iy = (iy0 + py) << isub, ix = (ix0 + px) << isub;
      const bool inb = px != 127 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      goff[i] = inb ? nbase + cb[i] + (unsigned)(iy * a.Win + ix) * 16u : 0xffffffffu;
    }
  };
  const int nchunks = (a.Cin + 31) >> 5;
  const int nst = (nchunks + G - 1) / G;  // stages per tile
  u32x4 stage[INZ ? 1 : NE];
  // ---- (INZ) the input is relu(BatchNorm(z)) of the producer's raw fp32 NHWC output: item e -> (patch row py, 8-channel block c8l of
  // the stage, column px); two 16-byte loads of z per item, the affine + ReLU + scale + split on the way into LDS (P2Args::in_z) --------
  float z_inv = 1.f;
  u32x4 stagez[NEZ][2];
  const __amdgpu_buffer_rsrc_t zinr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(INZ ? a.in_z : nullptr), 0, INZ ? (unsigned)min((int64_t)0xffffffff, (int64_t)a.N * a.Hin * a.Win * a.Cin * 4) : 0u, 0x00020000);
  if constexpr (INZ) {
    float bnd = 0.f;
    for (int c = lane; c < a.Cin; c += 64) bnd = fmaxf(bnd, __builtin_fmaf(fabsf(a.zin_gamma[c]), a.zin_sqrt_m1, fabsf(a.zin_beta[c])));
    bnd = __uint_as_float(p2_wave_umax(__float_as_uint(bnd))) * (1.f + 1e-6f);
    float z_mul;
    p2_scale_of(bnd, z_mul, z_inv);
    for (int c = tid; c < a.Cin; c += NTH) {
      const float alpha = a.zin_invstd[c] * a.zin_gamma[c];
      const float betap = __builtin_fmaf(-a.zin_mean[c], alpha, a.zin_beta[c]);
      ztab[2 * c] = alpha * z_mul;      // (2^s: exact; fma(z, alpha 2^s, beta' 2^s) == 2^s fma(z, alpha, beta'))
      ztab[2 * c + 1] = betap * z_mul;
    }
#pragma unroll
    for (int i = 0; i < NEZ; i++) {
      // consecutive lanes = the 4 G channel blocks of ONE pixel (32 bytes each: 128 G contiguous bytes of its NHWC row), then the next pixel
      // of the patch row: a wave's two load instructions use every byte of the lines they touch.  (P2_INZ_MAP 0, the first form: consecutive
      // lanes = consecutive pixels of one block, 16 bytes at a 4 C-byte stride -- 64 sectors per instruction; conv forward +0.6 ms per C3 step)
      const int e = tid + NTH * i;
      // (G = 2 -- the 8 x 8 maps' form -- keeps the first mapping: eight blocks of a pixel are eight LDS stores to one bank group;
      // 256 -> 256 @8x8 measured 34.0 us with the first mapping, 36.3 with this one; profiles/r06/inz_lane_map_ab.log)
      int c8l, py, px;
      if constexpr (P2_INZ_MAP && G == 1) {
        c8l = e & 3;
        const int t_ = e >> 2;
        py = t_ / PW;
        px = t_ - py * PW;
      } else {
        const int r = e / PW;
        px = e - r * PW;
        c8l = r & (4 * G - 1);
        py = r / (4 * G);
      }
      const int sp = (c8l >> 2) * 8 + (c8l & 3);  // plane h of chunk c8l / 4; plane l sits plane_b behind
      lpz[i] = py < PH ? ((unsigned)((sp * PPX + py * PW + px) * 16) << 16) | (c8l << 12) | (py << 7) | px : 127u;
    }
    __syncthreads();
  }
  auto load_stage = [&](int st) {
    const int c8b = st * 4 * G;
    if constexpr (INZ) {
#pragma unroll
      for (int i = 0; i < NEZ; i++) {
        const bool v = goffz[i] != 0xffffffffu;
        const unsigned o = goffz[i] + (unsigned)c8b * 32u;
        stagez[i][0] = __builtin_amdgcn_raw_buffer_load_b128(zinr, v ? o : 0xffffffffu, 0, 0);
        stagez[i][1] = __builtin_amdgcn_raw_buffer_load_b128(zinr, v ? o + 16u : 0xffffffffu, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NE; i++) {
        const bool v = goff[i] != 0xffffffffu && c8b + (int)((lp[i] >> 12) & 15) < C8;
        stage[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, v ? goff[i] + (unsigned)c8b * hw16 : 0xffffffffu, 0, 0);
      }
    }
  };
  // (st: the stage the granules belong to -- INZ reads its channels' factors)
  auto store_stage = [&](int buf, int st) {
    if constexpr (INZ) {
      const int c8b = st * 4 * G;
#pragma unroll
      for (int i = 0; i < NEZ; i++) {
        if ((lpz[i] & 127) == 127) continue;
        u32x4 gh = {0u, 0u, 0u, 0u}, gl = {0u, 0u, 0u, 0u};
        if (goffz[i] != 0xffffffffu) {  // (outside the image: the conv's zero padding pads the ACTIVATION, not z)
          const f32x4* t4 = reinterpret_cast<const f32x4*>(ztab + (c8b + (int)((lpz[i] >> 12) & 15)) * 16);
          const f32x4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];  // [a0 b0 a1 b1] [a2 b2 a3 b3] [a4 b4 a5 b5] [a6 b6 a7 b7]
          const f32x4 z0 = __builtin_bit_cast(f32x4, stagez[i][0]), z1 = __builtin_bit_cast(f32x4, stagez[i][1]);
          f32x4 y0, y1;
          y0.x = mval_relu(__builtin_fmaf(z0.x, t0.x, t0.y)); y0.y = mval_relu(__builtin_fmaf(z0.y, t0.z, t0.w));
          y0.z = mval_relu(__builtin_fmaf(z0.z, t1.x, t1.y)); y0.w = mval_relu(__builtin_fmaf(z0.w, t1.z, t1.w));
          y1.x = mval_relu(__builtin_fmaf(z1.x, t2.x, t2.y)); y1.y = mval_relu(__builtin_fmaf(z1.y, t2.z, t2.w));
          y1.z = mval_relu(__builtin_fmaf(z1.z, t3.x, t3.y)); y1.w = mval_relu(__builtin_fmaf(z1.w, t3.z, t3.w));
          f16x4 h0, l0, h1, l1;
          p2_split(y0, h0, l0);
          p2_split(y1, h1, l1);
          const u32x2 a0 = __builtin_bit_cast(u32x2, h0), a1 = __builtin_bit_cast(u32x2, h1);
          const u32x2 b0 = __builtin_bit_cast(u32x2, l0), b1 = __builtin_bit_cast(u32x2, l1);
          gh = (u32x4){a0.x, a0.y, a1.x, a1.y};
          gl = (u32x4){b0.x, b0.y, b1.x, b1.y};
        }
        char* d = smem + buf * buf_bytes + (lpz[i] >> 16);
        *reinterpret_cast<u32x4*>(d) = gh;
        *reinterpret_cast<u32x4*>(d + 4 * PPX * 16) = gl;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NE; i++)
        if ((lp[i] & 127) != 127) *reinterpret_cast<u32x4*>(smem + buf * buf_bytes + (lp[i] >> 16)) = stage[i];
    }
  };

  // ---- fragment addressing ------------------------------------------------------------------------------------
  // x fragment (MFMA B operand): lane -> pixel slot (lane & 15) of the sub-tile, k octet = 8-channel block (lane >> 4)
  int xb[RS ? 1 : MS];
  if constexpr (RS) {
    xb[0] = ((lane >> 4) * PPX + wm * MS * PW + (lane & 15)) * 16;
  } else {
#pragma unroll
    for (int ms = 0; ms < MS; ms++) {
      const int p = (wm * MS + ms) * 16 + (lane & 15);
      int ty = OW ? p / TWE : p >> TW_LOG2, tx = OW ? p - ty * TWE : p & (TW - 1);
      if (OW && ty >= TH) ty = tx = 0;  // padding slot of an odd tile: any address inside the patch
      xb[ms] = ((lane >> 4) * PPX + (S == 1 ? ty * PW + tx : 2 * ty * PWh + tx)) * 16;
    }
  }
  // K48 paired step: octets 0, 1 -> blocks 0, 1 (channels 32..47) at column tap 0; octets 2, 3 -> the same blocks one pixel to the right
  const int xb_pair = K48 ? ((((lane >> 4) & 1) * PPX + wm * MS * PW + (lane & 15) + (lane >> 5)) * 16) : 0;
  constexpr int plane_b = 4 * PPX * 16;  // plane l behind plane h inside a chunk
  constexpr int chunk_b = 8 * PPX * 16;
  // weight fragments through a buffer descriptor: block offset in SGPRs, per-lane 32-bit offset, plane as immediate
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const int blk_bytes = a.NS_total * 2048;
  // The MFMA row (= output channel of the sub-tile) a lane SUPPLIES is permuted: rows 4..7 carry couts 8..11 and rows
  // 8..11 couts 4..7.  Output lanes (quarter q = lane >> 4 holds rows 4q .. 4q+3) then own couts cq(q) = {0, 8, 4, 12}
  // + 0..3, i.e. lanes l and l + 32 hold the two halves of ONE 16-byte granule (8 channels) of the same pixel: one
  // v_permlane32_swap pair turns two 8-byte stores / residual loads per lane into one 16-byte one.
  const int wrow = lane & 15, wsrc = (lane & 48) | ((wrow & 3) | ((wrow & 4) << 1) | ((wrow & 8) >> 1));
  const int cq = ((lane >> 4) & 1) * 8 + (lane >> 5) * 4;  // first cout (inside the sub-tile) of the lane's four
  int wlane[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) wlane[nt] = (min(ns0 + nt, a.NS_total - 1) * 128 + wsrc) * 16;
  auto wfrag = [&](int blk, int nt, int p) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(wr, wlane[nt] + p * 1024, blk * blk_bytes, 0);
  };
  // K48 paired weight fragment of row tap ky: lanes of octets 0, 1 read block (ky, kx = 0, chunk 1) as usual, lanes of octets 2, 3 the
  // fragment lane 32 below theirs (octets 0, 1: channels 32..47) of block (ky, kx = 1, chunk 1) = two blocks further
  const int pair_adj = (K48 && lane >= 32) ? 2 * blk_bytes - 512 : 0;
  auto wfrag_pair = [&](int ky, int nt, int p) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(wr, wlane[nt] + p * 1024 + pair_adj, ((ky * 3) * 2 + 1) * blk_bytes, 0);
  };
  const float w_unscale = *a.w_unscale;
  // output / residual planes through buffer descriptors: per tile ONE per-lane 32-bit byte offset (+ a scalar offset per
  // pixel sub-tile); lanes outside the image or past Cout get offset 2^31 and the range check drops them -- the
  // epilogue has no branches and no 64-bit address arithmetic (the tensors are below 2^31 bytes: the launcher checks)
  const int osh = (EPI == 0 || EPI == 3) ? a.os : 0;  // (parity launches: the conv grid lands on every second pixel of the output)
  const unsigned obytes = (a.out_f32 || EPI == 3) ? 0u : (unsigned)(((int64_t)a.N * (a.Cout >> 3) * (a.Hout << a.up) * (a.Wout << a.up) * 32) << (2 * osh));
  const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(a.out_nhwc, 0, EPI == 3 ? (unsigned)(((int64_t)a.N * a.Hout * a.Wout * a.Cout * 4) << (2 * osh)) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res1), 0, a.res1 ? obytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res2), 0, a.res2 ? obytes : 0u, 0x00020000);
  // lane -> (row, column) inside a 16-pixel sub-tile, and the sub-tile's own (row, column) inside the tile
  const int ly = OW ? 0 : TW == 8 ? (lane >> 3) & 1 : 0, lx = OW ? 0 : TW == 8 ? lane & 7 : lane & 15;
  const float bound_a = (a.out_f32 || EPI == 3) ? 0.f : a.bound[0], bound_b = (a.out_f32 || EPI == 3) ? 0.f : a.bound[1];
  const int Ho = (a.Hout << a.up) << osh, Wo = (a.Wout << a.up) << osh, rep = 1 << a.up;
  const int C8o = a.Cout >> 3;
  const int64_t oplane = (int64_t)C8o * Ho * Wo * 8;  // halves per plane of one image
  // odd tiles: the lane's pixel of sub-tile ms -> its row inside the tile (-1: padding slot) and its byte offset from the
  // tile's first pixel inside an 8-channel block of the output planes
  int opty[OW ? MS : 1], opix[OW ? MS : 1];  // (EPI 3: opix holds the column instead)
  if constexpr (OW > 0) {
#pragma unroll
    for (int ms = 0; ms < MS; ms++) {
      const int p = (wm * MS + ms) * 16 + (lane & 15);
      const int ty = p / TWE, tx = p - ty * TWE;
      opty[ms] = ty < TH ? ty : -1;
      opix[ms] = EPI == 3 ? tx : ((ty << osh) * Wo + (tx << osh)) * 16;
    }
  }

  f32x4 acc[MS][NT];
  constexpr int SB = 0;  // sched_barrier mask: nothing crosses.  Left to itself the compiler sinks every weight load and
                         // LDS fragment read next to its first use (buffer_load; s_waitcnt vmcnt(1); v_mfma).
  constexpr int STEPS = G * TAPS;
  // register budget of the deeper rings: the light instantiations (<= 4 accumulator tiles per wave) have room, the others keep rounds 3-5's depth
  constexpr bool LIGHT = MS * NT <= 4;
  // (by the register tables, tools/kernel_resources.py: two pixel waves stage 6 granules per thread -- x ring only; the stride-2 patch is four
  // times the tile -- one more weight step, no x ring)
  constexpr int XD = (LIGHT && S == 1 && P2_XD >= 3) ? 3 : 2;                                   // x fragment ring
  constexpr int WDR = (RS && NT == 1 && MS <= 4 && WM == 1 && !K48 && P2_WD_RS >= 3) ? 3 : 2;   // row-sharing: weight column ring
  constexpr int WDN_ = LIGHT ? (P2_WD < 2 ? 2 : S == 2 && P2_WD > 3 ? 3 : P2_WD) : 2;
  constexpr int WDN = WDN_ > STEPS + 1 ? STEPS + 1 : WDN_;                            // other kernels: weight step ring
  constexpr int WD = RS ? WDR : WDN;
  u32x4 B[WD][RS ? 3 : 1][NT][2];  // weight fragments [ring slot][row tap (RS)][cout sub-tile][plane]
  // RS: a "column" = the three row taps of column tap kx; else a "step" = one (chunk of the stage, tap) block
  auto wload = [&](int par, int stg, int u) {
    if constexpr (RS) {
      if constexpr (K48) {
        if (stg == 1) {  // the remainder stage: column 0 = the paired step, column 1 = column tap 2
#pragma unroll
          for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
              for (int p = 0; p < 2; p++) B[par][ky][nt][p] = u == 0 ? wfrag_pair(ky, nt, p) : wfrag((ky * 3 + 2) * 2 + 1, nt, p);
          return;
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int p = 0; p < 2; p++) B[par][ky][nt][p] = wfrag((ky * 3 + u) * nchunks + stg, nt, p);
    } else {
      const int blk = (u % TAPS) * nchunks + min(stg * G + u / TAPS, nchunks - 1);
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int p = 0; p < 2; p++) B[par][0][nt][p] = wfrag(blk, nt, p);
    }
  };
  auto toff_of = [&](int step) {
    const int g = step / TAPS, tap = step % TAPS;
    const int ky = tap / KS, kx = tap % KS;
    return g * chunk_b + (S == 1 ? ky * PW + kx : ((kx & 1) * PH + ky) * PWh + (kx >> 1)) * 16;
  };

  // One stage's MFMAs from LDS buffer `buf`; `pre` != 0: request the first weight blocks of stage `st_next` at the end.  `early()` issues the
  // stage's other loads (the next stage's patch granules; in a tile's last stage also the epilogue's operands) at the point P2_ORDER names.
  auto mfma_stage = [&](auto rem_tag, int buf, int st, bool pre, int st_next, auto&& early) {
    constexpr bool REM = decltype(rem_tag)::value;  // (K48) the remainder stage: two column steps
    if constexpr (P2_ORDER == 0) early();
    if constexpr (RS) {
      const char* xs = smem + buf * buf_bytes + xb[0];
      const char* xsp = smem + buf * buf_bytes + xb_pair;
      constexpr int NC = REM ? 2 : 3;
      constexpr int Q = NC * (MS + 2);  // (column step, patch row pr) steps
      // fragment of (column step c, patch row pr): the remainder stage's step 0 is the paired one, its step 1 column tap 2
      auto frag_at = [&](int c, int pr) -> const char* {
        if constexpr (REM) return c == 0 ? xsp + (pr * PW) * 16 : xs + (pr * PW + 2) * 16;
        else return xs + (pr * PW + c) * 16;
      };
      u32x4 Xf[XD][2];
#pragma unroll
      for (int q0 = 0; q0 < XD - 1; q0++) {
        Xf[q0][0] = *reinterpret_cast<const u32x4*>(frag_at(q0 / (MS + 2), q0 % (MS + 2)));
        Xf[q0][1] = *reinterpret_cast<const u32x4*>(frag_at(q0 / (MS + 2), q0 % (MS + 2)) + plane_b);
      }
      auto body = [&](const int q) {
        const int kx = q / (MS + 2), pr = q % (MS + 2);
        if (pr == 0) {
          if constexpr (WDR == 3) {
            // column kx lives in slot kx; request column kx + 2 (kx >= 1: the next stage's column kx - 1) into the slot column kx - 1 left
            if (kx == 0) wload(2, st, 2);
            else if (pre) wload(kx - 1, st_next, kx - 1);
          } else {
            // request the next column's weights (last column: the first column of the next stage, into parity NC & 1)
            if (kx + 1 < NC) wload((kx + 1) & 1, st, kx + 1);
            else if (pre) wload(NC & 1, st_next, 0);
          }
        }
        if (q + XD - 1 < Q) {
          const int kx1 = (q + XD - 1) / (MS + 2), pr1 = (q + XD - 1) % (MS + 2);
          const char* ap = frag_at(kx1, pr1);
          Xf[(q + XD - 1) % XD][0] = *reinterpret_cast<const u32x4*>(ap);
          Xf[(q + XD - 1) % XD][1] = *reinterpret_cast<const u32x4*>(ap + plane_b);
        }
        __builtin_amdgcn_sched_barrier(SB);
        const u32x4 xh = Xf[q % XD][0], xl = Xf[q % XD][1];
        if (P2_MFMA_PRIO) __builtin_amdgcn_s_setprio(P2_MFMA_PRIO);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
#pragma unroll
          for (int t3 = 0; t3 < 3; t3++) {  // small products first, interleaved over the accumulators the fragment feeds
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
              const int ms = pr - ky;
              if (ms < 0 || ms >= MS) continue;
              const u32x4* wv = B[WDR == 3 ? kx : (kx & 1)][ky][nt];
              acc[ms][nt] = t3 == 0 ? p2_mfma(wv[1], xh, acc[ms][nt]) : t3 == 1 ? p2_mfma(wv[0], xl, acc[ms][nt]) : p2_mfma(wv[0], xh, acc[ms][nt]);
            }
          }
        }
        if (P2_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(SB);
      };
      // the stage's other loads go out between two steps (a call inside ONE unrolled loop left loops of `early` rolled: arrays in scratch)
      constexpr int QE = P2_ORDER == 1 ? 1 : P2_ORDER == 2 ? MS + 2 : 0;
#pragma unroll
      for (int q = 0; q < QE; q++) body(q);
      if constexpr (P2_ORDER != 0) early();
#pragma unroll
      for (int q = QE; q < Q; q++) body(q);
      if (WDR == 2 && pre && (NC & 1)) {  // the next stage starts on parity 0
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int p = 0; p < 2; p++) B[0][ky][nt][p] = B[1][ky][nt][p];
      }
    } else {
      const char* xs = smem + buf * buf_bytes;
      constexpr int Q = STEPS * MS;
      u32x4 Xf[XD][2];
#pragma unroll
      for (int q0 = 0; q0 < XD - 1 && q0 < Q; q0++) {
        Xf[q0][0] = *reinterpret_cast<const u32x4*>(xs + xb[q0 % MS] + toff_of(q0 / MS));
        Xf[q0][1] = *reinterpret_cast<const u32x4*>(xs + xb[q0 % MS] + toff_of(q0 / MS) + plane_b);
      }
      auto body = [&](const int q) {
        const int step = q / MS, ms = q % MS;
        if (ms == 0) {  // step u of a stage sits in ring slot u % WDN (the ring is rotated at the end of a stage); request step u + WDN - 1
          const int u = step + WDN - 1;
          if (u < STEPS) wload(u % WDN, st, u);
          else if (pre) wload(u % WDN, st_next, u - STEPS);
        }
        if (q + XD - 1 < Q) {
          const int s1 = (q + XD - 1) / MS, m1 = (q + XD - 1) % MS;
          const char* ap = xs + xb[m1] + toff_of(s1);
          Xf[(q + XD - 1) % XD][0] = *reinterpret_cast<const u32x4*>(ap);
          Xf[(q + XD - 1) % XD][1] = *reinterpret_cast<const u32x4*>(ap + plane_b);
        }
        __builtin_amdgcn_sched_barrier(SB);
        if (G == 1 || st * G + step / TAPS < nchunks) {  // (chunk count not a multiple of G: the tail stage is short)
          const u32x4 xh = Xf[q % XD][0], xl = Xf[q % XD][1];
          if (P2_MFMA_PRIO) __builtin_amdgcn_s_setprio(P2_MFMA_PRIO);
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            f32x4 c = acc[ms][nt];
            c = p2_mfma(B[step % WDN][0][nt][1], xh, c);
            c = p2_mfma(B[step % WDN][0][nt][0], xl, c);
            acc[ms][nt] = p2_mfma(B[step % WDN][0][nt][0], xh, c);
          }
          if (P2_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(SB);
      };
      constexpr int QE = P2_ORDER == 1 ? 1 : P2_ORDER == 2 ? (STEPS > 1 ? MS : 1) : 0;
#pragma unroll
      for (int q = 0; q < QE; q++) body(q);
      if constexpr (P2_ORDER != 0) early();
#pragma unroll
      for (int q = QE; q < Q; q++) body(q);
      if (pre && (STEPS % WDN) != 0) {  // the next stage's step u was requested into slot (STEPS + u) % WDN: rotate it to slot u
        u32x4 T[WDN][NT][2];
#pragma unroll
        for (int i = 0; i < WDN - 1; i++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int p = 0; p < 2; p++) T[i][nt][p] = B[(STEPS + i) % WDN][0][nt][p];
#pragma unroll
        for (int i = 0; i < WDN - 1; i++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int p = 0; p < 2; p++) B[i][0][nt][p] = T[i][nt][p];
      }
    }
  };

  // ---- prologue: first tile's first stage --------------------------------------------------------------------------
  decode(tile, tn, toy, tox);
  plan(tn, toy, tox);
  P2_MARK(0);
  load_stage(0);
  if (tid == 0) wgred[0] = wgred[1] = 0u;
  if (wave_active) {  // the ring's first slots: the first stage's first WD - 1 columns / steps
#pragma unroll
    for (int u = 0; u < WD - 1; u++) wload(u, 0, u);
  }
  store_stage(0, 0);
  __syncthreads();
  P2_MARK(1);
  P2_T0;
  int buf = 0;

  for (;;) {  // tiles
#pragma unroll
    for (int ms = 0; ms < MS; ms++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[ms][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st + 1 < nst; st++) {  // all but the tile's last stage
      auto early = [&]() { load_stage(st + 1); };
      if (wave_active) mfma_stage(std::false_type{}, buf, st, true, st + 1, early);
      else early();
      P2_ACC(0);
      store_stage(buf ^ 1, st + 1);
      __syncthreads();
      P2_ACC(1);
      buf ^= 1;
    }
    // ---- the tile's last stage: next tile's first stage and the epilogue's operands are requested before its MFMAs ----
    const int n = tn, oy0 = toy, ox0 = tox;
    const int next_tile = tile + wgx;
    const bool have_next = next_tile < tile_end;
    if (have_next) {
      decode(next_tile, tn, toy, tox);
      plan(tn, toy, tox);
    }
    P2_ACC(2);
    P2RowRegs row_in, row_r1, row_r2;
    f32x4 sc[NT], sh[NT];
    // residual granules: lanes < 32 request the h-plane granule of (pixel, 8-channel block), lanes >= 32 the l-plane one
    // (3x3 kernels with more than seven accumulator tiles per wave -- NT = 3: nine + six weight fragments per parity -- have no room for as
    // many residual granules held across the last stage: they are requested per cout sub-tile inside the epilogue instead)
    constexpr bool pre_res = EPI == 0 && !(KS == 3 && MS * NT > 7);
    u32x4 R1[pre_res ? MS : 1][pre_res ? NT : 1];
    const unsigned plane_bytes = (unsigned)(oplane * 2);
    const int yl = oy0 + ly, xl = ox0 + lx;
    unsigned vb[NT];  // byte offset of the lane's granule of sub-tile row 0 / column 0 (its plane: h for lanes < 32)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int c0 = (ns0 + nt) * 16 + cq;
      vb[nt] = c0 < a.Cout ? (unsigned)n * 2u * plane_bytes + (lane >= 32 ? plane_bytes : 0u) +
                                 (unsigned)(((c0 >> 3) * Ho + (yl << osh) + (osh ? a.oy : 0)) * Wo + (xl << osh) + (osh ? a.ox : 0)) * 16u
                           : 0x80000000u;
    }
    // pixel sub-tile ms of the wave: its first row / column inside the tile (uniform)
    auto sub_ty = [&](int ms) { return TW == 8 ? 2 * (wm * MS + ms) : ((wm * MS + ms) * 16) >> TW_LOG2; };
    auto sub_tx = [&](int ms) { return TW == 8 ? 0 : ((wm * MS + ms) * 16) & (TW - 1); };
    auto voff = [&](int nt, int ms) -> unsigned {
      if constexpr (OW > 0) return (opty[ms] >= 0 && oy0 + opty[ms] < a.Hout && vb[nt] != 0x80000000u) ? vb[nt] + (unsigned)opix[ms] : 0x80000000u;
      else return (yl + sub_ty(ms) < a.Hout && xl + sub_tx(ms) < a.Wout) ? vb[nt] : 0x80000000u;
    };
    auto soff = [&](int ms) -> int { return OW ? 0 : (((sub_ty(ms) << osh) * Wo + (sub_tx(ms) << osh)) * 16); };
    // the last stage's other loads, issued where P2_ORDER says (round 6: behind the stage's first weight request -- residual granules and
    // factor rows come from HBM / MALL and every later weight wait would wait for them too): the next tile's first patch granules (every
    // wave), then what the epilogue reads
    auto early_last = [&]() {
      if (have_next) load_stage(0);
      if (wave_active) {
        if constexpr (!INZ) p2_row_request(a.in_row, n, row_in);
        if (a.res1) p2_row_request(a.res1_row, n, row_r1);
        if (a.res2) p2_row_request(a.res2_row, n, row_r2);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const int c0 = (ns0 + nt) * 16 + cq;
          if (c0 + 3 < a.Cout) {
            sc[nt] = *reinterpret_cast<const f32x4*>(a.scale + c0);
            sh[nt] = *reinterpret_cast<const f32x4*>(a.shift + c0);
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              sc[nt][j] = c0 + j < a.Cout ? a.scale[c0 + j] : 0.f;
              sh[nt][j] = c0 + j < a.Cout ? a.shift[c0 + j] : 0.f;
            }
          }
        }
        if (pre_res && a.res1) {
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int ms = 0; ms < MS; ms++) R1[ms][nt] = __builtin_amdgcn_raw_buffer_load_b128(r1r, voff(nt, ms), soff(ms), P2_RES_AUX);
        }
        __builtin_amdgcn_sched_barrier(SB);
      }
    };
    if (wave_active) mfma_stage(std::integral_constant<bool, K48>{}, buf, nst - 1, have_next, 0, early_last);
    else early_last();
    P2_ACC(3);

    // ---- epilogue in registers: lane = (pixel lane & 15 of the sub-tile, couts cq .. cq + 3 of the sub-tile) -------------
    __builtin_amdgcn_s_setprio(P2_VALU_PRIO);  // the vector phase wins issue arbitration against the partner wave's MFMA stream
    float amax = 0.f;
    if (wave_active) {
      const float in_inv = INZ ? z_inv : __uint_as_float(row_in.inv);
      float r1_inv = 0.f, r2_inv = 0.f, out_mul = 1.f, out_inv = 1.f;
      if constexpr (EPI < 2) {
        float bound = bound_a * p2_row_amax(row_in) + bound_b;
        if (a.res1) {
          bound += p2_row_amax(row_r1);
          r1_inv = __uint_as_float(row_r1.inv);
        }
        if (a.res2) {
          bound += p2_row_amax(row_r2);
          r2_inv = __uint_as_float(row_r2.inv);
        }
        p2_scale_of(bound, out_mul, out_inv);
        if (oy0 == 0 && ox0 == 0 && by == 0 && tid == 0)
          a.out_row[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);
      }
      const float unscale = in_inv * w_unscale;
      u32x4 R2[pre_res ? MS : 1][pre_res ? NT : 1];  // a second residual (fuse layers) is requested here: its registers are the weight fragments'
      if (pre_res && a.res2) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int ms = 0; ms < MS; ms++) R2[ms][nt] = __builtin_amdgcn_raw_buffer_load_b128(r2r, voff(nt, ms), soff(ms), 0);
      }
      // residual granule as loaded (lanes < 32: [h of couts 0..3 | h of couts 4..7], lanes >= 32: [l 0..3 | l 4..7]) ->
      // this lane's four values: after the swap every lane has h in .xy and l in .zw
      auto res_of = [&](u32x4 g, float inv) -> f32x4 {
        const auto s0 = __builtin_amdgcn_permlane32_swap(g.x, g.z, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(g.y, g.w, false, false);
        const u32x2 h = {s0[0], s1[0]}, l = {s0[1], s1[1]};
        return p2_join(__builtin_bit_cast(f16x4, h), __builtin_bit_cast(f16x4, l)) * inv;
      };
      // four finished values -> scaled, split, the halves exchanged with the partner lane, ONE 16-byte store
      const float floor_ = a.relu ? 0.f : -INFINITY;  // (ReLU without a branch)
      auto put = [&](f32x4 r, unsigned vo, int so) {
        r.x = p2_max_nan(r.x, floor_); r.y = p2_max_nan(r.y, floor_); r.z = p2_max_nan(r.z, floor_); r.w = p2_max_nan(r.w, floor_);
        amax = conv_amax4(amax, r.x, r.y, r.z, r.w);
        f16x4 h, l;
        p2_split(r * out_mul, h, l);
        const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
        // lanes < 32 end with [own h | partner's h], lanes >= 32 with [partner's l | own l]
        const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
        // The sub-tile offset goes into the VECTOR offset, not soffset.  Measured on gfx950 (ROCm 7.2): after
        //     buffer_store_dwordx4 v[a:a+3], v, s[..], sN offen        (soffset in an SGPR)
        // a VALU write of v[a:a+1] in the very next instruction reaches memory in lanes 12-15 / 28-31 / ... of dword 1:
        // the store is still reading its data.  hipcc pads that hazard (s_nop 1) only when soffset is NOT a register.
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){s0[0], s1[0], s0[1], s1[1]}, orr, vo == 0x80000000u ? vo : vo + (unsigned)so, 0, 0);
        asm volatile("s_nop 1");
      };
      f32x4 scu[NT];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) scu[nt] = sc[nt] * unscale;
      // the residual / output-form cases are separate copies of the loop: tested per granule inside ONE unrolled loop
      // they were a dozen scalar branches per granule
      auto plain = [&](auto has_r1, auto has_r2) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          u32x4 q1[pre_res ? 1 : MS], q2[pre_res ? 1 : MS];
          if constexpr (!pre_res) {
#pragma unroll
            for (int ms = 0; ms < MS; ms++) {
              if constexpr (decltype(has_r1)::value) q1[ms] = __builtin_amdgcn_raw_buffer_load_b128(r1r, voff(nt, ms), soff(ms), P2_RES_AUX);
              if constexpr (decltype(has_r2)::value) q2[ms] = __builtin_amdgcn_raw_buffer_load_b128(r2r, voff(nt, ms), soff(ms), 0);
            }
          }
#pragma unroll
          for (int ms = 0; ms < MS; ms++) {
            f32x4 r = acc[ms][nt] * scu[nt] + sh[nt];
            if constexpr (decltype(has_r1)::value) {
              if constexpr (pre_res) r += res_of(R1[ms][nt], r1_inv);
              else r += res_of(q1[ms], r1_inv);
            }
            if constexpr (decltype(has_r2)::value) {
              if constexpr (pre_res) r += res_of(R2[ms][nt], r2_inv);
              else r += res_of(q2[ms], r2_inv);
            }
            put(r, voff(nt, ms), soff(ms));
#ifndef P2_NO_EPI_SB
            __builtin_amdgcn_sched_barrier(SB);  // one granule at a time: interleaving them all costs ~60 registers
#endif
          }
        }
      };
      using T_ = std::true_type;
      using F_ = std::false_type;
      if constexpr (EPI == 3) {
        // training: z = acc * 2^-s (no BatchNorm factors, no activation) as fp32 NHWC -- a lane's four couts of a pixel are 16
        // contiguous bytes, the four quarters of a 16-cout sub-tile 64 -- and, for the forward, the running (sum, sum of squares) per
        // channel; a data gradient (acc_nhwc) adds to what the slot holds: those loads are all requested before the first store
        unsigned zo[MS][NT];
        f32x4 ex[MS][NT];
        f32x4 zz[BSUM ? MS : 1][BSUM ? NT : 1];  // (BSUM) the producer's raw z at the lane's positions: same offsets as the store
        unsigned mk[BSUM ? MS : 1][BSUM ? NT : 1];  // ... and, for a producer with residuals, the ReLU mask byte of that float4
        const __amdgpu_buffer_rsrc_t bszr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BSUM ? a.bs_z : nullptr), 0,
            BSUM ? (unsigned)((int64_t)a.N * a.Hout * a.Wout * a.Cout * 4) : 0u, 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const int c0 = (ns0 + nt) * 16 + cq;
#pragma unroll
          for (int ms = 0; ms < MS; ms++) {
            int y, x;
            bool ok;
            if constexpr (OW > 0) {
              y = oy0 + opty[ms];
              x = ox0 + opix[ms];  // (round 5 fix: a map of several odd tiles per row -- 36 = 2 x 18, 24 = 2 x 12 -- lost the tile's column origin here)
              ok = opty[ms] >= 0 && y < a.Hout;
            } else {
              y = yl + sub_ty(ms);
              x = xl + sub_tx(ms);
              ok = y < a.Hout && x < a.Wout;
            }
            ok = ok && c0 < a.Cout;
            zo[ms][nt] = ok ? (unsigned)((((unsigned)n * (a.Hout << osh) + (y << osh) + (osh ? a.oy : 0)) * (a.Wout << osh) + (x << osh) + (osh ? a.ox : 0)) * a.Cout + c0) * 4u
                            : 0x80000000u;
            if (a.acc_nhwc) ex[ms][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr, zo[ms][nt], 0, 0));
            if constexpr (BSUM) {
              zz[ms][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bszr, zo[ms][nt], 0, 0));
              mk[ms][nt] = (a.bs_mask && zo[ms][nt] != 0x80000000u) ? (unsigned)a.bs_mask[zo[ms][nt] >> 4] : 0u;  // (byte of the float4 at element zo / 4)
            }
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
#pragma unroll
          for (int ms = 0; ms < MS; ms++) {
            f32x4 v = acc[ms][nt] * unscale;
            if (a.acc_nhwc) v += ex[ms][nt];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), zr, zo[ms][nt], 0, 0);
            asm volatile("s_nop 1");
            if (zo[ms][nt] != 0x80000000u) {
              if constexpr (BSUM) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                  const float zj = zz[ms][nt][j];
                  // (bwd_mask, train_ops.hip: mode 2 = BatchNorm(z) > 0, mode 3 = the kept bits)
                  const bool on = a.bs_mask ? ((mk[ms][nt] >> j) & 1u) != 0u : __builtin_fmaf(zj, bs_al[nt][j], bs_bp[nt][j]) > 0.f;
                  const float g = on ? v[j] : 0.f;
                  bsum[nt][j] += g;
                  bsq[nt][j] = __builtin_fmaf(g, (zj - bs_mu[nt][j]) * bs_is[nt][j], bsq[nt][j]);
                  bgmx[nt][j] = fmaxf(bgmx[nt][j], fabsf(g));
                }
              } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                  bsum[nt][j] += v[j];
                  bsq[nt][j] = __builtin_fmaf(v[j], v[j], bsq[nt][j]);
                }
              }
            }
          }
        }
      } else if constexpr (EPI == 2) {  // the heat-map layer: fp32 NCHW, no residuals, no upsample
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const int c0 = (ns0 + nt) * 16 + cq;
          // arg-max of the lane's pixels per cout: (value, flat index); the pixels come in increasing flat index, so a strict
          // "better" keeps the first of equal values (NaN beats everything but an earlier NaN: torch.argmax)
          float bv[4] = {0.f, 0.f, 0.f, 0.f};
          unsigned bi[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
          for (int ms = 0; ms < MS; ms++) {
            const f32x4 v = acc[ms][nt] * scu[nt] + sh[nt];
            const int y = yl + sub_ty(ms), x = xl + sub_tx(ms);
            if (y < a.Hout && x < a.Wout) {
#pragma unroll
              for (int j = 0; j < 4; j++)
                if (c0 + j < a.Cout) {
                  const float o = p2_max_nan(v[j], floor_);
                  a.out_f32[(((int64_t)n * a.Cout + c0 + j) * Ho + y) * Wo + x] = o;
                  if (bi[j] == 0xffffffffu || o > bv[j] || (o != o && bv[j] == bv[j])) {
                    bv[j] = o;
                    bi[j] = (unsigned)(y * Wo + x);
                  }
                }
            }
          }
          // decode from the epilogue (hrnet.py:344-350,500 -> utils/evaluation.py:13-30): the 16 pixel lanes of a cout
          // quarter fold their keys; the wave's key of (tile, cout) goes to its slot of the map's row
          if (a.argmax_keys) {
            const int timg = (oy0 / TH) * a.tiles_x + ox0 / TWE;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const unsigned long long kk = mval_key_row16_max(bi[j] == 0xffffffffu ? 0ull : mval_argmax_key(bv[j], bi[j]));
              if ((lane & 15) == 0 && c0 + j < a.Cout)
                mval_argmax_key_put(a.argmax_keys, n, c0 + j, a.Cout, timg * WM + wm, tiles_img * WM, kk);
            }
          }
        }
      } else if constexpr (EPI == 0) {
        if (a.res1 && a.res2) plain(T_{}, T_{});
        else if (a.res1) plain(T_{}, F_{});
        else plain(F_{}, F_{});
      } else {
        // fused nearest upsample: 2^up x 2^up replicas, each with its own residuals (uniform loops: every lane runs them)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const int c0 = (ns0 + nt) * 16 + cq;
#pragma unroll
          for (int ms = 0; ms < MS; ms++) {
            const f32x4 v = acc[ms][nt] * scu[nt] + sh[nt];
            const int y = yl + sub_ty(ms), x = xl + sub_tx(ms);
            const bool ok = y < a.Hout && x < a.Wout && c0 < a.Cout;
            const unsigned vo0 = (unsigned)n * 2u * plane_bytes + (lane >= 32 ? plane_bytes : 0u) +
                                 (unsigned)(((c0 >> 3) * Ho + (y << a.up)) * Wo + (x << a.up)) * 16u;
            // four replicas at a time (rep * rep is 4, 16 or 64): their residual granules are requested before the first
            // one is used -- one replica per round trip left these layers at half their HBM rate
            for (int i0 = 0; i0 < rep * rep; i0 += 4) {
              unsigned vo[4];
              u32x4 g1[4], g2[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int dy = (i0 + u) >> a.up, dx = (i0 + u) & (rep - 1);
                vo[u] = ok ? vo0 + (unsigned)(dy * Wo + dx) * 16u : 0x80000000u;
                if (a.res1) g1[u] = __builtin_amdgcn_raw_buffer_load_b128(r1r, vo[u], 0, P2_RES_AUX);
                if (a.res2) g2[u] = __builtin_amdgcn_raw_buffer_load_b128(r2r, vo[u], 0, 0);
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                f32x4 r = v;
                if (a.res1) r += res_of(g1[u], r1_inv);
                if (a.res2) r += res_of(g2[u], r2_inv);
                put(r, vo[u], 0);
              }
            }
          }
        }
      }
    }
    if constexpr (EPI < 2) {
      // the workgroup's max |x| without a barrier: LDS atomics, the wave that arrives last publishes and re-arms
      const unsigned amax_bits = p2_wave_umax(__float_as_uint(amax));
      if (lane == 0) {
#ifdef P2_FENCE
        __threadfence_block();
#endif
        atomicMax(&wgred[0], amax_bits);  // (a wave's LDS operations execute in order: no fence -- a fence here also
                                          // waits for every store and prefetch in flight, 2 us per tile)
        if (atomicAdd(&wgred[1], 1u) == (unsigned)(WN * WM - 1)) {
          const unsigned m = atomicExch(&wgred[0], 0u);
          wgred[1] = 0u;
          const int timg = (oy0 / TH) * a.tiles_x + ox0 / TWE;
          p2_slot_put(a.out_row + (int64_t)n * P2_ROW, a.slot_base + timg * gy + by, a.slot_total ? a.slot_total : tiles_img * gy, m);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    P2_ACC(4);
    if (!have_next) break;
    store_stage(buf ^ 1, 0);
    __syncthreads();
    P2_ACC(5);
    buf ^= 1;
    tile = next_tile;
  }
  stats_put();
  P2_FLUSH;
  P2_MARK(4);
}


template <int KS, int S, int G, int WN, int WM, int NT, int MS, int TW, bool RS, int EPI, int OW = 0, bool K48 = false, bool INZ = false, bool BSUM = false>
__global__ __launch_bounds__(64 * WN * WM) __attribute__((amdgpu_waves_per_eu(P2_WAVES(MS, NT, EPI), 8))) void conv_p2_kernel(P2Args a) {
  conv_p2_body<KS, S, G, WN, WM, NT, MS, TW, RS, EPI, OW, K48, INZ, BSUM>(a, (int)blockIdx.y, (int)gridDim.y);
}

static thread_local int g_p2_dry = 0;

template <int KS, int S, int G, int WN, int WM, int NT, int MS, int TW, bool RS, int EPI, int OW = 0, bool K48 = false, bool INZ = false, bool BSUM = false>
static int launch_p2e(P2Args a, hipStream_t s) {
  constexpr int TWE = OW ? OW : TW;
  constexpr int TH = 16 * MS * WM / TWE;
  constexpr int PH = (TH - 1) * S + KS, PW = (TWE - 1) * S + KS, PWh = (PW + 1) / 2;
  constexpr int slots = S == 1 ? PH * PW : 2 * PH * PWh;
  constexpr int PPX = (slots + 15) & ~15;
  constexpr size_t smem = (size_t)2 * 8 * G * PPX * 16 + 16 + (INZ ? 512 * 8 : 0);  // (INZ: the affine table of <= 512 input channels)
  static_assert(smem <= 160 * 1024, "LDS");
  if (INZ && (a.Cin > 512 || a.Cin % (32 * G) != 0)) return 1;
  constexpr int NTH = 64 * WN * WM;
  a.th = TH; a.tw = TWE;
  a.tiles_x = (a.Wout + TWE - 1) / TWE;
  a.tiles_y = (a.Hout + TH - 1) / TH;
  if (OW && a.Wout % OW != 0) return 1;  // (whole odd tiles per row: 18 -> 1, 36 -> 2 tiles of 18 columns)
  const unsigned groups = (unsigned)((a.NS_total + WN * NT - 1) / (WN * NT));
  a.amax_tiles = a.tiles_x * a.tiles_y;
  if (a.os && EPI < 2) {  // a parity launch: its quarter of the output rows' partial-maximum slots
    a.slot_total = 4 * a.amax_tiles * (int)groups;
    a.slot_base = (a.oy * 2 + a.ox) * a.amax_tiles * (int)groups;
  }
  a.tiles_total = a.amax_tiles * a.N;
  a.tiles_img_magic = a.amax_tiles > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)a.amax_tiles + 1) : 0u;  // (0: divide by one)
  a.tiles_x_magic = a.tiles_x > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)a.tiles_x + 1) : 0u;
  if (g_p2_dry) return 0;
#ifdef P2_STAMP
  a.dbg = g_p2_dbg;
#endif
  // persistent workgroups: as many as stay resident (the runtime's occupancy answer for this instantiation: LDS and
  // registers), a multiple of 8 per cout group so that every XCD walks its own contiguous tile range; fewer tiles than
  // that: one tile each.  (No workgroup waits for another one: an optimistic answer only costs a second round.)
  static std::atomic<int> occ{0};
  int per_cu = p2_resident_wgs(&conv_p2_kernel<KS, S, G, WN, WM, NT, MS, TW, RS, EPI, OW, K48, INZ, BSUM>, occ, smem, NTH / 64);
#ifdef P2_TUNE
  const char* pe = getenv("MVAL_P2_WGS");  // measurement builds only: workgroups per CU
  if (pe && atoi(pe) > 0) per_cu = atoi(pe);
#endif
  int wgs = (mval_cu_count() * per_cu / (int)groups) & ~7;
  if (wgs < 8) wgs = 8;
  // (a count >= 8 must be a multiple of 8: the kernel walks 8 XCD groups in steps of wgs / 8 -- with 12 workgroups for 12 tiles the
  // floor made four tiles run twice: harmless for stored outputs, wrong for the batch-statistics sums of EPI 3)
  if (wgs >= a.tiles_total) wgs = a.tiles_total < 8 ? a.tiles_total : (a.tiles_total + 7) & ~7;
  else {  // equal shares: the XCD groups' ranges are walked in steps of wgs / 8
    const int per = (a.tiles_total + 7) / 8, rounds = (per + wgs / 8 - 1) / (wgs / 8);
    wgs = 8 * ((per + rounds - 1) / rounds);
  }
  a.wgs_x = wgs;
  dim3 grid((unsigned)wgs, groups);
  if (EPI < 2 && (a.slot_total ? a.slot_total : (int64_t)a.amax_tiles * groups) > P2_SLOTS && !a.keep_rows)
    mval_launch_zero_rows(a.out_row, (int64_t)a.N * P2_ROW, s);  // (the kernel rewrites the scale slots)
  if constexpr (EPI == 3) {
    a.bn_slots = wgs * WM;
    if (a.bn_part && (int64_t)a.Cout * a.bn_slots * 2 > a.bn_part_cap) a.bn_part = nullptr;  // (no room: the caller runs the separate statistics pass)
    if (a.bn_part && a.bn_slots_host) *a.bn_slots_host = a.bn_slots;
  }
  if constexpr (BSUM) {
    a.bs_slots = wgs * WM;
    if ((int64_t)a.Cout * a.bs_slots * 3 > a.bs_cap) return 2;  // (no room for the partials: the caller launches the plain form)
    a.bs_gmax = reinterpret_cast<float*>(a.bs_part + (int64_t)a.Cout * a.bs_slots * 2);
    if (a.bs_slots_host) *a.bs_slots_host = a.bs_slots;
  }
  hipLaunchKernelGGL((conv_p2_kernel<KS, S, G, WN, WM, NT, MS, TW, RS, EPI, OW, K48, INZ, BSUM>), grid, dim3(NTH), smem, s, a);
  return 0;
}

template <int KS, int S, int G, int WN, int WM, int NT, int MS, int TW, bool RS = false, int OW = 0>
static int launch_p2(const P2Args& a, hipStream_t s) {
  if constexpr (OW > 0) {
    if (a.out_f32 || a.up) return 1;
    if (a.out_nhwc) return launch_p2e<KS, S, G, WN, WM, NT, MS, TW, RS, 3, OW>(a, s);
    return launch_p2e<KS, S, G, WN, WM, NT, MS, TW, RS, 0, OW>(a, s);
  } else {  // (else: the power-of-two forms of an odd-tile configuration are never instantiated)
    if (a.out_nhwc) {
      if (a.up || a.res1 || a.res2 || a.out_f32) return 1;
      if constexpr (MS <= 4 || KS == 3) return launch_p2e<KS, S, G, WN, WM, NT, MS, TW, RS, 3>(a, s);
      return 1;
    }
    if (a.out_f32) {
      if constexpr (KS == 1 && NT == 1 && MS <= 4) return launch_p2e<KS, S, G, WN, WM, NT, MS, TW, RS, 2>(a, s);
      return 1;
    }
    if (a.up) {
      if constexpr (KS == 1 && MS <= 4) return launch_p2e<KS, S, G, WN, WM, NT, MS, TW, RS, 1>(a, s);
      return 1;
    }
    return launch_p2e<KS, S, G, WN, WM, NT, MS, TW, RS, 0>(a, s);
  }
}

// measurement builds (-DP2_TUNE: tools/p2_sweep.py) read the tile choice from MVAL_P2_TILE="ms,nt,g" (0 = default) and the workgroups
// per CU from MVAL_P2_WGS; the product library reads no environment variable here
static void p2_override(int& ms, int& nt, int& g) {
#ifdef P2_TUNE
  const char* e = getenv("MVAL_P2_TILE");
  if (e) (void)sscanf(e, "%d,%d,%d", &ms, &nt, &g);
#endif
}

int mval_launch_conv_p2(const P2Args& a0, hipStream_t s) {
  P2Args a = a0;
  a.NS_total = (a.Cout + 15) / 16;
  if ((a.Cin & 7) || (!a.out_f32 && !a.out_nhwc && (a.Cout & 7)) || (a.out_nhwc && (a.Cout & 3))) return 1;
  if (a.out_f32 && (a.up || a.res1 || a.res2)) return 1;
  if ((int64_t)a.N * a.Hin * a.Win * a.Cin >= (int64_t)1 << 29) return 1;  // byte offsets into the planes below 2^31
  if ((int64_t)a.N * (a.Hout << a.up) * (a.Wout << a.up) * a.Cout >= (int64_t)1 << 29) return 1;
  int oms = 0, ont = 0, og = 0;
  p2_override(oms, ont, og);
  const int64_t px = (int64_t)a.N * a.Hout * a.Wout;
  if (a.bs_z) {  // (round 6) a 3x3 stride-1 data gradient that also keeps the BatchNorm backward's reduction of what it writes (P2Args::bs_z)
    int rc = 1;
    if (a.k == 3 && a.stride == 1 && a.out_nhwc && !a.up && !a.res1 && !a.res2 && !a.out_f32 && !a.os && !a.bn_part && a.bs_part &&
        a.bs_mean && a.bs_invstd && a.bs_gamma && a.bs_beta) {
      if (a.Wout >= 16 && a.Wout % 16 == 0 && a.Hout >= 4)
        rc = a.NS_total <= 2 ? launch_p2e<3, 1, 1, 2, 2, 1, 4, 16, true, 3, 0, false, false, true>(a, s)
                             : launch_p2e<3, 1, 1, 4, 1, 1, 4, 16, true, 3, 0, false, false, true>(a, s);
      else if (a.Wout == 8 && a.Hout == 8 && a.NS_total > 2 && a.Cin >= 64)
        rc = launch_p2e<3, 1, 2, 4, 1, 1, 4, 8, false, 3, 0, false, false, true>(a, s);
    }
    if (rc == 0) return 0;
    if (a.bs_slots_host) *a.bs_slots_host = 0;
    a.bs_z = nullptr;  // (shape not covered, or no room for the partials: the plain data gradient; the caller runs the reduction pass)
  }
  if (a.in_z) {  // (round 6) the training forward's 3x3 stride-1 conv that applies its producer's BatchNorm + ReLU while staging (P2Args::in_z)
    if (a.k != 3 || a.stride != 1 || !a.out_nhwc || a.up || a.res1 || a.res2 || a.out_f32 || a.acc_nhwc || a.os) return 1;
    if (!a.zin_mean || !a.zin_invstd || !a.zin_gamma || !a.zin_beta) return 1;
    if (a.Wout >= 16 && a.Wout % 16 == 0 && a.Hout >= 4) {
      if (a.NS_total <= 2) return launch_p2e<3, 1, 1, 2, 2, 1, 4, 16, true, 3, 0, false, true>(a, s);
      return launch_p2e<3, 1, 1, 4, 1, 1, 4, 16, true, 3, 0, false, true>(a, s);
    }
    if (a.Wout == 8 && a.Hout == 8 && a.NS_total > 2 && a.Cin >= 64) return launch_p2e<3, 1, 2, 4, 1, 1, 4, 8, false, 3, 0, false, true>(a, s);
    return 1;
  }
  if (a.k == 3 && a.stride == 1) {
    // (round 3 also built the v_mfma_f32_32x32x16_f16 form of this conv -- half the MFMA issues, two thirds of the LDS fragment reads per
    // FLOP: 33.1 vs 31.5 us on 64 -> 64 at 32x32, 115 vs 112.5 at 64x64: equal or slower; removed in round 5)
    // maps no power-of-two tile fits (HRNet-W48 at 384 x 288: 24 x 18 and 12 x 9): full-width odd tiles, 3 x 18 / 7 x 9 pixels
    // (round 5: THREE cout sub-tiles per wave on HRNet-W48's 96- and 192-channel branches: 96 couts = 6 sub-tiles are 2 cout waves x 3 -- with
    // one sub-tile per wave the second group of four waves ran half empty --, 192 couts = 4 x 3 in ONE group instead of three that each staged
    // the same patch; a third of the x fragment reads per MFMA.  96 -> 96 on 48 x 36 70.9 / 68.9 -> 65.9 / 64.8 us; 192 -> 192 on 24 x 18 is
    // slower on its own (68.5 -> 71.7 us) and faster inside the multi-stream forward: C4 19.79 -> 19.04 (96 only) -> 18.64 ms (both),
    // profiles/r05/w48_nt3.log.  The same on the 48-channel branch (one wave per 16 x 12 or 16 x 8 pixels x 48 couts: every wave pulls all
    // weight fragments) measured slower: 96.6 -> 107 / 102 us.)
    if (!a.up && !a.out_f32) {
      if (a.NS_total == 6 && a.Wout == 36 && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 2, 2, 3, 3, 8, false, 12>(a, s);
      if (a.NS_total == 12 && a.Wout == 18 && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 4, 1, 3, 3, 8, false, 6>(a, s);
    }
    // (round 5: TWO cout sub-tiles per wave where the cout sub-tiles come in multiples of eight -- HRNet-W32's 128- / 256-channel branches:
    // half the cout groups, so half the patch staging and barriers per MFMA.  Slower or equal launch by launch (tools/p2_sweep.py, rounds
    // 3-4) and faster inside the multi-stream forward: C2 10.12 -> 10.05 (128 ch) / 10.02 (256 ch) / 9.95 ms (both), three runs each on
    // one box, profiles/r05/w32_nt2_step.log.  Inference only: in the training step the same forms measure slower (C3 65.4 / 65.9 -> 65.6 / 66.1 with the
    // 256-channel branch, 66.1 / 66.3 with both: nt2_more.log).  Also measured flat there: two / four sub-tiles per wave on the stride-2
    // convs (C2 10.03 -> 9.98-10.02), two on HRNet-W48's 384-channel branch (92 bytes of spills; C4 17.88 -> 17.78), three on its stride-2
    // 96 -> 192 convs (17.92).)
    if (!a.up && !a.out_f32 && !a.out_nhwc && a.NS_total % 8 == 0 && !oms && !ont) {
      if (a.Wout % 16 == 0 && a.Hout >= 4) return launch_p2<3, 1, 1, 4, 1, 2, 4, 16, true>(a, s);
      if (a.Wout == 8 && a.Hout >= 4) return launch_p2<3, 1, 1, 4, 1, 2, 4, 8>(a, s);
    }
    if (a.NS_total > 2 && !a.up && !a.out_f32) {
      // (36-wide maps -- HRNet-W48's 96-channel branch -- as two 18-wide odd tiles per row: 54 of 64 slots used against 36 of 48 columns
      // of three 16-wide row-sharing tiles: 84 -> 78 us for 96 -> 96 on 48 x 36)
      // (round 5: exact tiles -- every slot a pixel, six / three sub-tiles per wave -- on HRNet-W48's 48 x 36 maps (8 x 12 pixels: 96 -> 96
      // 79 -> 68.6 us) and 24 x 18 maps (8 x 6: 192 -> 192 73 -> 66 us) where the 3 x 18 tiles below fill 54 of 64 slots; C4 20.93 -> 19.85 ms)
      if (a.Wout == 36 && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 4, 1, 1, 6, 8, false, 12>(a, s);
      if (a.Wout == 18 && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 4, 1, 1, 3, 8, false, 6>(a, s);
      if (a.Wout == 18 || a.Wout == 36) return launch_p2<3, 1, 1, 4, 1, 1, 4, 8, false, 18>(a, s);
      // (round 5: PoseResNet at 256 x 192 -- 16 x 12 / 32 x 24 maps in 8 x 12-pixel tiles (six sub-tiles per wave), 8 x 6 maps whole (three):
      // every slot a pixel, where 8-wide tiles compute 16 columns for 12 and 5 x 12 tiles 20 rows for 16)
      // (two cout sub-tiles per wave as on HRNet-W32's deep branches: PoseResNet C1 x 16 6.44 -> 6.36 ms, profiles/r05/r50_nt2.log)
      if (!a.out_nhwc && a.NS_total % 8 == 0) {
        if ((a.Wout == 12 || a.Wout == 24) && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 4, 1, 2, 6, 8, false, 12>(a, s);
        if (a.Wout == 6 && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 4, 1, 2, 3, 8, false, 6>(a, s);
      }
      if ((a.Wout == 12 || a.Wout == 24) && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 4, 1, 1, 6, 8, false, 12>(a, s);
      if (a.Wout == 6 && a.Hout % 8 == 0) return launch_p2<3, 1, 1, 4, 1, 1, 3, 8, false, 6>(a, s);
      // (9-wide odd tiles on the 72-wide maps -- 63 of 64 slots -- measured slower than the 16-wide row-sharing tiles: 105 vs 103 us)
      // (round 5: a whole 12 x 9 map per tile -- seven sub-tiles per wave, 108 of 112 slots, ONE tile per image instead of two 7 x 9 ones
      // of which the second is mostly padding: 384 -> 384 77.5 -> 61.3 us, C4 19.23 -> 18.89 ms)
      if (a.Wout == 9 && a.Hout == 12) return launch_p2<3, 1, 1, 4, 1, 1, 7, 8, false, 9>(a, s);
      if (a.Wout == 9) return launch_p2<3, 1, 1, 4, 1, 1, 4, 8, false, 9>(a, s);
    }
    // (round 5: 72-wide maps -- HRNet-W48's first branch -- in the same exact 8 x 12 tiles: 80 columns were computed for 72 by the 16-wide
    // row-sharing tiles; 48 -> 48 99.5 / 95.7 -> 97.5 / 92.9 us even without the paired K remainder, 256 -> 48 353 -> 321 us)
    if (a.Wout == 72 && a.Hout % 8 == 0 && a.NS_total > 2 && !a.up && !a.out_f32) return launch_p2<3, 1, 1, 4, 1, 1, 6, 8, false, 12>(a, s);
    if (a.Wout >= 16 && a.Hout >= 4) {
      // 48 input channels (HRNet-W48's first branch): the paired remainder stage (K48 above: 103 -> 96 us for 48 -> 48 on 96 x 72)
      if (a.Cin == 48 && a.NS_total > 2 && !a.up && !a.out_f32 && !a.out_nhwc && !oms && !ont)
        return launch_p2e<3, 1, 1, 4, 1, 1, 4, 16, true, 0, 0, true>(a, s);
      if (a.NS_total <= 2) return launch_p2<3, 1, 1, 2, 2, 1, 4, 16, true>(a, s);  // 32 couts: 2 x 2 waves, 4 rows each
      // 64-pixel tiles: measured faster than 128-pixel ones on every HRNet shape (128 -> 128 on 16x16: 28.8 vs 30.6 us,
      // 64 -> 64 on 32x32: 32.7 vs 39.5) -- more, shorter workgroups overlap their vector phases better
      const int ms = oms ? oms : 4;
      if (ms == 8 && a.Hout >= 8) return launch_p2<3, 1, 1, 4, 1, 1, 8, 16, true>(a, s);
      if (ont == 2) return launch_p2<3, 1, 1, 4, 1, 2, 4, 16, true>(a, s);
      return launch_p2<3, 1, 1, 4, 1, 1, 4, 16, true>(a, s);
    }
    // (round 5: also maps narrower than a tile -- PoseResNet's 8 x 6 layer4 at 256 x 192 inputs: one 8 x 8 tile per image, a quarter of its
    // columns outside the map; staging zero-fills them, the stores mask them)
    if (a.Wout >= 4 && a.Hout >= 4) {
      if (a.NS_total <= 2) return launch_p2<3, 1, 1, 2, 2, 1, 2, 8>(a, s);
      if (ont == 2) return launch_p2<3, 1, 1, 4, 1, 2, 4, 8>(a, s);
      return launch_p2<3, 1, 1, 4, 1, 1, 4, 8>(a, s);
    }
    return 1;
  }
  if (a.k == 3 && a.stride == 2) {
    if (a.Wout < 4 || a.Hout < 4) return 1;
    const int ms = oms ? oms : 2;
    // (round 5: exact 4 x 12 / 8 x 6-pixel tiles on output maps 36 / 12 resp. 18 / 6 columns wide -- HRNet-W48 at 384 x 288, PoseResNet at
    // 256 x 192 --, where 8-wide tiles compute 40 columns for 36 and 24 for 18: 48 -> 96 89 -> 83 us, 96 -> 192 68.5 -> 58, C4 20.77 -> 20.40 ms)
    if (a.NS_total > 2 && a.Wout % 12 == 0 && (a.Wout & 7) != 0 && a.Hout % 4 == 0) return launch_p2<3, 2, 1, 4, 1, 1, 3, 8, false, 12>(a, s);
    if (a.NS_total > 2 && a.Wout % 6 == 0 && a.Wout % 12 != 0 && a.Hout % 8 == 0) return launch_p2<3, 2, 1, 4, 1, 1, 3, 8, false, 6>(a, s);
    if (a.NS_total <= 2) return launch_p2<3, 2, 1, 2, 2, 1, 1, 8>(a, s);  // (8 x 8 tiles instead of 4 x 8: no change, 36.8 vs 36.2 us)
    // (16-wide tiles measured the same or slower: 2 x 16 px 34.0 / 24.8 / 22.2 us on 32 -> 64 / 64 -> 128 / 128 -> 256 against
    // 34.1 / 24.8 / 22.1 for 4 x 8; 4 x 16 px 39.7 / 29.3 / 22.3)
    if (ms == 4) return launch_p2<3, 2, 1, 4, 1, 1, 4, 8>(a, s);
    return launch_p2<3, 2, 1, 4, 1, 1, 2, 8>(a, s);
  }
  if (a.k == 2 && a.stride == 1) {
    // one parity of a transposed conv / of a stride-2 data gradient (P2Args.pad_y ...): plain 64-pixel tiles, 8 or 16 wide
    if (a.up || a.out_f32 || a.res1 || a.res2 || a.Hout != a.Hin || a.Wout != a.Win || (unsigned)a.pad_y > 1u || (unsigned)a.pad_x > 1u ||
        (unsigned)a.os > 1u)
      return 1;
    if (a.Wout < 4 || a.Hout < 4) return 1;
    // (four taps per 32-channel chunk: two chunks per stage -- twice the MFMA work per barrier -- from 64 input channels on)
    // (12-wide maps: 8 x 12-pixel tiles, one chunk per stage -- with two the six sub-tiles spill; 24-wide maps: three 8-wide tiles per row)
    // (the scattered output's byte offsets are 32-bit: every parity form is refused above 2^29 elements, the 12- and 6-wide ones included)
    if ((((int64_t)a.N * a.Hout * a.Wout * a.Cout) << (2 * a.os)) >= (int64_t)1 << 29) return 1;
    if (a.Wout == 12 && a.Hout % 8 == 0 && a.NS_total > 2) return launch_p2<2, 1, 1, 4, 1, 1, 6, 8, false, 12>(a, s);
    if (a.Wout == 6 && a.Hout % 8 == 0 && a.NS_total > 2)
      return a.Cin >= 64 ? launch_p2<2, 1, 2, 4, 1, 1, 3, 8, false, 6>(a, s) : launch_p2<2, 1, 1, 4, 1, 1, 3, 8, false, 6>(a, s);
    if (a.Wout >= 16 && a.Wout % 16 == 0) {
      if (a.NS_total <= 2) return launch_p2<2, 1, 1, 2, 2, 1, 2, 16>(a, s);
      return launch_p2<2, 1, 1, 4, 1, 1, 4, 16>(a, s);
    }
    if (a.NS_total <= 2) return launch_p2<2, 1, 1, 2, 2, 1, 2, 8>(a, s);
    return a.Cin >= 64 ? launch_p2<2, 1, 2, 4, 1, 1, 4, 8>(a, s) : launch_p2<2, 1, 1, 4, 1, 1, 4, 8>(a, s);
  }
  if (a.k == 1 && a.stride == 2 && !a.in_sub) {  // (pose_resnet.py's downsample branches) a stride-1 1x1 conv over every second input pixel
    a.in_sub = 1;
    a.stride = 1;
  }
  if (a.k == 1 && a.stride == 1) {
    if (a.Wout < 4 || a.Hout * a.Wout < 32) return 1;
    // (round 5) a 1 x 1 stride-1 conv has no halo and its planes are flat in the pixel index: maps whose width fills no power-of-two tile
    // (PoseResNet's 48 / 24 / 12 / 6 columns, HRNet-W48's 72 / 36 / 18 / 9) run as ONE row of H * W pixels in 64-pixel tiles
    if (!a.up && !a.in_sub && !oms && (a.Wout & 63) != 0 && a.Hout > 1 && a.Hout * a.Wout >= 64 && a.Hin == a.Hout && a.Win == a.Wout) {
      a.Wout = a.Win = a.Hout * a.Wout;
      a.Hout = a.Hin = 1;
    }
    const int g = og ? og : 2;
    const int ms = oms ? oms : 4;
    const int nt = ont ? ont : ((a.NS_total % 8 == 0 && ((px + 63) / 64) * (a.NS_total / 8) >= 1024) ? 2 : 1);
#define P2_1X1(TW_)                                                                                              \
  do {                                                                                                           \
    if (a.NS_total <= 2) return g == 1 ? launch_p2<1, 1, 1, 2, 2, 1, 2, TW_>(a, s) : launch_p2<1, 1, 2, 2, 2, 1, 2, TW_>(a, s); \
    if (ms == 8 && a.Hout * a.Wout >= 128)                                                                       \
      return nt == 2 ? launch_p2<1, 1, 2, 4, 1, 2, 8, TW_>(a, s) : launch_p2<1, 1, 2, 4, 1, 1, 8, TW_>(a, s);     \
    if (nt == 2) return g == 4 ? launch_p2<1, 1, 4, 4, 1, 2, 4, TW_>(a, s) : launch_p2<1, 1, 2, 4, 1, 2, 4, TW_>(a, s); \
    return g == 4 ? launch_p2<1, 1, 4, 4, 1, 1, 4, TW_>(a, s) : g == 1 ? launch_p2<1, 1, 1, 4, 1, 1, 4, TW_>(a, s) : launch_p2<1, 1, 2, 4, 1, 1, 4, TW_>(a, s); \
  } while (0)
    // tile width: the widest power of two among those that compute the fewest padded columns (36-wide maps: five 8-wide tiles = 40
    // columns, not two 32-wide ones = 64; round 5)
    int tw = 8, best = ((a.Wout + 7) / 8) * 8;
    for (int c = 16; c <= 64; c *= 2)
      if (a.Wout >= c && ((a.Wout + c - 1) / c) * c <= best) {
        tw = c;
        best = ((a.Wout + c - 1) / c) * c;
      }
    if (tw == 64) P2_1X1(64);
    if (tw == 32) P2_1X1(32);
    if (tw == 16) P2_1X1(16);
    P2_1X1(8);
#undef P2_1X1
  }
  return 1;
}

int mval_conv_p2_parity_supported(int cin, int cout, int h, int w, int n, int nhwc_out) {
  P2Args a = {};
  a.k = 2; a.stride = 1; a.os = 1;
  a.Cin = cin; a.Cout = cout; a.Hin = a.Hout = h; a.Win = a.Wout = w; a.N = n;
  a.out_nhwc = nhwc_out ? reinterpret_cast<float*>(1) : nullptr;
  g_p2_dry = 1;
  const int rc = mval_launch_conv_p2(a, nullptr);
  g_p2_dry = 0;
  return rc == 0;
}

int mval_conv_p2_supported(int k, int stride, int cin, int cout, int hin, int win, int up, int out_nchw, int n) {
  P2Args a = {};
  a.k = k; a.stride = stride;
  a.Cin = cin; a.Cout = cout; a.Hin = hin; a.Win = win; a.N = n;
  const int pad = k / 2;
  a.Hout = (hin + 2 * pad - k) / stride + 1;
  a.Wout = (win + 2 * pad - k) / stride + 1;
  a.up = up;
  a.out_f32 = out_nchw ? reinterpret_cast<float*>(1) : nullptr;
  g_p2_dry = 1;
  const int rc = mval_launch_conv_p2(a, nullptr);
  g_p2_dry = 0;
  return rc == 0;
}

int mval_conv_p2_bsum_supported(int cin, int cout, int h, int w, int n) {
  // (the same three tile forms as the in_z conv; room for the partials is decided at launch)
  const int ns = (cout + 15) / 16;
  if ((cin & 7) || (cout & 3) || (int64_t)n * h * w * cin >= ((int64_t)1 << 29) || (int64_t)n * h * w * cout >= ((int64_t)1 << 29)) return 0;
  if (w >= 16 && w % 16 == 0 && h >= 4) return 1;
  return w == 8 && h == 8 && ns > 2 && cin >= 64;
}

int mval_conv_p2_inz_supported(int cin, int cout, int h, int w, int n) {
  P2Args a = {};
  a.k = 3; a.stride = 1;
  a.Cin = cin; a.Cout = cout; a.Hin = a.Hout = h; a.Win = a.Wout = w; a.N = n;
  a.out_nhwc = reinterpret_cast<float*>(1);
  a.in_z = a.zin_mean = a.zin_invstd = a.zin_gamma = a.zin_beta = reinterpret_cast<const float*>(1);
  g_p2_dry = 1;
  const int rc = mval_launch_conv_p2(a, nullptr);
  g_p2_dry = 0;
  return rc == 0;
}

// ---- format conversion (network boundaries, tests) -----------------------------------------------------------------
// fp32 NHWC [n][H][W][C] with its max |x| rows (kept by the producer) -> P2 planes; the image's exact maximum is known
// here, so the scale puts IT in [2^13, 2^14).  One thread per (pixel, 8-channel block).
__global__ __launch_bounds__(256) void nhwc_to_p2_kernel(const float* __restrict__ x, _Float16* __restrict__ out, const unsigned* rows_in,
                                                       unsigned* rows_out, int HW, int C8) {
  const int n = blockIdx.y;
  const unsigned amax = conv_amax_read(rows_in + (int64_t)n * MVAL_AMAX_ROW);
  float mul, inv;
  p2_scale_of(__uint_as_float(amax), mul, inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned* row = rows_out + (int64_t)n * P2_ROW;
    row[0] = amax;  // (the other partial slots stay zero)
    row[P2_INV_SLOT] = __float_as_uint(inv);
  }
  const int64_t total = (int64_t)HW * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i / HW), p = (int)(i - (int64_t)c8 * HW);  // pixel-fastest: 16-byte stores coalesce
    const float* src = x + ((int64_t)n * HW + p) * (C8 * 8) + c8 * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    f16x4 h0, l0, h1, l1;
    p2_split(v0 * mul, h0, l0);
    p2_split(v1 * mul, h1, l1);
    _Float16* d = out + (((int64_t)n * 2 * C8 + c8) * HW + p) * 8;
    *reinterpret_cast<f16x8*>(d) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<f16x8*>(d + (int64_t)C8 * HW * 8) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

__global__ __launch_bounds__(256) void p2_to_nhwc_kernel(const _Float16* __restrict__ x, const unsigned* rows, float* __restrict__ out,
                                                       int HW, int C8) {
  const int n = blockIdx.y;
  const float inv = __uint_as_float(rows[(int64_t)n * P2_ROW + P2_INV_SLOT]);
  const int64_t total = (int64_t)HW * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i / HW), p = (int)(i - (int64_t)c8 * HW);
    const _Float16* s = x + (((int64_t)n * 2 * C8 + c8) * HW + p) * 8;
    const f16x8 h = *reinterpret_cast<const f16x8*>(s), l = *reinterpret_cast<const f16x8*>(s + (int64_t)C8 * HW * 8);
    float* d = out + ((int64_t)n * HW + p) * (C8 * 8) + c8 * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = ((float)h[j] + (float)l[j]) * inv;
  }
}

int mval_launch_nhwc_to_p2(const float* x, const unsigned* rows_in, _Float16* planes, unsigned* rows, int n_images, int HW, int C,
                            hipStream_t s) {
  int64_t blocks = ((int64_t)HW * (C / 8) + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(nhwc_to_p2_kernel, dim3((unsigned)blocks, (unsigned)n_images), dim3(256), 0, s, x, planes, rows_in, rows, HW, C / 8);
  return 0;
}

extern "C" int mval_nhwc_to_p2(const float* x, const uint32_t* rows_in, void* planes, uint32_t* rows, int n_images, int H, int W, int C,
                               void* stream) {
  MVAL_REQUIRE(x && planes && rows && rows_in && n_images > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "mval_nhwc_to_p2: bad arguments");
  mval_launch_nhwc_to_p2(x, rows_in, reinterpret_cast<_Float16*>(planes), rows, n_images, H * W, C, mval_stream(stream));
  MVAL_CHECK_LAUNCH("mval_nhwc_to_p2");
  return 0;
}

extern "C" int mval_p2_to_nhwc(const void* planes, const uint32_t* rows, float* out, int n_images, int H, int W, int C, void* stream) {
  MVAL_REQUIRE(out && planes && rows && n_images > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "mval_p2_to_nhwc: bad arguments");
  int64_t blocks = ((int64_t)H * W * (C / 8) + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(p2_to_nhwc_kernel, dim3((unsigned)blocks, (unsigned)n_images), dim3(256), 0, mval_stream(stream),
                     reinterpret_cast<const _Float16*>(planes), rows, out, H * W, C / 8);
  MVAL_CHECK_LAUNCH("mval_p2_to_nhwc");
  return 0;
}
