// Weight gradient of the 3x3 (stride 1 / 2) and wide 1x1 convs on the 16-bit matrix cores with the splits of
// conv_mfma_split.hip: PL = 3, the exact three-way bf16 split of both operands (six products, 2.67x the MFMA rate of
// the exact-fp32 kernel in conv_wgrad.hip), or PL = 2, the scaled two-way fp16 split (three products, 5.3x) when the
// magnitude rows of x and dz are at hand (training with MVAL_CONV=h2: x's row from its producer's BatchNorm apply,
// dz's from this op's BatchNorm backward); same results to fp32 rounding.
//
//   dW[tap][ci][co] = sum over pixels p of  x[p + tap][ci] * dz[p][co]
//
// is a GEMM whose K index is the PIXEL, while both operands are stored pixel-major (NHWC): the MFMA
// wants, per lane, 8 consecutive k of one channel.  gfx950's transposed LDS read delivers exactly
// that from a [pixel][channel] image (ds_read_b64_tr_b16: a 16-lane group reads 4 pixel rows x 16
// channels and each lane receives one channel's 4 pixels), so the tiles are staged like the forward
// kernel's -- three bf16 planes [plane][pixel][channels + pad] -- and both the x fragment (shifted
// by the tap: a constant row offset) and the dz fragment are two transposed reads per plane.
//   * tile = 4 x 16 (or 8 x 8) output pixels; a k-step is 32 of them, k = 8 g + j  <->  pixel
//     4 g + (j & 3) in tile row 2 s + (j >> 2)  (8-wide tiles: row 4 s + 2 (j >> 2) + (g >> 1),
//     column 4 (g & 1) + (j & 3)): the two 16-lane groups of a 32-lane half read pixel blocks four
//     rows of the image apart, which is conflict-free for 96-byte and 160-byte rows;
//   * a workgroup owns 32 cin x (32 or 64) cout x 9 taps: wave = (cin tile, cout tile pair), nine
//     taps x NT accumulator tiles in registers; per k-step and wave 54 + 6 NT transposed reads feed
//     54 NT MFMAs (1x1 convs: 64 cin x 64 cout per workgroup, 2 x 2 tiles per wave; stride 2: 32-pixel
//     tiles, fragment pixels two patch rows apart -- 2-way bank conflicts, accepted);
//   * split-K over workgroups, per-split slabs and the float64 slab reduction exactly as in
//     conv_wgrad.hip (same slab layout, same reduce kernel); the next tile's operands travel
//     global -> registers during the MFMA loop.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

#ifndef WB_PF
// register sets of staged operands: 1 = the next tile requested during this tile's MFMA phase; 2 = two tiles ahead (round 6 measurement knob:
// SLOWER -- weight-gradient family 15.29 -> 15.76 ms per C3 step, 256 -> 256 @8x8 48.3 -> 56.0 us, step 58.8 -> 58.5-59.3 ms
// (profiles/r06/wgrad_prefetch2_ab.log): three resident workgroups per CU already cover one another's load latency, and the second set costs
// 30-60 registers)
#define WB_PF 1
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

struct WgradBf3Args {
  const float* x;   // NHWC (N, Hin, Win, Cin)
  const float* dz;  // NHWC (N, H, W, Cout): the conv's output resolution
  float* slabs;     // [PS][k*k][Cin][Cout]
  int N, Hin, Win, H, W, Cin, Cout;
  int tiles_x, tiles_y, ntiles, PS;
  const unsigned* x_amax;   // PL = 2: magnitude rows ([count, partials], one row per tensor) of x and dz
  const unsigned* dz_amax;
  // XP2 (round 4): x as P2 planes [n][plane h,l][Cin/8][Hin][Win][8] with its rows (csrc/conv_p2.h; ONE scale for the tensor in
  // training: row 0's 2^-s) -- the activation the training forward's P2 convs read; staging is a 16-byte copy per (pixel, 8 channels,
  // plane) instead of a float4 load + scale + split
  const _Float16* x_p2;
  const unsigned* x_p2_rows;
  // ZP2: dz as P2 planes [n][plane][Cout/8][H][W][8] + rows (written by the BatchNorm backward for the P2 data-gradient conv)
  const _Float16* dz_p2;
  const unsigned* dz_p2_rows;
  // XZ (round 6): x does not exist as a tensor -- it is relu(BatchNorm(x)) of the producer's raw conv output (a.x: fp32 NHWC z) with the
  // producer's batch statistics and affine parameters; the staging applies it (conv_p2.h P2Args::in_z: same arithmetic, same scale, same
  // split, so the LDS image equals the one staged from the producer's planes bit for bit)
  const float* xz_mean; const float* xz_invstd; const float* xz_gamma; const float* xz_beta;
  float xz_sqrt_m1;
};

__device__ __forceinline__ void wb_split_pair(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
  const f32x2 r1 = x - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r1 - mf, bf16x2));
}

// x = h + m + l exactly (round-to-nearest bf16 of the running residual); 8-byte plane entries
__device__ __forceinline__ void wb_split_store(const f32x4 v, char* dst, int plane_bytes) {
  unsigned h0, m0, l0, h1, m1, l1;
  wb_split_pair((f32x2){v[0], v[1]}, h0, m0, l0);
  wb_split_pair((f32x2){v[2], v[3]}, h1, m1, l1);
  *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
  *reinterpret_cast<u32x2*>(dst + plane_bytes) = (u32x2){m0, m1};
  *reinterpret_cast<u32x2*>(dst + 2 * plane_bytes) = (u32x2){l0, l1};
}

// scaled two-way fp16 split: v * mul = h + l (+ 2^-22 relative); 8-byte plane entries
__device__ __forceinline__ void wb_split_store2(const f32x4 v, float mul, char* dst, int plane_bytes) {
  const f32x4 vs = v * mul;
  const f16x4 h = __builtin_convertvector(vs, f16x4);
  const f16x4 l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x4), f16x4);
  *reinterpret_cast<f16x4*>(dst) = h;
  *reinterpret_cast<f16x4*>(dst + plane_bytes) = l;
}

// 16 channels x 8 k of one plane: two transposed reads (k 0..3 and 4..7 of this lane's k-group)
__device__ __forceinline__ s16x8 wb_frag(const char* lo, const char* hi) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(hi));
  return (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
template <int PL>
__device__ __forceinline__ f32x4 wb_mfma(const s16x8 a, const s16x8 b, const f32x4 c) {
  if constexpr (PL == 3)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// power of two that puts a maximum with these float bits into [2^14, 2^15), and its inverse
__device__ __forceinline__ void wb_scale(unsigned amax_bits, float& mul, float& inv) {
  const int e = (int)((amax_bits >> 23) & 0xff);
  int s = (e == 0 || e == 255) ? 0 : 14 - (e - 127);
  s = max(-110, min(110, s));
  mul = __uint_as_float((unsigned)(127 + s) << 23);
  inv = __uint_as_float((unsigned)(127 - s) << 23);
}

// KS x KS taps, stride S, tile TH x TW output pixels, NT cout tiles and MI cin tiles (of 16) per wave
typedef unsigned wb_u32x4 __attribute__((ext_vector_type(4)));
template <int PL, int KS, int S, int TH, int TW, int NT, int MI, bool XP2 = false, bool ZP2 = false, bool XZ = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((KS == 3 && S == 1 && TH == 8 && NT == 2) ? 2 : 1, 8))) void conv_wgrad_bf3_kernel(WgradBf3Args a) {
  static_assert(!(XP2 || ZP2) || PL == 2, "P2 activations are fp16 pairs");
  static_assert(!XZ || (PL == 2 && !XP2 && ZP2), "XZ: x from the producer's z, dz from planes");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int T = KS * KS, MT = TH * TW;
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, PPX = PH * PW;
  constexpr int CI = 32 * MI, CO = 32 * NT;   // channels per workgroup
  constexpr int XROW = CI * 2 + 32, ZROW = CO * 2 + 32;  // bytes per pixel and plane (96 / 160)
  constexpr int XPLANE = PPX * XROW, ZPLANE = MT * ZROW;
  constexpr int XQ = CI / 4, ZQ = CO / 4;     // float4 per pixel
  constexpr int NEX = (PPX * XQ + 255) / 256, NEZ = (MT * ZQ + 255) / 256;
  char* xl = smem;                 // [PL][PPX][XROW]
  char* zl = smem + PL * XPLANE;   // [PL][MT][ZROW]
  float x_mul = 1.f, z_mul = 1.f, unscale = 1.f;
  if constexpr (PL == 2) {
    float xi, zi;
    if constexpr (XP2) xi = __uint_as_float(a.x_p2_rows[511]);  // (P2_INV_SLOT of row 0: one scale for the whole tensor in training)
    else if constexpr (XZ) {
      // the scale the producer's apply pass gives its planes: Samuelson's bound over ALL input channels into [2^13, 2^14) (conv_p2.h p2_scale_of)
      float bnd = 0.f;
      for (int c = threadIdx.x & 63; c < a.Cin; c += 64) bnd = fmaxf(bnd, __builtin_fmaf(fabsf(a.xz_gamma[c]), a.xz_sqrt_m1, fabsf(a.xz_beta[c])));
      for (int o = 32; o > 0; o >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, o));
      bnd *= (1.f + 1e-6f);
      const int e = (int)((__float_as_uint(bnd) >> 23) & 0xff);
      int sft = (e == 0 || e == 255) ? 0 : 13 - (e - 127);
      sft = max(-110, min(110, sft));
      x_mul = __uint_as_float((unsigned)(127 + sft) << 23);
      xi = __uint_as_float((unsigned)(127 - sft) << 23);
    }
    else wb_scale(conv_amax_read(a.x_amax), x_mul, xi);
    if constexpr (ZP2) zi = __uint_as_float(a.dz_p2_rows[511]);
    else wb_scale(conv_amax_read(a.dz_amax), z_mul, zi);
    unscale = xi * zi;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci_g = wave & 1, co_g = wave >> 1;  // cin group (16 * MI) / cout group (16 * NT) of this wave
  const int ci0 = blockIdx.y * CI, co0 = blockIdx.z * CO;
  const int pad = KS / 2;

  f32x4 acc[T][MI][NT];
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[t][mi][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed-read lane roles: group g = k-group, q = which of the block's 4 pixels this lane
  // addresses, p = which 4-channel quarter of the 16 channels
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int px_lo, py_lo, py_step;  // tile pixel of (g, q) for read 0; read 1 is py_step rows below
  if (TW == 16) {
    px_lo = 4 * g + q; py_lo = 0; py_step = 1;
  } else {
    px_lo = 4 * (g & 1) + q; py_lo = g >> 1; py_step = 2;
  }
  constexpr int KROWS = (TW == 16) ? 2 : 4;  // tile rows per k-step
  static_assert(TH % KROWS == 0, "tile height must hold whole k-steps");
  const int xoff = (py_lo * S * PW + px_lo * S) * XROW + ci_g * (32 * MI) + p * 8;
  const int zoff = (py_lo * TW + px_lo) * ZROW + co_g * (32 * NT) + p * 8;

  // staging slots (tile-invariant)
  int xrow[NEX], xcol[NEX];
#pragma unroll
  for (int i = 0; i < NEX; i++) {
    const int px = (tid + 256 * i) / XQ;
    xrow[i] = px / PW;
    xcol[i] = px - xrow[i] * PW;
  }
  const int xq4 = (tid % XQ) * 4, zq4 = (tid % ZQ) * 4;  // 256 % XQ == 0 == 256 % ZQ
  const bool cx_ok = ci0 + xq4 < a.Cin, cz_ok = co0 + zq4 < a.Cout;

  // (WB_PF 2, measurement: TWO register sets -- the operands of tile t + 2 PS requested while tile t is multiplied, so a load has a whole
  // iteration before its LDS store waits for it; measured slower, see WB_PF above)
  f32x4 xr[WB_PF][NEX], zr[WB_PF][NEZ];
  // (XZ) the thread's four input channels' factors, pre-multiplied by 2^s (exact), and which of its staged pixels lie inside the image
  // (the conv pads the ACTIVATION with zeros, not z)
  f32x4 xz_a = (f32x4){0.f, 0.f, 0.f, 0.f}, xz_b = xz_a;
  unsigned xz_ok[WB_PF] = {};
  if constexpr (XZ) {
    if (cx_ok) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c = ci0 + xq4 + k;
        const float alpha = a.xz_invstd[c] * a.xz_gamma[c];
        xz_a[k] = alpha * x_mul;
        xz_b[k] = __builtin_fmaf(-a.xz_mean[c], alpha, a.xz_beta[c]) * x_mul;
      }
    }
  }
  // XP2: item j = tid % XQ of a pixel -> plane j / (CI / 8), 8-channel block j % (CI / 8)  (XQ = CI / 4 = 2 planes x CI / 8 blocks)
  const int xp_plane = (tid % XQ) / (CI / 8), xp_c8 = (tid % XQ) % (CI / 8);
  const int zp_plane = (tid % ZQ) / (CO / 8), zp_c8 = (tid % ZQ) % (CO / 8);
  const int C8x = a.Cin >> 3;
  const int64_t xp_hw = (int64_t)a.Hin * a.Win;
  auto load_tile = [&](int t, auto set_) {
    constexpr int B = decltype(set_)::value;
    const int txi = t % a.tiles_x;
    t /= a.tiles_x;
    const int oy0 = (t % a.tiles_y) * TH, ox0 = txi * TW, n = t / a.tiles_y;
#pragma unroll
    for (int i = 0; i < NEX; i++) {
      const int iy = oy0 * S - pad + xrow[i], ix = ox0 * S - pad + xcol[i];
      if constexpr (XP2) {
        const bool ok = tid + 256 * i < PPX * XQ && ci0 + xp_c8 * 8 < a.Cin && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        const wb_u32x4 g = ok ? *reinterpret_cast<const wb_u32x4*>(a.x_p2 + ((((int64_t)n * 2 + xp_plane) * C8x + (ci0 >> 3) + xp_c8) * xp_hw + (int64_t)iy * a.Win + ix) * 8)
                              : (wb_u32x4){0u, 0u, 0u, 0u};
        xr[B][i] = __builtin_bit_cast(f32x4, g);
      } else {
      const bool ok = tid + 256 * i < PPX * XQ && cx_ok && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      xr[B][i] = ok ? *reinterpret_cast<const f32x4*>(a.x + (((int64_t)n * a.Hin + iy) * a.Win + ix) * a.Cin + ci0 + xq4)
                    : (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (XZ) xz_ok[B] = (xz_ok[B] & ~(1u << i)) | ((unsigned)ok << i);
      }
    }
#pragma unroll
    for (int i = 0; i < NEZ; i++) {
      const int pz = (tid + 256 * i) / ZQ;
      const int y = oy0 + pz / TW, x = ox0 + pz % TW;
      if constexpr (ZP2) {
        const bool ok = tid + 256 * i < MT * ZQ && co0 + zp_c8 * 8 < a.Cout && y < a.H && x < a.W;
        const wb_u32x4 g = ok ? *reinterpret_cast<const wb_u32x4*>(a.dz_p2 + ((((int64_t)n * 2 + zp_plane) * (a.Cout >> 3) + (co0 >> 3) + zp_c8) * ((int64_t)a.H * a.W) + (int64_t)y * a.W + x) * 8)
                              : (wb_u32x4){0u, 0u, 0u, 0u};
        zr[B][i] = __builtin_bit_cast(f32x4, g);
      } else {
      zr[B][i] = (tid + 256 * i < MT * ZQ && cz_ok && y < a.H && x < a.W)
                  ? *reinterpret_cast<const f32x4*>(a.dz + (((int64_t)n * a.H + y) * a.W + x) * a.Cout + co0 + zq4)
                  : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  using I0_ = std::integral_constant<int, 0>;
  using I1_ = std::integral_constant<int, WB_PF - 1>;
  int tile = blockIdx.x;
  if (tile < a.ntiles) load_tile(tile, I0_{});
  if (WB_PF > 1 && tile + a.PS < a.ntiles) load_tile(tile + a.PS, I1_{});
  for (; tile < a.ntiles; tile += a.PS) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NEX; i++) {
      const int e = tid + 256 * i;
      if (e < PPX * XQ) {
        if constexpr (XP2) *reinterpret_cast<wb_u32x4*>(xl + xp_plane * XPLANE + (e / XQ) * XROW + xp_c8 * 16) = __builtin_bit_cast(wb_u32x4, xr[0][i]);
        else if constexpr (XZ) {
          f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
          if ((xz_ok[0] >> i) & 1u) {
#pragma unroll
            for (int k = 0; k < 4; k++) y[k] = __builtin_elementwise_maximum(__builtin_fmaf(xr[0][i][k], xz_a[k], xz_b[k]), 0.f);
          }
          wb_split_store2(y, 1.f, xl + (e / XQ) * XROW + (e % XQ) * 8, XPLANE);
        }
        else if constexpr (PL == 3) wb_split_store(xr[0][i], xl + (e / XQ) * XROW + (e % XQ) * 8, XPLANE);
        else wb_split_store2(xr[0][i], x_mul, xl + (e / XQ) * XROW + (e % XQ) * 8, XPLANE);
      }
    }
#pragma unroll
    for (int i = 0; i < NEZ; i++) {
      const int e = tid + 256 * i;
      if (e < MT * ZQ) {
        if constexpr (ZP2) *reinterpret_cast<wb_u32x4*>(zl + zp_plane * ZPLANE + (e / ZQ) * ZROW + zp_c8 * 16) = __builtin_bit_cast(wb_u32x4, zr[0][i]);
        else if constexpr (PL == 3) wb_split_store(zr[0][i], zl + (e / ZQ) * ZROW + (e % ZQ) * 8, ZPLANE);
        else wb_split_store2(zr[0][i], z_mul, zl + (e / ZQ) * ZROW + (e % ZQ) * 8, ZPLANE);
      }
    }
    __syncthreads();
    if constexpr (WB_PF > 1) {
      // set 1 (requested one whole iteration ago) becomes set 0; the tile two strides ahead is requested into set 1.  (One loop body:
      // alternating the sets in two copies of the body doubled its registers.)
#pragma unroll
      for (int i = 0; i < NEX; i++) xr[0][i] = xr[WB_PF - 1][i];
#pragma unroll
      for (int i = 0; i < NEZ; i++) zr[0][i] = zr[WB_PF - 1][i];
      xz_ok[0] = xz_ok[WB_PF - 1];
      if (tile + 2 * a.PS < a.ntiles) load_tile(tile + 2 * a.PS, I1_{});
    } else {
      if (tile + a.PS < a.ntiles) load_tile(tile + a.PS, I0_{});
    }
#pragma unroll
    for (int ks = 0; ks < TH / KROWS; ks++) {
      // dz fragments of this k-step: [cout tile][plane]
      s16x8 zf[NT][PL];
      const char* zb = zl + zoff + ks * KROWS * TW * ZROW;
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int pl = 0; pl < PL; pl++)
          zf[nt][pl] = wb_frag(zb + pl * ZPLANE + nt * 32, zb + pl * ZPLANE + nt * 32 + py_step * TW * ZROW);
      const char* xb = xl + xoff + ks * KROWS * S * PW * XROW;
#pragma unroll
      for (int t = 0; t < T; t++) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
          const char* xt = xb + ((t / KS) * PW + t % KS) * XROW + mi * 32;
          s16x8 xf[PL];
#pragma unroll
          for (int pl = 0; pl < PL; pl++) xf[pl] = wb_frag(xt + pl * XPLANE, xt + pl * XPLANE + py_step * S * PW * XROW);
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            f32x4 c = acc[t][mi][nt];  // small terms first
            if constexpr (PL == 3) {
              c = wb_mfma<3>(xf[2], zf[nt][0], c);
              c = wb_mfma<3>(xf[0], zf[nt][2], c);
              c = wb_mfma<3>(xf[1], zf[nt][1], c);
              c = wb_mfma<3>(xf[1], zf[nt][0], c);
              c = wb_mfma<3>(xf[0], zf[nt][1], c);
              c = wb_mfma<3>(xf[0], zf[nt][0], c);
            } else {
              c = wb_mfma<2>(xf[1], zf[nt][0], c);
              c = wb_mfma<2>(xf[0], zf[nt][1], c);
              c = wb_mfma<2>(xf[0], zf[nt][0], c);
            }
            acc[t][mi][nt] = c;
          }
        }
      }
    }
  }
  // slab[ps][t][ci][co]; C layout: row (cin) = (lane >> 4) * 4 + r, col (cout) = lane & 15
#ifdef WB_NO_SLAB_STORE  // (measurement: the loop without its 28-56 MB slab burst; results are garbage)
  if (acc[0][0][0][0] != 12345.678f) return;
#endif
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int co = co0 + co_g * (16 * NT) + nt * 16 + (lane & 15);
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int ci = ci0 + ci_g * (16 * MI) + mi * 16 + (lane >> 4) * 4 + r;
          if (ci < a.Cin && co < a.Cout)
            a.slabs[(((int64_t)blockIdx.x * T + t) * a.Cin + ci) * a.Cout + co] = acc[t][mi][nt][r] * unscale;
        }
  }
}

template <int KS, int S, int TH, int TW, int NT, int MI>
static void wb_launch(const WgradBf3Args& a, dim3 grid, hipStream_t s) {
  constexpr int PPX = ((TH - 1) * S + KS) * ((TW - 1) * S + KS);
  constexpr size_t plane = (size_t)PPX * (64 * MI + 32) + (size_t)TH * TW * (64 * NT + 32);
  if (a.xz_mean && a.dz_p2) {
    if constexpr (KS == 3 && S == 1)
      hipLaunchKernelGGL((conv_wgrad_bf3_kernel<2, KS, S, TH, TW, NT, MI, false, true, true>), grid, dim3(256), 2 * plane, s, a);
  } else if (a.x_p2 && a.dz_p2)
    hipLaunchKernelGGL((conv_wgrad_bf3_kernel<2, KS, S, TH, TW, NT, MI, true, true>), grid, dim3(256), 2 * plane, s, a);
  else if (a.dz_p2 && a.x_amax)
    hipLaunchKernelGGL((conv_wgrad_bf3_kernel<2, KS, S, TH, TW, NT, MI, false, true>), grid, dim3(256), 2 * plane, s, a);
  else if (a.x_p2 && a.dz_amax)
    hipLaunchKernelGGL((conv_wgrad_bf3_kernel<2, KS, S, TH, TW, NT, MI, true>), grid, dim3(256), 2 * plane, s, a);
  else if (a.x_amax && a.dz_amax)
    hipLaunchKernelGGL((conv_wgrad_bf3_kernel<2, KS, S, TH, TW, NT, MI>), grid, dim3(256), 2 * plane, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_bf3_kernel<3, KS, S, TH, TW, NT, MI>), grid, dim3(256), 3 * plane, s, a);
}

// Launches the split-bf16 weight gradient of a conv (3x3 stride 1 / 2, pad 1; 1x1 stride 1 with >= 64
// channels on both sides) into `slabs` ([PS][k*k][Cin][Cout]); returns the number of slabs written, or
// 0 when the shape is not covered (the caller then uses the exact-fp32 kernel).  max_slabs bounds PS
// (workspace size).  x is (N, Hin, Win, Cin), dz (N, Hout, Wout, Cout).
// x_amax / dz_amax: both non-null = the fp16x2 split with those magnitude rows, else bf16x3.
int mval_launch_wgrad_bf3_p2(const float* x, const void* x_p2, const unsigned* x_p2_rows, const float* dz, float* slabs, int N, int Hin,
                             int Win, int Cin, int Hout, int Wout, int Cout, int k, int stride, int max_slabs, const unsigned* x_amax,
                             const unsigned* dz_amax, hipStream_t s, const void* dz_p2 = nullptr, const unsigned* dz_p2_rows = nullptr);
int mval_wgrad_bf3_covers(int Cin, int Cout, int k, int stride) {
  const char* e = getenv("MVAL_CONV");
  if ((e && e[0] == 'f') || (Cin & 3) || (Cout & 3) || Cin < 16 || Cout < 16) return 0;
  return (k == 3 && (stride == 1 || stride == 2)) || (k == 1 && stride == 1 && Cin >= 64 && Cout >= 64);
}

int mval_launch_wgrad_bf3(const float* x, const float* dz, float* slabs, int N, int Hin, int Win, int Cin, int Hout,
                          int Wout, int Cout, int k, int stride, int max_slabs, const unsigned* x_amax,
                          const unsigned* dz_amax, hipStream_t s) {
  return mval_launch_wgrad_bf3_p2(x, nullptr, nullptr, dz, slabs, N, Hin, Win, Cin, Hout, Wout, Cout, k, stride, max_slabs, x_amax, dz_amax, s);
}

static thread_local const float* g_wb_xz[4] = {nullptr, nullptr, nullptr, nullptr};
static thread_local float g_wb_xz_sqrt_m1 = 1.f;
// The NEXT weight gradient's x is relu(BatchNorm(z)) of the tensor passed as x (WgradBf3Args, XZ)
void mval_conv_wgrad_set_z_x(const float* mean, const float* invstd, const float* gamma, const float* beta, float sqrt_m1) {
  g_wb_xz[0] = mean; g_wb_xz[1] = invstd; g_wb_xz[2] = gamma; g_wb_xz[3] = beta;
  g_wb_xz_sqrt_m1 = sqrt_m1;
}

// x_p2 != nullptr: x as P2 planes (+ rows) instead of fp32 NHWC (needs dz_amax: the fp16 split; Cin % 8 == 0)
int mval_launch_wgrad_bf3_p2(const float* x, const void* x_p2, const unsigned* x_p2_rows, const float* dz, float* slabs, int N, int Hin,
                             int Win, int Cin, int Hout, int Wout, int Cout, int k, int stride, int max_slabs, const unsigned* x_amax,
                             const unsigned* dz_amax, hipStream_t s, const void* dz_p2, const unsigned* dz_p2_rows) {
  // (XZ: mval_conv_wgrad_set_z_x before the call -- x is then the producer's raw z; consumed by THIS call whatever it returns)
  const float* xz[4] = {g_wb_xz[0], g_wb_xz[1], g_wb_xz[2], g_wb_xz[3]};
  g_wb_xz[0] = g_wb_xz[1] = g_wb_xz[2] = g_wb_xz[3] = nullptr;
  const int refuse = xz[0] ? -1 : 0;  // (no other kernel can stand in: x is not the activation)
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("MVAL_CONV");
    enabled = (e && e[0] == 'f') ? 0 : 1;  // MVAL_CONV=fp32: exact-fp32 MFMA kernels everywhere
  }
  if (!enabled || (Cin & 3) || (Cout & 3) || Cin < 16 || Cout < 16) return refuse;
  const bool k3 = k == 3 && (stride == 1 || stride == 2);
  const bool k1 = k == 1 && stride == 1 && Cin >= 64 && Cout >= 64;
  if (!k3 && !k1) return refuse;
  WgradBf3Args a;
  a.x = x; a.dz = dz; a.slabs = slabs;
  a.N = N; a.Hin = Hin; a.Win = Win; a.H = Hout; a.W = Wout; a.Cin = Cin; a.Cout = Cout;
  a.x_amax = x_amax; a.dz_amax = dz_amax;
  a.xz_mean = xz[0]; a.xz_invstd = xz[1]; a.xz_gamma = xz[2]; a.xz_beta = xz[3];
  a.xz_sqrt_m1 = g_wb_xz_sqrt_m1;
  if (a.xz_mean && !(k == 3 && stride == 1 && dz_p2 && dz_p2_rows && (Cout & 7) == 0 && !x_p2)) return -1;  // (3x3 stride 1, dz as planes)
  a.dz_p2 = (dz_p2 && dz_p2_rows && (Cout & 7) == 0 && (x_p2 || x_amax || a.xz_mean)) ? reinterpret_cast<const _Float16*>(dz_p2) : nullptr;
  a.dz_p2_rows = dz_p2_rows;
  if (dz_p2 && !a.dz_p2) return 0;
  a.x_p2 = (x_p2 && (dz_amax || a.dz_p2) && (Cin & 7) == 0) ? reinterpret_cast<const _Float16*>(x_p2) : nullptr;
  a.x_p2_rows = x_p2_rows;
  if (x_p2 && !a.x_p2) return 0;
  // 8-wide tiles also for widths like 72 / 36 / 18 / 24 where they waste fewer (zero-padded) columns than 16-wide ones
  const int tw = (Wout > 8 && ((Wout + 15) / 16) * 16 <= ((Wout + 7) / 8) * 8) ? 16 : 8;
  const int th = (stride == 2 ? 32 : 64) / tw;  // stride-2 patches are ~4x larger per pixel: 32-pixel tiles
  a.tiles_x = (Wout + tw - 1) / tw;
  a.tiles_y = (Hout + th - 1) / th;
  a.ntiles = a.tiles_x * a.tiles_y * N;
  const int nt = (k1 || Cout > 32) ? 2 : 1, mi = k1 ? 2 : 1;
  const int cb = ((Cin + 32 * mi - 1) / (32 * mi)) * ((Cout + 32 * nt - 1) / (32 * nt));
#ifndef WB_PS_TARGET
#define WB_PS_TARGET 768  // (measurement knob: -DWB_PS_TARGET=512 / 384 / 256 -- profiles/r05/wgrad_slabs.log)
#endif
  int PS = WB_PS_TARGET / cb;  // ~3 workgroups per CU resident (50-62 KB of LDS each)
  if (PS < 1) PS = 1;
  if (PS > max_slabs) PS = max_slabs;
  if (PS > a.ntiles) PS = a.ntiles;
  a.PS = PS;
  dim3 grid(PS, (Cin + 32 * mi - 1) / (32 * mi), (Cout + 32 * nt - 1) / (32 * nt));
  if (k1) {
    if (tw == 16) wb_launch<1, 1, 4, 16, 2, 2>(a, grid, s);
    else wb_launch<1, 1, 8, 8, 2, 2>(a, grid, s);
  } else if (stride == 1) {
    if (tw == 16 && nt == 2) wb_launch<3, 1, 4, 16, 2, 1>(a, grid, s);
    else if (tw == 16) wb_launch<3, 1, 4, 16, 1, 1>(a, grid, s);
    else if (nt == 2) wb_launch<3, 1, 8, 8, 2, 1>(a, grid, s);
    else wb_launch<3, 1, 8, 8, 1, 1>(a, grid, s);
  } else {
    if (tw == 16 && nt == 2) wb_launch<3, 2, 2, 16, 2, 1>(a, grid, s);
    else if (tw == 16) wb_launch<3, 2, 2, 16, 1, 1>(a, grid, s);
    else if (nt == 2) wb_launch<3, 2, 4, 8, 2, 1>(a, grid, s);
    else wb_launch<3, 2, 4, 8, 1, 1>(a, grid, s);
  }
  return PS;
}
