// Weight gradient of the 3x3 stride-1 convs on the bf16 matrix cores with the exact three-way
// bf16 split (see conv_mfma_bf3.hip): 2.67x the MFMA rate of the exact-fp32 kernel in
// conv_wgrad.hip, same results to fp32 rounding.
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
//     54 NT MFMAs;
//   * split-K over workgroups, per-split slabs and the float64 slab reduction exactly as in
//     conv_wgrad.hip (same slab layout, same reduce kernel); the next tile's operands travel
//     global -> registers during the MFMA loop.
#include <stdlib.h>

#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

#define WB_XROW 96  // bytes per patch pixel and plane: 32 cin bf16 + 32 pad

struct WgradBf3Args {
  const float* x;   // NHWC (N, H, W, Cin)
  const float* dz;  // NHWC (N, H, W, Cout)
  float* slabs;     // [PS][9][Cin][Cout]
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, ntiles, PS;
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

// 16 channels x 8 k of one plane: two transposed reads (k 0..3 and 4..7 of this lane's k-group)
__device__ __forceinline__ bf16x8 wb_frag(const char* lo, const char* hi) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(hi));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int TW, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_bf3_kernel(WgradBf3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TH = 64 / TW, MT = 64;
  constexpr int PH = TH + 2, PW = TW + 2, PPX = PH * PW;
  constexpr int CO = 32 * NT;               // couts per workgroup
  constexpr int ZROW = CO * 2 + 32;         // bytes per dz pixel and plane (96 / 160)
  constexpr int XPLANE = PPX * WB_XROW, ZPLANE = MT * ZROW;
  constexpr int NEX = (PPX * 8 + 255) / 256, NEZ = MT * (CO / 4) / 256;
  char* xl = smem;                 // [3][PPX][WB_XROW]
  char* zl = smem + 3 * XPLANE;    // [3][MT][ZROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci_t = wave & 1, co_p = wave >> 1;  // cin tile (16) / cout group (16 * NT) of this wave
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * CO;

  f32x4 acc[9][NT];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
  const int xoff = (py_lo * PW + px_lo) * WB_XROW + ci_t * 32 + p * 8;
  const int zoff = (py_lo * TW + px_lo) * ZROW + co_p * (32 * NT) + p * 8;

  // staging slots (tile-invariant)
  const int q4 = (tid & 7) * 4;
  int xrow[NEX], xcol[NEX];
#pragma unroll
  for (int i = 0; i < NEX; i++) {
    const int e = tid + 256 * i, px = e >> 3;
    xrow[i] = px / PW;
    xcol[i] = px - xrow[i] * PW;
  }
  constexpr int ZQ = CO / 4;  // float4 per dz pixel
  const int zq4 = (tid % ZQ) * 4;
  const bool cx_ok = ci0 + q4 < a.Cin, cz_ok = co0 + zq4 < a.Cout;

  f32x4 xr[NEX], zr[NEZ];
  auto load_tile = [&](int t) {
    const int txi = t % a.tiles_x;
    t /= a.tiles_x;
    const int oy0 = (t % a.tiles_y) * TH, ox0 = txi * TW, n = t / a.tiles_y;
#pragma unroll
    for (int i = 0; i < NEX; i++) {
      const int iy = oy0 - 1 + xrow[i], ix = ox0 - 1 + xcol[i];
      const bool ok = tid + 256 * i < PPX * 8 && cx_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      xr[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + (((int64_t)n * a.H + iy) * a.W + ix) * a.Cin + ci0 + q4)
                 : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NEZ; i++) {
      const int pz = (tid + 256 * i) / ZQ;
      const int y = oy0 + pz / TW, x = ox0 + pz % TW;
      zr[i] = (cz_ok && y < a.H && x < a.W)
                  ? *reinterpret_cast<const f32x4*>(a.dz + (((int64_t)n * a.H + y) * a.W + x) * a.Cout + co0 + zq4)
                  : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) load_tile(tile);
  for (; tile < a.ntiles; tile += a.PS) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NEX; i++) {
      const int e = tid + 256 * i;
      if (e < PPX * 8) wb_split_store(xr[i], xl + (e >> 3) * WB_XROW + (e & 7) * 8, XPLANE);
    }
#pragma unroll
    for (int i = 0; i < NEZ; i++) {
      const int e = tid + 256 * i;
      wb_split_store(zr[i], zl + (e / ZQ) * ZROW + (e % ZQ) * 8, ZPLANE);
    }
    __syncthreads();
    if (tile + a.PS < a.ntiles) load_tile(tile + a.PS);
#pragma unroll
    for (int ks = 0; ks < TH / KROWS; ks++) {
      // dz fragments of this k-step: [cout tile][plane]
      bf16x8 zf[NT][3];
      const char* zb = zl + zoff + ks * KROWS * TW * ZROW;
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
          zf[nt][pl] = wb_frag(zb + pl * ZPLANE + nt * 32, zb + pl * ZPLANE + nt * 32 + py_step * TW * ZROW);
      const char* xb = xl + xoff + ks * KROWS * PW * WB_XROW;
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const char* xt = xb + ((t / 3) * PW + t % 3) * WB_XROW;
        const bf16x8 xh = wb_frag(xt, xt + py_step * PW * WB_XROW);
        const bf16x8 xm = wb_frag(xt + XPLANE, xt + XPLANE + py_step * PW * WB_XROW);
        const bf16x8 xlo = wb_frag(xt + 2 * XPLANE, xt + 2 * XPLANE + py_step * PW * WB_XROW);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          f32x4 c = acc[t][nt];  // small terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xlo, zf[nt][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, zf[nt][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, zf[nt][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, zf[nt][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, zf[nt][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, zf[nt][0], c, 0, 0, 0);
          acc[t][nt] = c;
        }
      }
    }
  }
  // slab[ps][t][ci][co]; C layout: row (cin) = (lane >> 4) * 4 + r, col (cout) = lane & 15
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int co = co0 + co_p * (16 * NT) + nt * 16 + (lane & 15);
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ci = ci0 + ci_t * 16 + (lane >> 4) * 4 + r;
        if (ci < a.Cin && co < a.Cout)
          a.slabs[(((int64_t)blockIdx.x * 9 + t) * a.Cin + ci) * a.Cout + co] = acc[t][nt][r];
      }
  }
}

// Launches the split-bf16 weight gradient of a 3x3 stride-1 pad-1 conv into `slabs`
// ([PS][9][Cin][Cout]); returns the number of slabs written, or 0 when the shape is not covered
// (the caller then uses the exact-fp32 kernel).  max_slabs bounds PS (workspace size).
int mval_launch_wgrad_bf3(const float* x, const float* dz, float* slabs, int N, int H, int W, int Cin, int Cout,
                          int max_slabs, hipStream_t s) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("MVAL_CONV");
    enabled = (e && e[0] == 'f') ? 0 : 1;  // MVAL_CONV=fp32: exact-fp32 MFMA kernels everywhere
  }
  if (!enabled || (Cin & 3) || (Cout & 3) || Cin < 16 || Cout < 16) return 0;
  WgradBf3Args a;
  a.x = x; a.dz = dz; a.slabs = slabs;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  const int tw = W > 8 ? 16 : 8, th = 64 / tw;
  a.tiles_x = (W + tw - 1) / tw;
  a.tiles_y = (H + th - 1) / th;
  a.ntiles = a.tiles_x * a.tiles_y * N;
  const int nt = Cout > 32 ? 2 : 1;
  const int cb = ((Cin + 31) / 32) * ((Cout + 32 * nt - 1) / (32 * nt));
  int PS = 768 / cb;  // ~3 workgroups per CU resident (50-62 KB of LDS each)
  if (PS < 1) PS = 1;
  if (PS > max_slabs) PS = max_slabs;
  if (PS > a.ntiles) PS = a.ntiles;
  a.PS = PS;
  const int ppx = (th + 2) * (tw + 2);
  const size_t smem = (size_t)3 * ppx * WB_XROW + (size_t)3 * 64 * (64 * nt + 32);
  dim3 grid(PS, (Cin + 31) / 32, (Cout + 32 * nt - 1) / (32 * nt));
  if (tw == 16 && nt == 2) hipLaunchKernelGGL((conv_wgrad_bf3_kernel<16, 2>), grid, dim3(256), smem, s, a);
  else if (tw == 16) hipLaunchKernelGGL((conv_wgrad_bf3_kernel<16, 1>), grid, dim3(256), smem, s, a);
  else if (nt == 2) hipLaunchKernelGGL((conv_wgrad_bf3_kernel<8, 2>), grid, dim3(256), smem, s, a);
  else hipLaunchKernelGGL((conv_wgrad_bf3_kernel<8, 1>), grid, dim3(256), smem, s, a);
  return PS;
}
