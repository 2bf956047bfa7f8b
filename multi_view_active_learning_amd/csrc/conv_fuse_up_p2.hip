// One launch for the up-sampling terms of an HRNet fuse-layer output (hrnet.py:269-287, 424-447) over P2 activations:
//
//     out = act( ((r + up_{u1}(bn1(conv1x1(x_1)))) + up_{u2}(bn2(conv1x1(x_2)))) [+ up_{u3}(bn3(conv1x1(x_3)))] )
//           r: the partial sum so far (the branch's own activation, or the output of the down-sampling terms), C = 32 / 64
//           (HRNet-W32) or 48 / 96 (HRNet-W48, round 4: 8 x 24 / 4 x 36 tiles for its 72- / 36-pixel-wide maps) channels at H x W;  x_j: C * 2^u_j ... channels at (H >> u_j) x (W >> u_j);  up_u = nearest neighbour, 2^u
//
// Op by op every term reads the full-resolution partial sum and writes it again (the 1x1 convs at the low resolutions are
// nothing next to that: 134 MB per link for the 32-channel 64x64 branch of 128 images, 3 links = 0.40 GB + 0.06 GB of inputs,
// ~135 us at the ~4 TB/s those launches reach).  Fused: r is read once and out written once (0.19 GB).  The additions keep
// the reference's left-to-right order in fp32 (the chain rounds each partial sum to the 22-bit pair format in between).
//
// Workgroup = 4 waves on an 8 x 32 output tile (16 x 32 tiles: 1 024 workgroups of 140 registers = 1.33 rounds of the chip; 8 x 32:
// 2 048 of ~90):
//   1.  the 1x1 convs of the terms on the matrix cores (conv_p2.hip arithmetic), B fragments straight from global memory
//       (no halo, no reuse between pixels): term j covers (8 >> u_j) x (32 >> u_j) low-resolution pixels in sixteen-pixel
//       fragments; BN_j in registers, fp32 results into LDS [term][pixel][C];
//   2.  every thread finishes whole 8-channel granules of the output: r's (h, l) granules from global, the terms' values of
//       its low-resolution parents from LDS (lanes of a 2^u block read the same address), sum, act, max |x|, split, two
//       16-byte stores.
#include <stdlib.h>

#include "conv_p2.h"

typedef p2_f32x4 f32x4;
typedef p2_f16x8 f16x8;
typedef p2_f16x4 f16x4;
typedef p2_u32x4 u32x4;
typedef p2_u32x2 u32x2;

#define FU_MAX_TERMS 3
#ifndef FU_TH
#define FU_TH 8  // rows of the output tile (x 32 columns)
#endif

struct FuseUpTerm {
  const _Float16* in;
  const unsigned* in_row;
  unsigned w, w_unscale, scale, shift, bound;  // byte offsets into params
  int cin, up;
};
struct FuseUpArgs {
  const _Float16* res;
  const unsigned* res_row;
  _Float16* out;
  unsigned* out_row;
  const float* params;
  FuseUpTerm t[FU_MAX_TERMS];
  int n_terms, relu;
  int N, H, W;
  int tiles_x, tiles_y;
};

__device__ __forceinline__ f32x4 fu_mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int COUT, int TH = FU_TH, int TW = 32>
__global__ __launch_bounds__(256) void conv_fuse_up_p2_kernel(FuseUpArgs a) {
  static_assert(COUT == 32 || COUT == 64 || COUT == 48 || COUT == 96, "fuse-layer outputs of 32 / 64 (HRNet-W32) or 48 / 96 (HRNet-W48) channels");
  constexpr int NCT = COUT / 16, C8 = COUT / 8;
  constexpr bool EVEN = 4 % NCT == 0;  // cout sub-tiles divide the four waves (32 / 64 channels): wave = (sub-tile, fragment group)
  constexpr int NH = EVEN ? 4 / NCT : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned wgmax;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_img = a.tiles_x * a.tiles_y;
  const int n = (int)blockIdx.x / tiles_img, tr = (int)blockIdx.x - n * tiles_img;
  const int oy0 = (tr / a.tiles_x) * TH, ox0 = (tr % a.tiles_x) * TW;
  if (tid == 0) wgmax = 0u;

  const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.params), 0, 0x7fffffff, 0x00020000);
  auto ps = [&](unsigned off) -> float { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.params) + off); };
  P2RowRegs row_r;
  p2_row_request(a.res_row, n, row_r);
  // the partial sum's granules of this thread's items travel during the matrix phase
  constexpr int NITEMS = TH * TW * C8, ITEMS = (NITEMS + 255) / 256;
  const unsigned ohw16 = (unsigned)(a.H * a.W) * 16u, oimg = 2u * C8 * ohw16, oplane = C8 * ohw16;
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res), 0, (unsigned)a.N * oimg, 0x00020000);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.N * oimg, 0x00020000);
  unsigned go[ITEMS];
  u32x4 R[ITEMS][2];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const int e = tid + 256 * i;
    const int c8 = e / (TH * TW), q = e - c8 * (TH * TW);
    const int ly = q / TW, lx = q - ly * TW;
    const int y = oy0 + ly, x = ox0 + lx;
    go[i] = (e < NITEMS && y < a.H && x < a.W) ? (unsigned)n * oimg + (unsigned)((c8 * a.H + y) * a.W + x) * 16u : 0x80000000u;
    R[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rr, go[i], 0, 0);
    R[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rr, __builtin_elementwise_add_sat(go[i], oplane), 0, 0);
  }

  // ---- 1. the terms' 1x1 convs: wave = cout sub-tile ct, every NH-th sixteen-pixel fragment ---------------------------------
  const int wrow = lane & 15, wsrc = (lane & 48) | ((wrow & 3) | ((wrow & 4) << 1) | ((wrow & 8) >> 1));
  const int cq = ((lane >> 4) & 1) * 8 + (lane >> 5) * 4;  // the lane's four output channels inside a cout sub-tile
  float bound = 0.f;
  int lds_base = 0;
#pragma unroll
  for (int j = 0; j < FU_MAX_TERMS; j++) {
    if (j >= a.n_terms) break;
    const FuseUpTerm& t = a.t[j];
    const int u = t.up, Hj = a.H >> u, Wj = a.W >> u, th = TH >> u, tw = TW >> u, npx = th * tw;
    const int nfrag = (npx + 15) >> 4, C8j = t.cin >> 3, nks = t.cin >> 5;
    const unsigned hw16 = (unsigned)(Hj * Wj) * 16u, img = 2u * C8j * hw16, plane = C8j * hw16;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(t.in), 0, (unsigned)a.N * img, 0x00020000);
    P2RowRegs row_j;
    p2_row_request(t.in_row, n, row_j);
    const int blk = NCT * 2048;  // bytes per 32-channel step of the packed weights
    // work item = (cout sub-tile ct, sixteen-pixel fragment f): 32 / 64 channels -- a wave keeps its sub-tile and walks every
    // NH-th fragment; 48 / 96 channels (3 / 6 sub-tiles) -- the items are dealt to the four waves in turn
    for (int wi = wave; wi < (EVEN ? 4 * ((nfrag + NH - 1) / NH) : NCT * nfrag); wi += 4) {
      const int ct = EVEN ? wave % NCT : wi % NCT, f = EVEN ? (wave / NCT) + NH * (wi >> 2) : wi / NCT;
      if (f >= nfrag) break;
      const int c0 = ct * 16 + cq;
      const f32x4 sc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, c0 * 4, t.scale, 0));
      const f32x4 sh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, c0 * 4, t.shift, 0));
      const unsigned wv = t.w + (unsigned)((ct * 128 + wsrc) * 16);
      // the lane's pixel of the fragment (MFMA column lane & 15), its k octet = 8-channel block lane >> 4 of a step
      const int p = f * 16 + (lane & 15);
      const int ly = p / tw, lx = p - ly * tw;
      const int y = (oy0 >> u) + ly, x = (ox0 >> u) + lx;
      const bool ok = p < npx && y < Hj && x < Wj;
      const unsigned xo = ok ? (unsigned)n * img + (unsigned)(((lane >> 4) * Hj + y) * Wj + x) * 16u : 0x80000000u;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int s = 0; s < nks; s++) {
        const u32x4 xh = __builtin_amdgcn_raw_buffer_load_b128(xr, __builtin_elementwise_add_sat(xo, (unsigned)s * 4u * hw16), 0, 0);
        const u32x4 xl = __builtin_amdgcn_raw_buffer_load_b128(xr, __builtin_elementwise_add_sat(xo, (unsigned)s * 4u * hw16 + plane), 0, 0);
        const u32x4 wh = __builtin_amdgcn_raw_buffer_load_b128(pr, wv, s * blk, 0);
        const u32x4 wl = __builtin_amdgcn_raw_buffer_load_b128(pr, wv + 1024, s * blk, 0);
        acc = fu_mfma(wl, xh, acc);
        acc = fu_mfma(wh, xl, acc);
        acc = fu_mfma(wh, xh, acc);
      }
      const float unscale = __uint_as_float(row_j.inv) * ps(t.w_unscale);
      const f32x4 v = acc * (sc * unscale) + sh;
      if (p < npx) *reinterpret_cast<f32x4*>(smem + lds_base + (p * COUT + c0) * 4) = v;
    }
    bound += ps(t.bound) * p2_row_amax(row_j) + ps(t.bound + 4);
    lds_base += npx * COUT * 4;
  }
  const float r_inv = __uint_as_float(row_r.inv);
  bound += p2_row_amax(row_r);
  float out_mul, out_inv;
  p2_scale_of(bound, out_mul, out_inv);
  if (tr == 0 && tid == 0) a.out_row[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);
  __syncthreads();

  // ---- 2. whole granules: item = (8-channel block, pixel of the tile), a row of 32 pixels per half wave ------------------------
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const int e = tid + 256 * i;
    const int c8 = e / (TH * TW), q = e - c8 * (TH * TW);
    const int ly = q / TW, lx = q - ly * TW;
    const f16x8 h8 = __builtin_bit_cast(f16x8, R[i][0]), l8 = __builtin_bit_cast(f16x8, R[i][1]);
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = ((float)h8[k] + (float)l8[k]) * r_inv;
    int tb = 0;
#pragma unroll
    for (int j = 0; j < FU_MAX_TERMS; j++) {
      if (j >= a.n_terms) break;
      const int u = a.t[j].up, tw = TW >> u;
      const char* tp = smem + tb + (((ly >> u) * tw + (lx >> u)) * COUT + c8 * 8) * 4;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 16);
      v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w;
      v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
      tb += (TH >> u) * tw * COUT * 4;
    }
    const bool ok = go[i] != 0x80000000u;
    f16x8 ho, lo;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (a.relu) v[k] = p2_max_nan(v[k], 0.f);
      if (ok) amax = fmaxf(amax, fabsf(v[k]));
      const float sv = v[k] * out_mul;
      ho[k] = (_Float16)sv;
      lo[k] = (_Float16)(sv - (float)ho[k]);
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ho), orr, go[i], 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), orr, __builtin_elementwise_add_sat(go[i], oplane), 0, 0);
  }
  const unsigned bits = p2_wave_umax(__float_as_uint(amax));
  if (lane == 0) atomicMax(&wgmax, bits);
  __syncthreads();
  if (tid == 0) p2_slot_put(a.out_row + (int64_t)n * P2_ROW, tr, tiles_img, wgmax);
}

// tile of the output a workgroup covers: 8 x 32 (32 / 64 / 48 channels), 8 x 24 for 48 channels on widths that 24 divides and 32 does not
// (HRNet-W48's 96 x 72 branch: 3 tiles of 24 instead of 2.25 of 32), 4 x 36 / 4 x 32 for 96 channels (12 granule blocks per pixel)
static void fuse_up_tile(int cout, int W, int* th, int* tw) {
  *th = cout == 96 ? 4 : FU_TH;
  *tw = 32;
  if (cout == 48 && W % 24 == 0 && W % 32 != 0) *tw = 24;
  if (cout == 96 && W % 36 == 0 && W % 32 != 0) *tw = 36;
}

int mval_conv_fuse_up_p2_supported(int cout, int n_terms, const int* cin, const int* up, int N, int H, int W) {
  if ((cout != 32 && cout != 64 && cout != 48 && cout != 96) || n_terms < 2 || n_terms > FU_MAX_TERMS) return 0;
  int th, tw;
  fuse_up_tile(cout, W, &th, &tw);
  for (int j = 0; j < n_terms; j++)
    if ((th >> up[j]) < 1 || (tw & ((1 << up[j]) - 1))) return 0;
  for (int j = 0; j < n_terms; j++) {
    if (up[j] < 1 || up[j] > 3 || (cin[j] & 31) || cin[j] > 512) return 0;
    if ((H & ((1 << up[j]) - 1)) || (W & ((1 << up[j]) - 1))) return 0;
    if ((int64_t)N * (H >> up[j]) * (W >> up[j]) * cin[j] >= (int64_t)1 << 29) return 0;
  }
  if (H < 8 || W < 8 || (int64_t)N * H * W * cout >= (int64_t)1 << 29) return 0;
  return 1;
}

int mval_launch_conv_fuse_up_p2(int cout, int n_terms, int relu, const void* res, const unsigned* res_row, void* out, unsigned* out_row, const float* params,
                                const void* const* in, const unsigned* const* in_row, const int* cin, const int* up, const int64_t* w,
                                const int64_t* w_unscale, const int64_t* scale, const int64_t* shift, const int64_t* bound, int N, int H, int W,
                                hipStream_t s) {
  if (!mval_conv_fuse_up_p2_supported(cout, n_terms, cin, up, N, H, W)) return 1;
  FuseUpArgs a = {};
  a.res = reinterpret_cast<const _Float16*>(res); a.res_row = res_row;
  a.out = reinterpret_cast<_Float16*>(out); a.out_row = out_row;
  a.params = params;
  a.n_terms = n_terms; a.relu = relu;
  a.N = N; a.H = H; a.W = W;
  size_t lds = 0;
  int th, tw;
  fuse_up_tile(cout, W, &th, &tw);
  for (int j = 0; j < n_terms; j++) {
    if ((w[j] | w_unscale[j] | scale[j] | shift[j] | bound[j]) < 0 || (w[j] | w_unscale[j] | scale[j] | shift[j] | bound[j]) >= ((int64_t)1 << 28)) return 1;
    a.t[j].in = reinterpret_cast<const _Float16*>(in[j]); a.t[j].in_row = in_row[j];
    a.t[j].w = (unsigned)w[j] * 4u; a.t[j].w_unscale = (unsigned)w_unscale[j] * 4u; a.t[j].scale = (unsigned)scale[j] * 4u;
    a.t[j].shift = (unsigned)shift[j] * 4u; a.t[j].bound = (unsigned)bound[j] * 4u;
    a.t[j].cin = cin[j]; a.t[j].up = up[j];
    lds += (size_t)(th >> up[j]) * (tw >> up[j]) * cout * 4;
  }
  a.tiles_x = (W + tw - 1) / tw;
  a.tiles_y = (H + th - 1) / th;
  const int tiles_img = a.tiles_x * a.tiles_y;
  if (tiles_img > P2_SLOTS) mval_launch_zero_rows(out_row, (int64_t)N * P2_ROW, s);
  const dim3 grid((unsigned)(tiles_img * N));
  if (cout == 32) hipLaunchKernelGGL((conv_fuse_up_p2_kernel<32>), grid, dim3(256), lds, s, a);
  else if (cout == 64) hipLaunchKernelGGL((conv_fuse_up_p2_kernel<64>), grid, dim3(256), lds, s, a);
  else if (cout == 48 && tw == 24) hipLaunchKernelGGL((conv_fuse_up_p2_kernel<48, FU_TH, 24>), grid, dim3(256), lds, s, a);
  else if (cout == 48) hipLaunchKernelGGL((conv_fuse_up_p2_kernel<48, FU_TH, 32>), grid, dim3(256), lds, s, a);
  else if (tw == 36) hipLaunchKernelGGL((conv_fuse_up_p2_kernel<96, 4, 36>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv_fuse_up_p2_kernel<96, 4, 32>), grid, dim3(256), lds, s, a);
  return 0;
}
