// Fused BasicBlock on the fp16 matrix cores (hrnet.py:19-52 in eval mode):
//
//     out = relu( bn2(conv3x3(relu(bn1(conv3x3(x))))) + x )          x, out: NHWC fp32, C = 32, 48 or 64 channels
//     (48 = HRNet-W48's first branch: two 32-channel K chunks whose upper 16 channels are zeros in LDS)
//
// ONE launch instead of two conv launches, and the intermediate activation never leaves the CU.  These are the
// high-resolution branches of HRNet (32 channels on 64x64 maps, 64 on 32x32 at 256x256 input): 128 of HRNet-W32's 293
// launches, and as two separate convs they are bound by their HBM traffic and by per-workgroup latency, not by the
// matrix cores (the MFMA pipe is ~20 % busy on the 32-channel layer).  Per block the unfused pair moves
// x + mid + mid + x + out = 5 tensors through HBM; this kernel moves x + out (the residual re-read of x hits L2).
//
// Arithmetic = the fp16x2 split of conv_mfma_split.hip (PL = 2: three exact fp16 products per fp32 product, fp32
// accumulate, power-of-two scales undone exactly in the epilogues):
//   * x is scaled by its image's max |x| (the row its producer kept, conv_common.h);
//   * the intermediate tile is scaled by ITS OWN maximum (a workgroup-local reduction): the tile decomposition
//     depends on the image geometry only, so results stay deterministic and independent of the batch.
//
// Workgroup = 4 waves on an 8 x 16 output tile (the kernel is written for TH / 2 waves on TH x 16 tiles).  Every wave
// streams ALL weight fragments of both convs from L2 and computes all C output channels for its pixels:
//   1. stage the 12 x 20 input patch (2-pixel halo) through a buffer descriptor (out-of-image lanes read zeros
//      without a branch), split, into LDS planes X[plane][chunk][pixel][32 ch + pad];
//   2. conv1 over the 10 x 18 intermediate pixels (1-pixel halo for conv2; 12 sub-tiles of 16 pixel slots, 3 per
//      wave).  The WEIGHT fragment is the MFMA's first operand, so a lane ends up with 4 consecutive channels of one
//      pixel: BN1 + ReLU (+ zero outside the image = conv2's zero padding) in registers, tile maximum through LDS,
//      split, 8-byte LDS stores into M[plane][chunk][pixel][32 ch + pad], which overlays X (conv1 is done with it);
//   3. conv2 over the 8 x 16 output pixels from M (2 rows per wave); BN2 + residual + ReLU in registers, float4
//      stores straight from the accumulators (64-byte runs per pixel; no LDS round trip, no barrier).
// Measured alternatives (32-channel block, 128 images of 64 x 64; this form: 96 us): two waves per workgroup with
// twice the pixels each (half the weight stream through L1) 102 us; 16 x 16 tiles with 8 waves (10 % fewer conv1
// MFMAs, 17 % fewer staged pixels) 96.8 us.
// Halo recompute: conv1 does 12 / 8 = 1.5x the MFMAs of a plain conv; at 3 MFMAs per product that is cheaper than
// the two staging passes, two epilogues and 3 tensor round trips it replaces.
#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BK_TW 16
#define BK_XW (BK_TW + 4)  // 20: input patch width (2-pixel halo)
#define BK_MW (BK_TW + 2)  // 18: intermediate width (1-pixel halo)
#define BK_ROWB 80              // bytes per LDS pixel row: 32 fp16 + 8 pad (16 consecutive pixels cover all banks)
// tile height TH (8 or 16 output rows): TH / 2 waves, each with 3 intermediate sub-tiles and 2 output rows

__device__ __forceinline__ f32x4 bk_mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void bk_split(const f32x4 v, f16x4& h, f16x4& l) {
  h = __builtin_convertvector(v, f16x4);
  l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
}
// 2^s that puts a maximum with these float bits into [2^14, 2^15), and its inverse (zero / inf / NaN: unscaled)
__device__ __forceinline__ void bk_scale(unsigned amax_bits, float& mul, float& inv) {
  const int e = (int)((amax_bits >> 23) & 0xff);
  int s = (e == 0 || e == 255) ? 0 : 14 - (e - 127);
  s = max(-110, min(110, s));
  mul = __uint_as_float((unsigned)(127 + s) << 23);
  inv = __uint_as_float((unsigned)(127 - s) << 23);
}

struct BlockArgs {
  const float* in;
  float* out;
  const float *w1, *scale1, *shift1, *w1_unscale;
  const float *w2, *scale2, *shift2, *w2_unscale;
  const unsigned* in_amax;
  unsigned* out_amax;
  int N, H, W;
  int tiles_x, tiles_y;
};

template <int C, int TH>
__global__ __launch_bounds__(32 * TH) __attribute__((amdgpu_waves_per_eu(C == 32 ? 4 : 2, 8))) void conv_block_kernel(BlockArgs a) {
  constexpr int BK_TH = TH, BK_WAVES = TH / 2, BK_NTH = 64 * BK_WAVES;
  constexpr int BK_XH = BK_TH + 4, BK_MH = BK_TH + 2;
  constexpr int BK_XPX = BK_XH * BK_XW, BK_MPX = BK_MH * BK_MW;  // 240 / 180 pixels (TH = 8), 400 / 324 (TH = 16)
  constexpr int BK_MSLOTS = BK_WAVES * 3 * 16;                   // intermediate pixel slots: 3 sub-tiles of 16 per wave
  static_assert(BK_MSLOTS >= BK_MPX, "three sub-tiles per wave cover the intermediate tile");
  constexpr int NCH = (C + 31) / 32;  // 32-channel K chunks (C = 48: the second one is half zeros)
  constexpr int NS = C / 16;          // 16-cout sub-tiles: every wave computes all of them
  constexpr int NSP = NCH * 2;        // sub-tiles of the padded channel count (what the LDS planes hold)
  constexpr int Q = C / 4;            // float4 per pixel in memory
  constexpr int QP = NCH * 8;         // ... and in the LDS planes
  constexpr int NE = (BK_XPX * QP + BK_NTH - 1) / BK_NTH;
  // weight fragments in flight: one (tap, chunk) step is ~0.1 us of MFMAs per wave, an L2 hit several times that; two
  // steps ahead where the registers allow it (64 channels: 96 registers would cost the second wave per SIMD)
  constexpr int WD = C == 32 ? 3 : 2;
  constexpr int XCHUNK = BK_XPX * BK_ROWB, XPLANE = NCH * XCHUNK;
  constexpr int MCHUNK = BK_MSLOTS * BK_ROWB, MPLANE = NCH * MCHUNK;
  static_assert(2 * MPLANE <= 2 * XPLANE, "M overlays X");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float tile_max[BK_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  int t = blockIdx.x;
  const int txi = t % a.tiles_x;
  t /= a.tiles_x;
  const int tyi = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int oy0 = tyi * BK_TH, ox0 = txi * BK_TW;
  // the image through a buffer descriptor: out-of-image lanes get an out-of-range offset and read zeros, without
  // a branch per staged element
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (int64_t)n * a.H * a.W * C), 0, a.H * a.W * C * 4, 0x00020000);

  // ---- 1. input patch: global -> registers -> (scale, split) -> LDS ------------------------------------------
  f32x4 stage[NE];
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e = tid + BK_NTH * i;
    const int px = e / QP, q = e % QP;
    const int py = px / BK_XW, pxx = px - py * BK_XW;
    const int iy = oy0 - 2 + py, ix = ox0 - 2 + pxx;
    const bool ok = e < BK_XPX * QP && q < Q && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? ((iy * a.W + ix) * C + q * 4) * 4 : -1, 0, 0));
  }
  float in_mul, in_inv;
  bk_scale(conv_amax_read(a.in_amax + (int64_t)n * MVAL_AMAX_ROW), in_mul, in_inv);
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e = tid + BK_NTH * i;
    if (e < BK_XPX * QP) {
      const int px = e / QP, q = e % QP;
      f16x4 h, l;
      bk_split(stage[i] * in_mul, h, l);
      const int off = (q >> 3) * XCHUNK + px * BK_ROWB + (q & 7) * 8;
      *reinterpret_cast<f16x4*>(smem + off) = h;
      *reinterpret_cast<f16x4*>(smem + XPLANE + off) = l;
    }
  }
  __syncthreads();

  // Weight fragments ([tap][chunk][cout/16][plane h,l][lane][8 fp16], mval_pack_conv_weights) through buffer
  // descriptors: block offset in SGPRs, a per-lane 32-bit offset, the plane as immediate.
  const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w2), 0, 0x7fffffff, 0x00020000);
  constexpr int BLK = NS * 2048;  // bytes per (tap, chunk): NS sub-tiles x 2 planes x 1 KiB
  auto wfrag = [&](const __amdgpu_buffer_rsrc_t& r, int blk, int ns, int p) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (ns * 128 + p * 64 + lane) * 16, blk * BLK, 0);
  };

  // ---- 2. conv1 over the intermediate pixels: sub-tiles wave * MS1 + {0 .. MS1 - 1} -------------------------------
  constexpr int MS1 = 3;
  int xb[MS1];  // LDS byte offset of the lane's pixel (tap (0,0)) + its k octet
  int mpix[MS1];
#pragma unroll
  for (int ms = 0; ms < MS1; ms++) {
    const int p = (wave * MS1 + ms) * 16 + (lane & 15);
    mpix[ms] = p;
    const int pc = p < BK_MPX ? p : 0;
    const int my = pc / BK_MW, mx = pc - my * BK_MW;
    xb[ms] = (my * BK_XW + mx) * BK_ROWB + (lane >> 4) * 16;
  }
  // BN factors of the lane's 4 channels per sub-tile: with 32 channels they are fetched BEFORE the MFMA loop that
  // precedes their use (16 registers); with 64 channels those 32 registers would cost the second wave per SIMD
  constexpr bool PRE = C == 32;
  static_assert(C == 32 || C == 48 || C == 64, "fused BasicBlock: 32, 48 or 64 channels");
  f32x4 bn_sc[NS], bn_sh[NS];
  auto load_bn = [&](const float* sc, const float* sh) {
#pragma unroll
    for (int ns = 0; ns < NS; ns++) {
      bn_sc[ns] = *reinterpret_cast<const f32x4*>(sc + ns * 16 + (lane >> 4) * 4);
      bn_sh[ns] = *reinterpret_cast<const f32x4*>(sh + ns * 16 + (lane >> 4) * 4);
    }
  };
  if constexpr (PRE) load_bn(a.scale1, a.shift1);
  f32x4 acc1[MS1][NS];
#pragma unroll
  for (int ms = 0; ms < MS1; ms++)
#pragma unroll
    for (int ns = 0; ns < NS; ns++) acc1[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    u32x4 wf[WD][NS][2];  // fetched WD - 1 (tap, chunk) steps ahead
#pragma unroll
    for (int st = 0; st < WD - 1; st++)
#pragma unroll
      for (int ns = 0; ns < NS; ns++)
#pragma unroll
        for (int p = 0; p < 2; p++) wf[st][ns][p] = wfrag(w1r, (st % 9) * NCH + st / 9, ns, p);
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        const int step = ch * 9 + tap;
        if (step + WD - 1 < NCH * 9) {
          const int nt = (step + WD - 1) % 9, nc = (step + WD - 1) / 9;
#pragma unroll
          for (int ns = 0; ns < NS; ns++)
#pragma unroll
            for (int p = 0; p < 2; p++) wf[(step + WD - 1) % WD][ns][p] = wfrag(w1r, nt * NCH + nc, ns, p);
        }
        const int toff = ch * XCHUNK + ((tap / 3) * BK_XW + (tap % 3)) * BK_ROWB;
#pragma unroll
        for (int ms = 0; ms < MS1; ms++) {
          const u32x4 xh = *reinterpret_cast<const u32x4*>(smem + xb[ms] + toff);
          const u32x4 xl = *reinterpret_cast<const u32x4*>(smem + XPLANE + xb[ms] + toff);
#pragma unroll
          for (int ns = 0; ns < NS; ns++) {
            f32x4 c = acc1[ms][ns];
            c = bk_mfma(wf[step % WD][ns][1], xh, c);  // wl * xh
            c = bk_mfma(wf[step % WD][ns][0], xl, c);  // wh * xl
            acc1[ms][ns] = bk_mfma(wf[step % WD][ns][0], xh, c);
          }
        }
      }
    }
  }
  // BN1 + ReLU; zero outside the image (conv2's padding).  Lane = (pixel lane & 15, channels (lane >> 4) * 4 .. + 3).
  if constexpr (!PRE) load_bn(a.scale1, a.shift1);
  const float u1 = in_inv * *a.w1_unscale;
  float tmax = 0.f;
#pragma unroll
  for (int ms = 0; ms < MS1; ms++) {
    const int p = mpix[ms];
    const int my = p / BK_MW, mx = p - my * BK_MW;
    const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
    const bool inside = p < BK_MPX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
    for (int ns = 0; ns < NS; ns++) {
      f32x4 v = acc1[ms][ns] * (bn_sc[ns] * u1) + bn_sh[ns];
      v.x = inside ? mval_relu(v.x) : 0.f;
      v.y = inside ? mval_relu(v.y) : 0.f;
      v.z = inside ? mval_relu(v.z) : 0.f;
      v.w = inside ? mval_relu(v.w) : 0.f;
      acc1[ms][ns] = v;
      tmax = fmaxf(fmaxf(tmax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
  }
  tmax = wave_max(tmax);
  if (lane == 0) tile_max[wave] = tmax;
  __syncthreads();  // every wave is done reading X; the tile maximum is complete
  float m_mul, m_inv;
  float tm = tile_max[0];
#pragma unroll
  for (int w = 1; w < BK_WAVES; w++) tm = fmaxf(tm, tile_max[w]);
  bk_scale(__float_as_uint(tm), m_mul, m_inv);
#pragma unroll
  for (int ms = 0; ms < MS1; ms++) {
#pragma unroll
    for (int ns = 0; ns < NSP; ns++) {  // (sub-tiles past C are the zero padding of conv2's last K chunk)
      f16x4 h, l;
      bk_split(ns < NS ? acc1[ms][ns < NS ? ns : 0] * m_mul : (f32x4){0.f, 0.f, 0.f, 0.f}, h, l);
      const int off = (ns >> 1) * MCHUNK + mpix[ms] * BK_ROWB + ((ns & 1) * 16 + (lane >> 4) * 4) * 2;
      *reinterpret_cast<f16x4*>(smem + off) = h;
      *reinterpret_cast<f16x4*>(smem + MPLANE + off) = l;
    }
  }

  // ---- 3. conv2 over the output rows wave * MS2 + {0 .. MS2 - 1}; the residual loads travel during its MFMA loop --
  constexpr int MS2 = BK_TH / BK_WAVES;
  const int ox = ox0 + (lane & 15);
  f32x4 res[MS2][NS];
  int64_t ooff[MS2];
#pragma unroll
  for (int ms = 0; ms < MS2; ms++) {
    const int oy = oy0 + wave * MS2 + ms;
    const bool ok = oy < a.H && ox < a.W;
    ooff[ms] = ok ? (((int64_t)n * a.H + oy) * a.W + ox) * C + (lane >> 4) * 4 : -1;
#pragma unroll
    for (int ns = 0; ns < NS; ns++)
      res[ms][ns] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  xr, ok ? ((oy * a.W + ox) * C + (lane >> 4) * 4 + ns * 16) * 4 : -1, 0, 0));
  }
  if constexpr (PRE) load_bn(a.scale2, a.shift2);
  f32x4 acc2[MS2][NS];
#pragma unroll
  for (int ms = 0; ms < MS2; ms++)
#pragma unroll
    for (int ns = 0; ns < NS; ns++) acc2[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // M is complete
  {
    u32x4 wf[WD][NS][2];  // fetched WD - 1 (tap, chunk) steps ahead
#pragma unroll
    for (int st = 0; st < WD - 1; st++)
#pragma unroll
      for (int ns = 0; ns < NS; ns++)
#pragma unroll
        for (int p = 0; p < 2; p++) wf[st][ns][p] = wfrag(w2r, (st % 9) * NCH + st / 9, ns, p);
    const int mb = ((wave * MS2) * BK_MW + (lane & 15)) * BK_ROWB + (lane >> 4) * 16;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        const int step = ch * 9 + tap;
        if (step + WD - 1 < NCH * 9) {
          const int nt = (step + WD - 1) % 9, nc = (step + WD - 1) / 9;
#pragma unroll
          for (int ns = 0; ns < NS; ns++)
#pragma unroll
            for (int p = 0; p < 2; p++) wf[(step + WD - 1) % WD][ns][p] = wfrag(w2r, nt * NCH + nc, ns, p);
        }
        const int toff = ch * MCHUNK + ((tap / 3) * BK_MW + (tap % 3)) * BK_ROWB;
#pragma unroll
        for (int ms = 0; ms < MS2; ms++) {
          const u32x4 mh = *reinterpret_cast<const u32x4*>(smem + mb + ms * (BK_MW * BK_ROWB) + toff);
          const u32x4 ml = *reinterpret_cast<const u32x4*>(smem + MPLANE + mb + ms * (BK_MW * BK_ROWB) + toff);
#pragma unroll
          for (int ns = 0; ns < NS; ns++) {
            f32x4 c = acc2[ms][ns];
            c = bk_mfma(wf[step % WD][ns][1], mh, c);
            c = bk_mfma(wf[step % WD][ns][0], ml, c);
            acc2[ms][ns] = bk_mfma(wf[step % WD][ns][0], mh, c);
          }
        }
      }
    }
  }
  if constexpr (!PRE) load_bn(a.scale2, a.shift2);
  const float u2 = m_inv * *a.w2_unscale;
  float amax = 0.f;
#pragma unroll
  for (int ms = 0; ms < MS2; ms++) {
    if (ooff[ms] < 0) continue;
#pragma unroll
    for (int ns = 0; ns < NS; ns++) {
      f32x4 v = acc2[ms][ns] * (bn_sc[ns] * u2) + bn_sh[ns];
      v += res[ms][ns];
      v.x = mval_relu(v.x); v.y = mval_relu(v.y); v.z = mval_relu(v.z); v.w = mval_relu(v.w);
      *reinterpret_cast<f32x4*>(a.out + ooff[ms] + ns * 16) = v;
      amax = fmaxf(fmaxf(amax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
  }
  if (a.out_amax)
    conv_amax_commit(a.out_amax + (int64_t)n * MVAL_AMAX_ROW, (int)(blockIdx.x % (unsigned)(a.tiles_x * a.tiles_y)), a.tiles_x * a.tiles_y, amax);
}

// 1 when the fused kernel covers this geometry
int mval_conv_block_supported(int C, int N, int H, int W) {
  if (C != 32 && C != 48 && C != 64) return 0;
  if (H < 8 || W < 16) return 0;  // small maps: the unfused kernels pack several images into a tile instead
  if ((int64_t)N * H * W * C >= (int64_t)1 << 31) return 0;
  if ((int64_t)H * W * C >= (int64_t)1 << 29) return 0;  // one image's bytes index a 32-bit buffer descriptor
  return 1;
}

int mval_launch_conv_block(int C, const float* in, float* out, const float* w1, const float* scale1, const float* shift1,
                           const float* w1_unscale, const float* w2, const float* scale2, const float* shift2,
                           const float* w2_unscale, const unsigned* in_amax, unsigned* out_amax, int N, int H, int W,
                           hipStream_t s) {
  if (!mval_conv_block_supported(C, N, H, W) || !in_amax) return 1;
  BlockArgs a;
  a.in = in; a.out = out;
  a.w1 = w1; a.scale1 = scale1; a.shift1 = shift1; a.w1_unscale = w1_unscale;
  a.w2 = w2; a.scale2 = scale2; a.shift2 = shift2; a.w2_unscale = w2_unscale;
  a.in_amax = in_amax; a.out_amax = out_amax;
  a.N = N; a.H = H; a.W = W;
  // 8-row tiles.  16-row tiles (8 waves, two workgroups per CU for 32 channels: 10 % fewer conv1 MFMAs and 17 % fewer
  // staged pixels, halo 324 / 256 vs 180 / 128) measured the same: 96.8 vs 96 us per 32-channel block.
  const int th = 8;
  a.tiles_x = (W + BK_TW - 1) / BK_TW;
  a.tiles_y = (H + th - 1) / th;
  const int tiles = a.tiles_x * a.tiles_y;
  if (out_amax && (int64_t)tiles * (th / 2) > MVAL_AMAX_ROW - 1)
    mval_launch_zero_rows(out_amax, (int64_t)N * MVAL_AMAX_ROW, s);
  const size_t smem = (size_t)2 * ((C + 31) / 32) * (th + 4) * BK_XW * BK_ROWB;
  dim3 grid((unsigned)(tiles * N));
  if (C == 32 && th == 16)
    hipLaunchKernelGGL((conv_block_kernel<32, 16>), grid, dim3(512), smem, s, a);
  else if (C == 32)
    hipLaunchKernelGGL((conv_block_kernel<32, 8>), grid, dim3(256), smem, s, a);
  else if (C == 48)
    hipLaunchKernelGGL((conv_block_kernel<48, 8>), grid, dim3(256), smem, s, a);
  else
    hipLaunchKernelGGL((conv_block_kernel<64, 8>), grid, dim3(256), smem, s, a);
  return 0;
}
