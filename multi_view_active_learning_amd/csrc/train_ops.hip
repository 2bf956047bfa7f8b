// Train-mode BatchNorm and the elementwise forward/backward pieces of a fused operator
//   out = act(((bn(z) [nearest-upsampled 2^up]) + res1) + res2),   z = conv(x, w)
// (reference: nn.BatchNorm2d in training mode + ReLU + residual adds inside
// pose_estimators/hrnet.py:36-52,75-95,269-287, differentiated by strategy.py:478
// ``batch_loss.backward()``).  All tensors NHWC fp32; every kernel is an HBM-bound stream with
// float4 lanes along channels; per-channel reductions accumulate in float64 and are
// two-stage (per-workgroup partials -> finalize) so results are deterministic.
#include "conv_p2.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" int mval_bn_apply_fwd_mask(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                                      int, int, int, int, int, int, uint32_t*, uint8_t*, void*);
extern "C" int mval_bn_bwd_fused_mask(const float*, const float*, const uint8_t*, const float*, const float*, const float*, const float*,
                                      const float*, float*, float*, float*, float*, float*, double*, float*, int, int, int, int, int, int,
                                      uint32_t*, void*);
extern "C" int mval_bn_bwd_fused_p2(const float*, const float*, const uint8_t*, const float*, const float*, const float*, const float*,
                                    const float*, float*, float*, float*, float*, float*, double*, float*, int, int, int, int, int, int,
                                    uint32_t*, void*, uint32_t*, float*, uint32_t*, void*);
#define TR_BLOCKS 512
// The two elementwise BatchNorm streams run 1024-thread workgroups, at most two per CU: every workgroup leaves ONE
// partial maximum, and every workgroup of the consuming conv reads them all -- 2048 partials (256-thread workgroups)
// cost the 32-channel convs 10-14 us each (134 MB of L2 reads per launch), 512 cost ~2.
#define TR_APPLY_THREADS 1024
#define TR_APPLY_BLOCKS 512

// max |value| of what this workgroup wrote -> its slot of the tensor's magnitude row ([count, partials ...], the
// activation scale of the fp16-split convs, conv_common.h; ONE row per tensor in training: BatchNorm couples the
// batch anyway).  Every thread of the workgroup calls it.
__device__ __forceinline__ void tr_amax_store(unsigned* row, float m) {
  __shared__ float red[TR_APPLY_THREADS / 64];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) m = fmaxf(m, red[w]);
    row[0] = gridDim.x;
    row[1 + blockIdx.x] = __float_as_uint(m);
  }
}
#define TR_UNROLL 8
#define TR_EW 4  // pixels per thread and pass in the backward reduction

// Train-mode normalisation of one float4 of z: alpha = invstd * gamma, beta' = beta - mean * alpha, r = z * alpha + beta'
// (torch's batch_norm elementwise formula).  Explicit FMAs: the forward apply and the backward kernels that RECOMPUTE
// the ReLU mask from z (no residual: out > 0 <=> r > 0) must round identically whatever the compiler contracts.
__device__ __forceinline__ f32x4 bn_affine(const f32x4 zv, const f32x4 mu, const f32x4 is, const f32x4 g, const f32x4 b) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float alpha = is[k] * g[k];
    r[k] = __builtin_fmaf(zv[k], alpha, __builtin_fmaf(-mu[k], alpha, b[k]));
  }
  return r;
}

// ---- batch statistics -----------------------------------------------------------------
// partial[block][c][2] = (sum, sum of squares) over the block's pixel slice
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, double* __restrict__ part,
                                                               int64_t M, int C) {
  extern __shared__ double sh[];  // [rows_per_block][C][2] folded below
  const int c4n = C >> 2;
  const int lanes = min(c4n, 256);            // float4 columns handled per pass
  const int rows = 256 / lanes;               // pixel rows in flight per block
  const int col = threadIdx.x % lanes, row = threadIdx.x / lanes;
  for (int cb = 0; cb < c4n; cb += lanes) {
    const int q = cb + col;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (q < c4n && row < rows) {
      // TR_UNROLL independent 16-byte loads in flight per thread: 512 workgroups x 256 threads with
      // one load each keep only 2 MB in flight, far below what HBM needs to stream
      const int64_t G = (int64_t)gridDim.x * rows;
      for (int64_t p = (int64_t)blockIdx.x * rows + row; p < M; p += TR_UNROLL * G) {
        f32x4 v[TR_UNROLL];
#pragma unroll
        for (int u = 0; u < TR_UNROLL; u++)
          v[u] = p + u * G < M ? *reinterpret_cast<const f32x4*>(z + (p + u * G) * C + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < TR_UNROLL; u++)
#pragma unroll
          for (int k = 0; k < 4; k++) {
            s[k] += (double)v[u][k];
            ss[k] += (double)v[u][k] * (double)v[u][k];
          }
      }
    }
    // fold the rows of this block through LDS
    __syncthreads();
    if (q < c4n && row < rows)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sh[((row * lanes + col) * 4 + k) * 2] = s[k];
        sh[((row * lanes + col) * 4 + k) * 2 + 1] = ss[k];
      }
    __syncthreads();
    if (row == 0 && q < c4n) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        double a = 0, b = 0;
        for (int r = 0; r < rows; r++) {
          a += sh[((r * lanes + col) * 4 + k) * 2];
          b += sh[((r * lanes + col) * 4 + k) * 2 + 1];
        }
        part[((int64_t)blockIdx.x * C + q * 4 + k) * 2] = a;
        part[((int64_t)blockIdx.x * C + q * 4 + k) * 2 + 1] = b;
      }
    }
  }
}

// mean / invstd (biased variance, as normalisation uses) + running-stat update with the
// unbiased variance (torch: running = (1 - m) * running + m * stat)
// one 64-lane wave per channel: lanes stride over the per-workgroup partials, wave reduction
__global__ __launch_bounds__(64) void bn_stats_finalize_kernel(const double* __restrict__ part, int nblocks, int64_t M,
                                                               int C, float eps, float momentum,
                                                               float* __restrict__ mean, float* __restrict__ invstd,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var) {
  const int c = blockIdx.x;
  double s = 0, ss = 0;
  for (int b = threadIdx.x; b < nblocks; b += 64) {
    s += part[((int64_t)b * C + c) * 2];
    ss += part[((int64_t)b * C + c) * 2 + 1];
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (threadIdx.x != 0) return;
  const double mu = s / (double)M;
  double var = ss / (double)M - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// The same from the partials the forward conv's epilogue kept (ConvArgs.bn_part, conv_common.h): part[c][tiles][2]
// float64 (sum, sum of squares) per (channel, conv workgroup).  One 256-thread workgroup per channel, contiguous reads.
__global__ __launch_bounds__(256) void bn_stats_finalize_tiles_kernel(const double* __restrict__ part, int tiles, int64_t M,
                                                                      float eps, float momentum, float* __restrict__ mean,
                                                                      float* __restrict__ invstd,
                                                                      float* __restrict__ running_mean,
                                                                      float* __restrict__ running_var) {
  __shared__ double red[2][4];
  const int c = blockIdx.x;
  const double* p = part + (int64_t)c * tiles * 2;
  double s = 0, ss = 0;
  for (int b = threadIdx.x; b < tiles; b += 256) {
    s += p[2 * b];
    ss += p[2 * b + 1];
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = ss;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double mu = s / (double)M;
  double var = ss / (double)M - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

extern "C" int mval_bn_finalize_stats(const double* part, int tiles, int64_t M, int C, float eps, float momentum, float* mean,
                                      float* invstd, float* running_mean, float* running_var, void* stream) {
  MVAL_REQUIRE(part && tiles > 0 && M > 0 && C > 0 && mean && invstd, "mval_bn_finalize_stats: bad arguments");
  hipLaunchKernelGGL(bn_stats_finalize_tiles_kernel, dim3(C), dim3(256), 0, mval_stream(stream), part, tiles, M, eps, momentum,
                     mean, invstd, running_mean, running_var);
  MVAL_CHECK_LAUNCH("mval_bn_finalize_stats");
  return 0;
}

extern "C" int mval_bn_batch_stats(const float* z, int64_t M, int C, float eps, float momentum, float* mean,
                                   float* invstd, float* running_mean, float* running_var, double* ws, void* stream) {
  MVAL_REQUIRE(M > 0 && C > 0 && (C & 3) == 0, "mval_bn_batch_stats: bad dims (C must be a multiple of 4)");
  const int c4n = C >> 2;
  const int lanes = c4n < 256 ? c4n : 256;
  const int rows = 256 / lanes;
  int nb = (int)((M + rows - 1) / rows);
  if (nb > TR_BLOCKS) nb = TR_BLOCKS;
  size_t sh = (size_t)rows * lanes * 4 * 2 * sizeof(double);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(256), sh, mval_stream(stream), z, ws, M, C);
  MVAL_CHECK_LAUNCH("mval_bn_batch_stats/partial");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(64), 0, mval_stream(stream), ws, nb, M, C, eps, momentum,
                     mean, invstd, running_mean, running_var);
  MVAL_CHECK_LAUNCH("mval_bn_batch_stats/finalize");
  return 0;
}

// ---- forward apply: out = act(((z*alpha + beta') up) + res1 + res2) ------------------------
__global__ __launch_bounds__(TR_APPLY_THREADS) void bn_apply_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ res1,
                                                           const float* __restrict__ res2, float* __restrict__ out,
                                                           int N, int H, int W, int C, int up, int relu,
                                                           unsigned* __restrict__ amax_row, unsigned char* __restrict__ relu_mask) {
  const int c4n = C >> 2;
  const int Ho = H << up, Wo = W << up;
  const int64_t total = (int64_t)N * Ho * Wo * c4n;
  float amax = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * TR_APPLY_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * TR_APPLY_THREADS) {
    const int q = (int)(t % c4n);
    int64_t p = t / c4n;
    const int X = (int)(p % Wo);
    const int Y = (int)((p / Wo) % Ho);
    const int n = (int)(p / ((int64_t)Wo * Ho));
    const int64_t zi = (((int64_t)n * H + (Y >> up)) * W + (X >> up)) * C + q * 4;
    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + zi);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + q * 4);
    f32x4 r = bn_affine(zv, mu, is, g, b);  // torch: alpha = invstd * weight ; beta' = bias - mean * alpha
    const int64_t o = p * C + q * 4;
    if (res1) r += *reinterpret_cast<const f32x4*>(res1 + o);
    if (res2) r += *reinterpret_cast<const f32x4*>(res2 + o);
    if (relu) {
      r.x = mval_relu(r.x); r.y = mval_relu(r.y); r.z = mval_relu(r.z); r.w = mval_relu(r.w);
    }
    *reinterpret_cast<f32x4*>(out + o) = r;
    // (out > 0) of the float4 as four bits of one byte: what the backward pass of a ReLU behind residual adds needs of
    // `out` -- a sixteenth of the bytes (mval_bn_bwd_fused, mask_mode 3)
    if (relu_mask) relu_mask[o >> 2] = (unsigned char)((r.x > 0.f) | ((r.y > 0.f) << 1) | ((r.z > 0.f) << 2) | ((r.w > 0.f) << 3));
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
  }
  if (amax_row) tr_amax_store(amax_row, amax);
}

extern "C" int mval_bn_apply_fwd_amax(const float* z, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, const float* res1, const float* res2, float* out, int N, int H,
                                      int W, int C, int up, int relu, uint32_t* amax_row, void* stream) {
  return mval_bn_apply_fwd_mask(z, mean, invstd, gamma, beta, res1, res2, out, N, H, W, C, up, relu, amax_row, nullptr, stream);
}
extern "C" int mval_bn_apply_fwd_mask(const float* z, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, const float* res1, const float* res2, float* out, int N, int H,
                                      int W, int C, int up, int relu, uint32_t* amax_row, uint8_t* relu_mask, void* stream) {
  MVAL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && up >= 0, "mval_bn_apply_fwd: bad dims");
  int64_t total = (int64_t)N * (H << up) * (W << up) * (C >> 2);
  int nb = (int)((total + TR_APPLY_THREADS - 1) / TR_APPLY_THREADS);
  if (nb > TR_APPLY_BLOCKS) nb = TR_APPLY_BLOCKS;
  hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(nb), dim3(TR_APPLY_THREADS), 0, mval_stream(stream), z, mean, invstd, gamma, beta,
                     res1, res2, out, N, H, W, C, up, relu, amax_row, relu_mask);
  MVAL_CHECK_LAUNCH("mval_bn_apply_fwd");
  return 0;
}
extern "C" int mval_bn_apply_fwd(const float* z, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, const float* res1, const float* res2, float* out, int N, int H,
                                 int W, int C, int up, int relu, void* stream) {
  return mval_bn_apply_fwd_amax(z, mean, invstd, gamma, beta, res1, res2, out, N, H, W, C, up, relu, nullptr, stream);
}

// ---- NHWC float4 -> P2 granules without LDS (round 4) -------------------------------------------------------------------------
// A wave covers PXW pixels x CB channels (CB = 32: 8 pixels x 8 float4; 16: 16 x 4; 8: 32 x 2): lane = (pixel lane / CQ, channel quad
// lane % CQ).  The NHWC side reads / writes CB * 4 contiguous bytes per pixel (full 128-byte lines at CB = 32).  A P2 granule is 8
// channels = the float4s of an EVEN lane and its odd neighbour: the pair swaps halves (one DPP quad permute per dword) so that the even
// lane holds the whole h granule and the odd lane the whole l granule -- each lane stores ONE 16-byte granule, PXW consecutive pixels per
// (plane, 8-channel block) and store instruction.
struct P2WaveMap {
  int CQ, PXW, NCB;  // float4 per pixel and channel block, pixels per wave, channel blocks
};
__device__ __forceinline__ P2WaveMap p2_wave_map(int C) {
  const int CB = (C & 31) == 0 ? 32 : (C & 15) == 0 ? 16 : 8;
  P2WaveMap m;
  m.CQ = CB >> 2;
  m.PXW = 64 / m.CQ;
  m.NCB = C / CB;
  return m;
}
// this lane's granule of the pair (even lane: [own h | neighbour's h] -> plane h; odd lane: [neighbour's l | own l] -> plane l)
__device__ __forceinline__ p2_u32x4 p2_pair_granule(const p2_f16x4 h, const p2_f16x4 l, bool odd) {
  const p2_u32x2 hu = __builtin_bit_cast(p2_u32x2, h), lu = __builtin_bit_cast(p2_u32x2, l);
  const unsigned s0 = odd ? hu.x : lu.x, s1 = odd ? hu.y : lu.y;  // what the neighbour needs of this lane
  const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
  const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xf, 0xf, false);
  return odd ? (p2_u32x4){r0, r1, lu.x, lu.y} : (p2_u32x4){hu.x, hu.y, r0, r1};
}

// ---- forward apply that ALSO writes the activation as P2 planes (round 4: the training forward's convs on conv_p2.hip) ----------
// out = act(((bn(z) up) + res1) + res2) as fp32 NHWC (residual consumers, weight gradients) AND as the fp16 plane pairs
// [n][plane][C/8][Ho][Wo][8] the P2 convs stage by copy.  The P2 scale must exist before the first store: it comes from a rigorous
// a-priori bound of train-mode BatchNorm's output -- Samuelson's inequality: |z_i - mean| <= std * sqrt(M - 1) over the M samples of
// a channel, so |bn(z)| <= |gamma| sqrt(M - 1) + |beta| -- plus the exact maxima of the residuals (their magnitude rows).  On the
// BASELINE shapes the bound sits 2^4 .. 2^7 above the actual maximum (|xhat| reaches 4 - 6 of sqrt(M - 1) = 90 .. 724): inside the
// range where the pair keeps all 22 bits (conv_p2.h).  One scale for the tensor (BatchNorm couples the batch anyway): every image's row
// gets the same 2^-s.  Thread = 8 channels of one pixel, sixteen consecutive pixels per 8-channel block and half-wave quarter: 128-byte
// NHWC reads, 256-byte P2 stores.
// the other direction: the granule this lane loaded of a RESIDUAL's planes (even lane: plane h of the pair's 8 channels, odd lane: plane l)
// -> the (h, l) halves of this lane's own four channels
__device__ __forceinline__ f32x4 p2_pair_ungranule(const p2_u32x4 g, bool odd) {
  const unsigned s0 = odd ? g.x : g.z, s1 = odd ? g.y : g.w;  // what the neighbour needs of this lane's granule
  const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
  const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xf, 0xf, false);
  const p2_u32x2 h = odd ? (p2_u32x2){r0, r1} : (p2_u32x2){g.x, g.y}, l = odd ? (p2_u32x2){g.z, g.w} : (p2_u32x2){r0, r1};
  return p2_join(__builtin_bit_cast(p2_f16x4, h), __builtin_bit_cast(p2_f16x4, l));
}

// res1_p2 / res2_p2 (round 4, late): a residual read from ITS P2 planes (same shape as the output; rows: its 2^-s in the scale slot)
// instead of an fp32 NHWC copy -- block outputs whose every reader takes the planes are then never written as fp32.
__global__ __launch_bounds__(TR_APPLY_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void bn_apply_fwd_p2_kernel(
    const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res1, const float* __restrict__ res2, float* __restrict__ out,
    _Float16* __restrict__ planes, unsigned* __restrict__ p2_rows, int N, int H, int W, int C, int up, int relu,
    unsigned* __restrict__ amax_row, unsigned char* __restrict__ relu_mask, const unsigned* __restrict__ res1_row,
    const unsigned* __restrict__ res2_row, float sqrt_m1, const _Float16* __restrict__ res1_p2, const unsigned* __restrict__ res1_p2_rows,
    const _Float16* __restrict__ res2_p2, const unsigned* __restrict__ res2_p2_rows) {
  const int C8 = C >> 3;
  const int Ho = H << up, Wo = W << up;
  const int HWo = Ho * Wo;
  const int64_t npx = (int64_t)N * HWo;
  // the tensor's scale (every wave computes it: a few loads per lane)
  float bnd = 0.f;
  for (int c = threadIdx.x & 63; c < C; c += 64) bnd = fmaxf(bnd, __builtin_fmaf(fabsf(gamma[c]), sqrt_m1, fabsf(beta[c])));
  bnd = __uint_as_float(p2_wave_umax(__float_as_uint(bnd))) * (1.f + 1e-6f);
  if (res1_row) bnd += __uint_as_float(conv_amax_read(res1_row));
  if (res2_row) bnd += __uint_as_float(conv_amax_read(res2_row));
  float out_mul, out_inv;
  p2_scale_of(bnd, out_mul, out_inv);
  if (blockIdx.x == 0)
    for (int n = threadIdx.x; n < N; n += TR_APPLY_THREADS) p2_rows[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);
  const P2WaveMap wm = p2_wave_map(C);
  const int lane = threadIdx.x & 63;
  const int p_lo = lane / wm.CQ, q_lo = lane - p_lo * wm.CQ;
  const bool odd = q_lo & 1;
  const int64_t nchunks = (npx + wm.PXW - 1) / wm.PXW;           // pixel chunks of a wave
  const int64_t nwork = nchunks * wm.NCB;                          // wave work items: (pixel chunk, channel block)
  const int64_t plane_halves = (int64_t)C8 * HWo * 8;
  const int64_t wave0 = ((int64_t)blockIdx.x * TR_APPLY_THREADS + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * TR_APPLY_THREADS) >> 6;
  float amax = 0.f;
  const float r1_inv = res1_p2 ? __uint_as_float(res1_p2_rows[P2_INV_SLOT]) : 0.f, r2_inv = res2_p2 ? __uint_as_float(res2_p2_rows[P2_INV_SLOT]) : 0.f;
  for (int64_t wi = wave0; wi < nwork; wi += nwaves) {
    const int64_t pchunk = wi / wm.NCB;
    const int cb = (int)(wi - pchunk * wm.NCB);
    const int64_t pix = pchunk * wm.PXW + p_lo;
    const int q = cb * wm.CQ + q_lo;
    const bool ok = pix < npx;
    f32x4 rr = (f32x4){0.f, 0.f, 0.f, 0.f};
    int n = 0, pin = 0;
    if (ok) {
      n = (int)(pix / HWo);
      pin = (int)(pix - (int64_t)n * HWo);
    }
    // residuals kept as planes: this lane's granule (the exchange with its neighbour takes every lane of the wave)
    const int64_t gaddr = (((int64_t)n * 2 * C8 + (q >> 1)) * HWo + pin) * 8 + (odd ? plane_halves : 0);
    const p2_u32x4 zero4 = {0u, 0u, 0u, 0u};
    f32x4 rp1 = (f32x4){0.f, 0.f, 0.f, 0.f}, rp2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (res1_p2) rp1 = p2_pair_ungranule(ok ? *reinterpret_cast<const p2_u32x4*>(res1_p2 + gaddr) : zero4, odd) * r1_inv;
    if (res2_p2) rp2 = p2_pair_ungranule(ok ? *reinterpret_cast<const p2_u32x4*>(res2_p2 + gaddr) : zero4, odd) * r2_inv;
    if (ok) {
      const int Y = pin / Wo, X = pin - Y * Wo;
      const int64_t zi = (((int64_t)n * H + (Y >> up)) * W + (X >> up)) * C + q * 4;
      const int64_t o = pix * C + q * 4;
            // (z is not needed again before the backward pass: a non-temporal load leaves the caches to the planes the next conv reads;
      // with the same hint on the backward apply's LAST reads of the output gradient, z and the mask bytes: C3 61.28 -> 60.81 ms, three
      // runs each on one box, profiles/r05/bn_nt_z.log)
      const f32x4 zv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + zi));
      const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
      const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + q * 4);
      rr = bn_affine(zv, mu, is, g, b);
      if (res1) rr += *reinterpret_cast<const f32x4*>(res1 + o);
      else if (res1_p2) rr += rp1;
      if (res2) rr += *reinterpret_cast<const f32x4*>(res2 + o);
      else if (res2_p2) rr += rp2;
      if (relu) {
        rr.x = mval_relu(rr.x); rr.y = mval_relu(rr.y); rr.z = mval_relu(rr.z); rr.w = mval_relu(rr.w);
      }
      if (out) *reinterpret_cast<f32x4*>(out + o) = rr;
      if (relu_mask) relu_mask[o >> 2] = (unsigned char)((rr.x > 0.f) | ((rr.y > 0.f) << 1) | ((rr.z > 0.f) << 2) | ((rr.w > 0.f) << 3));
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(rr.x), fabsf(rr.y))), fmaxf(fabsf(rr.z), fabsf(rr.w)));
    }
    p2_f16x4 h, l;
    p2_split(rr * out_mul, h, l);
    const p2_u32x4 gran = p2_pair_granule(h, l, odd);  // (every lane of the wave takes part in the exchange)
    if (ok) *reinterpret_cast<p2_u32x4*>(planes + (((int64_t)n * 2 * C8 + (q >> 1)) * HWo + pin) * 8 + (odd ? plane_halves : 0)) = gran;
  }
  if (amax_row) tr_amax_store(amax_row, amax);
}

#ifdef MVAL_TRAIN_ABLATE
int g_train_ablate = 0;  // bit 0: skip the forward apply of residual-free P2-only ops (net_train.hip); bit 1: skip the backward reduction of residual-free ops
#endif
extern "C" int mval_bn_apply_fwd_p2_res(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                        const float* res1, const float* res2, float* out, void* p2_planes, uint32_t* p2_rows, int N, int H,
                                        int W, int C, int up, int relu, uint32_t* amax_row, uint8_t* relu_mask, const uint32_t* res1_row,
                                        const uint32_t* res2_row, const void* res1_p2, const uint32_t* res1_p2_rows, const void* res2_p2,
                                        const uint32_t* res2_p2_rows, void* stream) {
  MVAL_REQUIRE(z && p2_planes && p2_rows && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && up >= 0, "mval_bn_apply_fwd_p2: bad arguments (C % 8)");
  MVAL_REQUIRE(((!res1 && !res1_p2) || res1_row) && ((!res2 && !res2_p2) || res2_row), "mval_bn_apply_fwd_p2: a residual needs its magnitude row (the P2 scale is a bound)");
  MVAL_REQUIRE(!(res1 && res1_p2) && !(res2 && res2_p2) && (!res1_p2 || res1_p2_rows) && (!res2_p2 || res2_p2_rows),
               "mval_bn_apply_fwd_p2: a residual comes either as fp32 NHWC or as planes with their rows");
  MVAL_REQUIRE((int64_t)N * (H << up) * (W << up) * C < ((int64_t)1 << 31), "mval_bn_apply_fwd_p2: tensor too large");
  MVAL_REQUIRE(C <= 4096, "mval_bn_apply_fwd_p2: more than 4096 channels");
  const int64_t total4 = (int64_t)N * (H << up) * (W << up) * (C >> 2);
  int nb = (int)((total4 + TR_APPLY_THREADS - 1) / TR_APPLY_THREADS);
  if (nb > TR_APPLY_BLOCKS) nb = TR_APPLY_BLOCKS;
  const double M = (double)N * H * W;
  hipLaunchKernelGGL(bn_apply_fwd_p2_kernel, dim3(nb), dim3(TR_APPLY_THREADS), 0, mval_stream(stream), z, mean, invstd, gamma, beta, res1,
                     res2, out, reinterpret_cast<_Float16*>(p2_planes), p2_rows, N, H, W, C, up, relu, amax_row, relu_mask, res1_row, res2_row,
                     (float)sqrt(M > 1 ? M - 1.0 : 1.0), reinterpret_cast<const _Float16*>(res1_p2), res1_p2_rows,
                     reinterpret_cast<const _Float16*>(res2_p2), res2_p2_rows);
  MVAL_CHECK_LAUNCH("mval_bn_apply_fwd_p2");
  return 0;
}

extern "C" int mval_bn_apply_fwd_p2(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    const float* res1, const float* res2, float* out, void* p2_planes, uint32_t* p2_rows, int N, int H,
                                    int W, int C, int up, int relu, uint32_t* amax_row, uint8_t* relu_mask, const uint32_t* res1_row,
                                    const uint32_t* res2_row, void* stream) {
  return mval_bn_apply_fwd_p2_res(z, mean, invstd, gamma, beta, res1, res2, out, p2_planes, p2_rows, N, H, W, C, up, relu, amax_row, relu_mask,
                                  res1_row, res2_row, nullptr, nullptr, nullptr, nullptr, stream);
}

// ---- backward, stage 1 ------------------------------------------------------------------
// g_hi = gout * (out > 0 if relu) ; gres1 += g_hi ; gres2 += g_hi ; gz = window-sum(g_hi) at the
// conv resolution ; per-channel partial sums of gz and gz * xhat (xhat = (z - mean) * invstd).
// has_bn == 0: only the masking / residual / window-sum part (gz is then dz itself) and the
// per-channel sum of gz (bias gradient).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ gout,
                                                            const float* __restrict__ out,
                                                            const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ gres1,
                                                            float* __restrict__ gres2, float* __restrict__ gz,
                                                            double* __restrict__ part, int N, int H, int W, int C, int up,
                                                            int relu, int has_bn, int overwrite) {
  extern __shared__ double sh[];
  const int c4n = C >> 2;
  const int lanes = min(c4n, 256);
  const int rows = 256 / lanes;
  const int col = threadIdx.x % lanes, row = threadIdx.x / lanes;
  const int64_t M = (int64_t)N * H * W;
  const int rep = 1 << up;
  const int Ho = H << up, Wo = W << up;
  for (int cb = 0; cb < c4n; cb += lanes) {
    const int q = cb + col;
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    if (q < c4n && row < rows) {
      f32x4 mu = (f32x4){0, 0, 0, 0}, is = (f32x4){0, 0, 0, 0};
      if (has_bn) {
        mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
      }
      const int G = gridDim.x * rows;
      if (up == 0) {
        // plain stream: TR_EW pixels per pass, every load issued before the first use
        for (int p0 = blockIdx.x * rows + row; p0 < (int)M; p0 += TR_EW * G) {
          f32x4 g[TR_EW], ov[TR_EW], zv[TR_EW], a1[TR_EW], a2[TR_EW];
          const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < TR_EW; u++) {
            const int p = p0 + u * G;
            const bool ok = p < (int)M;
            const int64_t o = (int64_t)p * C + q * 4;
            g[u] = ok ? *reinterpret_cast<const f32x4*>(gout + o) : zero;
            ov[u] = (ok && relu) ? *reinterpret_cast<const f32x4*>(out + o) : zero;
            zv[u] = (ok && has_bn) ? *reinterpret_cast<const f32x4*>(z + o) : zero;
            a1[u] = (ok && gres1 && !(overwrite & 1)) ? *reinterpret_cast<const f32x4*>(gres1 + o) : zero;
            a2[u] = (ok && gres2 && !(overwrite & 2)) ? *reinterpret_cast<const f32x4*>(gres2 + o) : zero;
          }
#pragma unroll
          for (int u = 0; u < TR_EW; u++) {
            const int p = p0 + u * G;
            if (p >= (int)M) break;
            const int64_t o = (int64_t)p * C + q * 4;
            f32x4 gv = g[u];
            if (relu) {
              gv.x = ov[u].x > 0.f ? gv.x : 0.f; gv.y = ov[u].y > 0.f ? gv.y : 0.f;
              gv.z = ov[u].z > 0.f ? gv.z : 0.f; gv.w = ov[u].w > 0.f ? gv.w : 0.f;
            }
            if (gres1) *reinterpret_cast<f32x4*>(gres1 + o) = a1[u] + gv;
            if (gres2) *reinterpret_cast<f32x4*>(gres2 + o) = a2[u] + gv;
            *reinterpret_cast<f32x4*>(gz + o) = gv;
            const f32x4 xh = has_bn ? (zv[u] - mu) * is : zero;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              sb[k] += (double)gv[k];
              sg[k] += (double)gv[k] * (double)xh[k];
            }
          }
        }
      } else {
        for (int p = blockIdx.x * rows + row; p < (int)M; p += G) {
          const int x = p % W;
          const int y = (p / W) % H;
          const int n = p / (W * H);
          f32x4 acc = (f32x4){0, 0, 0, 0};
          if (!relu) {
            // the fuse layers' up-sampled terms (no activation: all 28 such ops of HRNet-W32): the 4 / 16 / 64 replicas four at a
            // time -- their loads (gout, and the residual gradients this op adds to) are all requested before the first use --
            // added in the same (dy, dx) order as the general loop below.  One load per round trip left the 67 MB read of gout at
            // 1.4 TB/s (48.9 us per op).
            const int64_t o0 = (((int64_t)n * Ho + (y << up)) * Wo + (x << up)) * C + q * 4;
            const bool acc1 = gres1 && !(overwrite & 1), acc2 = gres2 && !(overwrite & 2);
            const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int r0 = 0; r0 < rep * rep; r0 += 4) {
              f32x4 g4[4], a1[4], a2[4];
              int64_t o[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int r = r0 + u, dy = r >> up, dx = r & (rep - 1);
                o[u] = o0 + ((int64_t)dy * Wo + dx) * C;
                g4[u] = *reinterpret_cast<const f32x4*>(gout + o[u]);
                a1[u] = acc1 ? *reinterpret_cast<const f32x4*>(gres1 + o[u]) : zero;
                a2[u] = acc2 ? *reinterpret_cast<const f32x4*>(gres2 + o[u]) : zero;
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if (gres1) *reinterpret_cast<f32x4*>(gres1 + o[u]) = a1[u] + g4[u];
                if (gres2) *reinterpret_cast<f32x4*>(gres2 + o[u]) = a2[u] + g4[u];
                acc += g4[u];
              }
            }
          } else
          for (int dy = 0; dy < rep; dy++)
            for (int dx = 0; dx < rep; dx++) {
              const int64_t o = (((int64_t)n * Ho + (y << up) + dy) * Wo + (x << up) + dx) * C + q * 4;
              f32x4 g = *reinterpret_cast<const f32x4*>(gout + o);
              if (relu) {
                const f32x4 ov = *reinterpret_cast<const f32x4*>(out + o);
                g.x = ov.x > 0.f ? g.x : 0.f; g.y = ov.y > 0.f ? g.y : 0.f;
                g.z = ov.z > 0.f ? g.z : 0.f; g.w = ov.w > 0.f ? g.w : 0.f;
              }
              if (gres1) {
                if (overwrite & 1) *reinterpret_cast<f32x4*>(gres1 + o) = g;
                else *reinterpret_cast<f32x4*>(gres1 + o) += g;
              }
              if (gres2) {
                if (overwrite & 2) *reinterpret_cast<f32x4*>(gres2 + o) = g;
                else *reinterpret_cast<f32x4*>(gres2 + o) += g;
              }
              acc += g;
            }
          *reinterpret_cast<f32x4*>(gz + (int64_t)p * C + q * 4) = acc;
          f32x4 xh = (f32x4){0, 0, 0, 0};
          if (has_bn) xh = (*reinterpret_cast<const f32x4*>(z + (int64_t)p * C + q * 4) - mu) * is;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            sb[k] += (double)acc[k];
            sg[k] += (double)acc[k] * (double)xh[k];
          }
        }
      }
    }
    __syncthreads();
    if (q < c4n && row < rows)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sh[((row * lanes + col) * 4 + k) * 2] = sb[k];
        sh[((row * lanes + col) * 4 + k) * 2 + 1] = sg[k];
      }
    __syncthreads();
    if (row == 0 && q < c4n) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        double a = 0, b = 0;
        for (int r = 0; r < rows; r++) {
          a += sh[((r * lanes + col) * 4 + k) * 2];
          b += sh[((r * lanes + col) * 4 + k) * 2 + 1];
        }
        part[((int64_t)blockIdx.x * C + q * 4 + k) * 2] = a;
        part[((int64_t)blockIdx.x * C + q * 4 + k) * 2 + 1] = b;
      }
    }
  }
}

// dbeta / dgamma (float, also kept in `sums` for stage 2)
// The same, and this channel's share of an upper bound of |dz| = |gamma invstd (g - dbeta / M - xhat dgamma / M)|:
//   |gamma invstd| (max|g| + |dbeta| / M + sqrt(M - 1) |dgamma| / M)      (|xhat| <= sqrt(M - 1): Samuelson)
// folded into *bound_slot with atomicMax (non-negative float bits order like integers) -- the P2 scale of dz.
// (256 threads per channel: the partials of a channel sit C * 16 bytes apart, one cache line per lane and trip -- with 64 threads the
// ~1 000 partials of a full-size reduction were 16 dependent round trips, 5.8 us per launch, 293 launches per step)
__global__ __launch_bounds__(256) void bn_bwd_finalize_bound_kernel(const double* __restrict__ part, int nblocks, int C,
                                                                    float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                                    float* __restrict__ sums, const float* __restrict__ gmax_part,
                                                                    const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                                    float inv_m, float sqrt_m1, unsigned* __restrict__ bound_slot) {
  __shared__ double red[2][4];
  __shared__ float redm[4];
  const int c = blockIdx.x;
  double a = 0, b = 0;
  float gm = 0.f;
  for (int k = threadIdx.x; k < nblocks; k += 256) {
    a += part[((int64_t)k * C + c) * 2];
    b += part[((int64_t)k * C + c) * 2 + 1];
    gm = fmaxf(gm, gmax_part[k]);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  gm = wave_max(gm);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    redm[threadIdx.x >> 6] = gm;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  gm = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  dbeta[c] = (float)a;
  dgamma[c] = (float)b;
  sums[c] = (float)a;
  sums[C + c] = (float)b;
  const float bd = fabsf(gamma[c] * invstd[c]) * (gm + fabsf((float)a) * inv_m + sqrt_m1 * fabsf((float)b) * inv_m) * (1.f + 1e-5f);
  atomicMax(bound_slot, __float_as_uint(bd));
}

// The same from the partials a data-gradient conv's epilogue left (conv_p2.h P2Args::bs_z): sums as above, the maxima per (slot, channel)
// -- the bound then uses the CHANNEL's own max |masked gradient| (at most the tensor's: still a bound, and a tighter one)
__global__ __launch_bounds__(256) void bn_bwd_finalize_bound_c_kernel(const double* __restrict__ part, const float* __restrict__ gmaxc, int nslots, int C,
                                                                      float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                                      float* __restrict__ sums, const float* __restrict__ gamma,
                                                                      const float* __restrict__ invstd, float inv_m, float sqrt_m1,
                                                                      unsigned* __restrict__ bound_slot) {
  __shared__ double red[2][4];
  __shared__ float redm[4];
  const int c = blockIdx.x;
  double a = 0, b = 0;
  float gm = 0.f;
  for (int k = threadIdx.x; k < nslots; k += 256) {
    a += part[((int64_t)k * C + c) * 2];
    b += part[((int64_t)k * C + c) * 2 + 1];
    gm = fmaxf(gm, gmaxc[(int64_t)k * C + c]);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  gm = wave_max(gm);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    redm[threadIdx.x >> 6] = gm;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  gm = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  dbeta[c] = (float)a;
  dgamma[c] = (float)b;
  sums[c] = (float)a;
  sums[C + c] = (float)b;
  const float bd = fabsf(gamma[c] * invstd[c]) * (gm + fabsf((float)a) * inv_m + sqrt_m1 * fabsf((float)b) * inv_m) * (1.f + 1e-5f);
  atomicMax(bound_slot, __float_as_uint(bd));
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nblocks, int C,
                                                              float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                              float* __restrict__ sums) {
  __shared__ double red[2][4];
  const int c = blockIdx.x;
  double a = 0, b = 0;
  for (int k = threadIdx.x; k < nblocks; k += 256) {
    a += part[((int64_t)k * C + c) * 2];
    b += part[((int64_t)k * C + c) * 2 + 1];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  if (dbeta) dbeta[c] = (float)a;
  if (dgamma) dgamma[c] = (float)b;
  if (sums) {
    sums[c] = (float)a;
    sums[C + c] = (float)b;
  }
}

// stage 2 (in place on gz): dz = gamma * invstd * (gz - dbeta / M - xhat * dgamma / M)
__global__ __launch_bounds__(TR_APPLY_THREADS) void bn_bwd_apply_kernel(float* __restrict__ gz, const float* __restrict__ z,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, int64_t M, int C,
                                                           unsigned* __restrict__ amax_row) {
  const int c4n = C >> 2;
  const int64_t total = M * c4n;
  const float invM = 1.0f / (float)M;
  float amax = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * TR_APPLY_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * TR_APPLY_THREADS) {
    const int q = (int)(t % c4n);
    const int64_t i = (t / c4n) * C + q * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
    const f32x4 db = *reinterpret_cast<const f32x4*>(sums + q * 4);
    const f32x4 dg = *reinterpret_cast<const f32x4*>(sums + C + q * 4);
    const f32x4 xh = (*reinterpret_cast<const f32x4*>(z + i) - mu) * is;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gz + i);
    const f32x4 r = (g * is) * (gv - db * invM - xh * (dg * invM));
    *reinterpret_cast<f32x4*>(gz + i) = r;
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
  }
  if (amax_row) tr_amax_store(amax_row, amax);
}

// channel counts that are not a multiple of 4 (the 19-joint heat-map layer): plain conv with
// bias, no BN / residual / upsample -- copy the (masked) gradient and sum it per channel.
// thread = (pixel row r of the workgroup, channel c): a workgroup reads rows * C contiguous floats per
// pass (coalesced), partial sums per (workgroup, channel) go through the two-stage reduction.
__global__ __launch_bounds__(256) void bias_bwd_scalar_kernel(const float* __restrict__ gout,
                                                              const float* __restrict__ out, float* __restrict__ gz,
                                                              double* __restrict__ part, int64_t M, int C, int relu) {
  __shared__ double red[256];
  const int rows = 256 / C;
  const int r = threadIdx.x / C, c = threadIdx.x % C;
  double acc = 0;
  if (r < rows)
    for (int64_t p = (int64_t)blockIdx.x * rows + r; p < M; p += (int64_t)gridDim.x * rows) {
      float g = gout[p * C + c];
      if (relu && !(out[p * C + c] > 0.f)) g = 0.f;
      gz[p * C + c] = g;
      acc += (double)g;
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    double a = 0;
    for (int k = 0; k < rows; k++) a += red[k * C + threadIdx.x];
    part[((int64_t)blockIdx.x * C + threadIdx.x) * 2] = a;
    part[((int64_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = 0.0;
  }
}

extern "C" int mval_bn_bwd(const float* gout, const float* out, const float* z, const float* mean, const float* invstd,
                           const float* gamma, float* gres1, float* gres2, float* gz, float* dgamma, float* dbeta,
                           double* ws, float* sums, int N, int H, int W, int C, int up, int relu, int has_bn,
                           int overwrite, void* stream) {
  return mval_bn_bwd_amax(gout, out, z, mean, invstd, gamma, gres1, gres2, gz, dgamma, dbeta, ws, sums, N, H, W, C, up, relu,
                          has_bn, overwrite, nullptr, stream);
}

extern "C" int mval_bn_bwd_amax(const float* gout, const float* out, const float* z, const float* mean, const float* invstd,
                                const float* gamma, float* gres1, float* gres2, float* gz, float* dgamma, float* dbeta,
                                double* ws, float* sums, int N, int H, int W, int C, int up, int relu, int has_bn,
                                int overwrite, uint32_t* gz_amax_row, void* stream) {
  MVAL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && up >= 0, "mval_bn_bwd: bad dims");
  MVAL_REQUIRE((int64_t)N * (H << up) * (W << up) * C < ((int64_t)1 << 33), "mval_bn_bwd: more than 2^31 float4 elements");
  MVAL_REQUIRE(!gres1 || gres1 != gres2, "mval_bn_bwd: the two residual gradients must be distinct buffers");
  if (C & 3) {
    MVAL_REQUIRE(!has_bn && up == 0 && !gres1 && !gres2, "mval_bn_bwd: odd channel count only for plain conv+bias");
    MVAL_REQUIRE(C <= 256, "mval_bn_bwd: odd channel count above 256");
    const int64_t Mp = (int64_t)N * H * W;
    const int rows_s = 256 / C;
    int nbs = (int)((Mp + rows_s - 1) / rows_s);
    if (nbs > TR_BLOCKS) nbs = TR_BLOCKS;
    hipLaunchKernelGGL(bias_bwd_scalar_kernel, dim3(nbs), dim3(256), 0, mval_stream(stream), gout, out, gz, ws, Mp, C, relu);
    MVAL_CHECK_LAUNCH("mval_bn_bwd/scalar");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, mval_stream(stream), ws, nbs, C, dbeta, nullptr, nullptr);
    MVAL_CHECK_LAUNCH("mval_bn_bwd/scalar finalize");
    return 0;
  }
  const int64_t M = (int64_t)N * H * W;
  const int c4n = C >> 2;
  const int lanes = c4n < 256 ? c4n : 256;
  const int rows = 256 / lanes;
  int nb = (int)((M + rows - 1) / rows);
  if (nb > TR_BLOCKS) nb = TR_BLOCKS;
  size_t sh = (size_t)rows * lanes * 4 * 2 * sizeof(double);
  hipStream_t s = mval_stream(stream);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), sh, s, gout, out, z, mean, invstd, gres1, gres2, gz, ws,
                     N, H, W, C, up, relu, has_bn, overwrite);
  MVAL_CHECK_LAUNCH("mval_bn_bwd/reduce");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, ws, nb, C, dbeta, has_bn ? dgamma : nullptr, sums);
  MVAL_CHECK_LAUNCH("mval_bn_bwd/finalize");
  if (has_bn) {
    int64_t total = M * c4n;
    int nb2 = (int)((total + TR_APPLY_THREADS - 1) / TR_APPLY_THREADS);
    if (nb2 > TR_APPLY_BLOCKS) nb2 = TR_APPLY_BLOCKS;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb2), dim3(TR_APPLY_THREADS), 0, s, gz, z, mean, invstd, gamma, sums, M, C, gz_amax_row);
    MVAL_CHECK_LAUNCH("mval_bn_bwd/apply");
  }
  return 0;
}

// ---- backward, round 4 form (no upsample, BatchNorm): the masked gradient is NOT written by the reduction -------
// Round 3's pair moved gout -> gz (masked copy) -> gz (dz in place): one write and one read more than needed, and the
// mask always came from a read of `out`.  Here the reduction only reads (gout, z, and `out` when a residual makes the
// mask unrecoverable from z) and scatters the residual gradients; the apply pass re-reads its gradient source -- the
// residual slot this op stored the masked gradient in (first writer), else gout with the mask re-derived -- and writes
// dz once.  mask_mode: 0 none, 1 out > 0, 2 bn_affine(z) > 0 (ReLU without residual: out = max(bn_affine(z), 0)), 3 the
// (out > 0) bits the forward apply kept (one byte per float4; the caller passes it in ov.x's bit pattern).
__device__ __forceinline__ f32x4 bwd_mask(f32x4 g, const int mask_mode, const f32x4 ov, const f32x4 zv, const f32x4 mu,
                                          const f32x4 is, const f32x4 gm, const f32x4 bt) {
  if (mask_mode == 3) {
    const unsigned m = __float_as_uint(ov.x);
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] = ((m >> k) & 1u) ? g[k] : 0.f;
  } else if (mask_mode == 1) {
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] = ov[k] > 0.f ? g[k] : 0.f;
  } else if (mask_mode == 2) {
    const f32x4 r = bn_affine(zv, mu, is, gm, bt);
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] = r[k] > 0.f ? g[k] : 0.f;
  }
  return g;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce2_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                             const float* __restrict__ z, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ gres1, float* __restrict__ gres2,
                                                             double* __restrict__ part, int M, int C, int mask_mode,
                                                             int overwrite, const unsigned char* __restrict__ relu_mask,
                                                             float* __restrict__ gmax_part, unsigned* __restrict__ bound_slot) {
  extern __shared__ double sh[];
  const int c4n = C >> 2;
  const int lanes = min(c4n, 256);
  const int rows = 256 / lanes;
  const int col = threadIdx.x % lanes, row = threadIdx.x / lanes;
  const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gmax = 0.f;  // (gmax_part) max |masked gradient| of this workgroup: the data gradient's P2 scale is a bound built from it
  if (bound_slot && blockIdx.x == 0 && threadIdx.x == 0) *bound_slot = 0u;  // (the finalize kernel folds the channels' bounds into it)
  for (int cb = 0; cb < c4n; cb += lanes) {
    const int q = cb + col;
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    if (q < c4n && row < rows) {
      const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
      const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
      f32x4 gm = zero, bt = zero;
      if (mask_mode == 2) {
        gm = *reinterpret_cast<const f32x4*>(gamma + q * 4);
        bt = *reinterpret_cast<const f32x4*>(beta + q * 4);
      }
      const int G = gridDim.x * rows;
      for (int p0 = blockIdx.x * rows + row; p0 < M; p0 += TR_EW * G) {
        f32x4 g[TR_EW], ov[TR_EW], zv[TR_EW], a1[TR_EW], a2[TR_EW];
#pragma unroll
        for (int u = 0; u < TR_EW; u++) {
          const int p = p0 + u * G;
          const bool ok = p < M;
          const int64_t o = (int64_t)p * C + q * 4;
          g[u] = ok ? *reinterpret_cast<const f32x4*>(gout + o) : zero;
          ov[u] = (ok && mask_mode == 1) ? *reinterpret_cast<const f32x4*>(out + o) : zero;
          if (ok && mask_mode == 3) ov[u].x = __uint_as_float((unsigned)relu_mask[o >> 2]);
          zv[u] = ok ? *reinterpret_cast<const f32x4*>(z + o) : zero;
          a1[u] = (ok && gres1 && !(overwrite & 1)) ? *reinterpret_cast<const f32x4*>(gres1 + o) : zero;
          a2[u] = (ok && gres2 && !(overwrite & 2)) ? *reinterpret_cast<const f32x4*>(gres2 + o) : zero;
        }
#pragma unroll
        for (int u = 0; u < TR_EW; u++) {
          const int p = p0 + u * G;
          if (p >= M) break;
          const int64_t o = (int64_t)p * C + q * 4;
          const f32x4 gv = bwd_mask(g[u], mask_mode, ov[u], zv[u], mu, is, gm, bt);
          gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(gv.x), fabsf(gv.y))), fmaxf(fabsf(gv.z), fabsf(gv.w)));
          if (gres1) *reinterpret_cast<f32x4*>(gres1 + o) = a1[u] + gv;
          if (gres2) *reinterpret_cast<f32x4*>(gres2 + o) = a2[u] + gv;
          const f32x4 xh = (zv[u] - mu) * is;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            sb[k] += (double)gv[k];
            sg[k] += (double)gv[k] * (double)xh[k];
          }
        }
      }
    }
    __syncthreads();
    if (q < c4n && row < rows)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sh[((row * lanes + col) * 4 + k) * 2] = sb[k];
        sh[((row * lanes + col) * 4 + k) * 2 + 1] = sg[k];
      }
    __syncthreads();
    if (row == 0 && q < c4n) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        double a = 0, b = 0;
        for (int r = 0; r < rows; r++) {
          a += sh[((r * lanes + col) * 4 + k) * 2];
          b += sh[((r * lanes + col) * 4 + k) * 2 + 1];
        }
        part[((int64_t)blockIdx.x * C + q * 4 + k) * 2] = a;
        part[((int64_t)blockIdx.x * C + q * 4 + k) * 2 + 1] = b;
      }
    }
  }
  if (gmax_part) {
    __shared__ float gred[4];
    gmax = wave_max(gmax);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) gmax_part[blockIdx.x] = fmaxf(fmaxf(gred[0], gred[1]), fmaxf(gred[2], gred[3]));
  }
}

// dz = gamma * invstd * (g - dbeta / M - xhat * dgamma / M), g = gsrc masked as the reduction masked it
// (gsrc = a residual slot holding the masked gradient: mask_mode 0)
__global__ __launch_bounds__(TR_APPLY_THREADS) void bn_bwd_apply2_kernel(const float* __restrict__ gsrc, const float* __restrict__ out,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ sums,
                                                           float* __restrict__ gz, int64_t M, int C, int mask_mode,
                                                           unsigned* __restrict__ amax_row, const unsigned char* __restrict__ relu_mask) {
  const int c4n = C >> 2;
  const int64_t total = M * c4n;
  const float invM = 1.0f / (float)M;
  const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * TR_APPLY_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * TR_APPLY_THREADS) {
    const int q = (int)(t % c4n);
    const int64_t i = (t / c4n) * C + q * 4;
    const f32x4 gv0 = *reinterpret_cast<const f32x4*>(gsrc + i);
    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + i);
    f32x4 ov = mask_mode == 1 ? *reinterpret_cast<const f32x4*>(out + i) : zero;
    if (mask_mode == 3) ov.x = __uint_as_float((unsigned)relu_mask[i >> 2]);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
    const f32x4 bt = mask_mode == 2 ? *reinterpret_cast<const f32x4*>(beta + q * 4) : zero;
    const f32x4 db = *reinterpret_cast<const f32x4*>(sums + q * 4);
    const f32x4 dg = *reinterpret_cast<const f32x4*>(sums + C + q * 4);
    const f32x4 gv = bwd_mask(gv0, mask_mode, ov, zv, mu, is, g, bt);
    const f32x4 xh = (zv - mu) * is;
    const f32x4 r = (g * is) * (gv - db * invM - xh * (dg * invM));
    *reinterpret_cast<f32x4*>(gz + i) = r;
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
  }
  if (amax_row) tr_amax_store(amax_row, amax);
}

// dz as above, written as fp32 NHWC (gz, optional) AND as P2 planes for the data-gradient conv on the P2 kernels: the float4 side in
// NHWC order, the granules through LDS (as bn_apply_fwd_p2_kernel); the scale from the bound the finalize step left in *bound_slot.
// (MM = the mask mode as a template parameter and 8 waves per SIMD: with the mode at run time the kernel held 70 VGPRs -- 7 waves per
// SIMD, i.e. ONE 1024-thread block per CU with 12 of 28 wave slots empty; bn_apply_fwd_p2_kernel likewise: 36.4 -> 31.9 us per launch)
// SC (round 6, pre-summed reductions of an op WITH residuals): the pass also scatters the masked gradient into the residual slots -- what
// the reduction pass, which did not run, does otherwise (sc_ow bit 0 / 1: store instead of accumulate: first writer of that slot)
template <int MM, bool SC = false>
__global__ __launch_bounds__(TR_APPLY_THREADS) __attribute__((amdgpu_waves_per_eu(SC ? 6 : 8, 8))) void bn_bwd_apply2_p2_kernel(
    const float* __restrict__ gsrc, const float* __restrict__ out, const float* __restrict__ z, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ sums,
    float* __restrict__ gz, _Float16* __restrict__ planes, unsigned* __restrict__ p2_rows, const unsigned* __restrict__ bound_slot, int N,
    int HW, int C, int, unsigned* __restrict__ amax_row, const unsigned char* __restrict__ relu_mask, float* __restrict__ sc_g1 = nullptr,
    float* __restrict__ sc_g2 = nullptr, int sc_ow = 0) {
  const int C8 = C >> 3;
  const int64_t npx = (int64_t)N * HW;
  const float invM = 1.0f / (float)npx;
  float out_mul, out_inv;
  p2_scale_of(__uint_as_float(*bound_slot), out_mul, out_inv);
  if (blockIdx.x == 0)
    for (int n = threadIdx.x; n < N; n += TR_APPLY_THREADS) p2_rows[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);
  const P2WaveMap wm = p2_wave_map(C);
  const int lane = threadIdx.x & 63;
  const int p_lo = lane / wm.CQ, q_lo = lane - p_lo * wm.CQ;
  const bool odd = q_lo & 1;
  const int64_t nchunks = (npx + wm.PXW - 1) / wm.PXW, nwork = nchunks * wm.NCB;
  const int64_t plane_halves = (int64_t)C8 * HW * 8;
  const int64_t wave0 = ((int64_t)blockIdx.x * TR_APPLY_THREADS + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * TR_APPLY_THREADS) >> 6;
  const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
  for (int64_t wi = wave0; wi < nwork; wi += nwaves) {
    const int64_t pchunk = wi / wm.NCB;
    const int cb = (int)(wi - pchunk * wm.NCB);
    const int64_t pix = pchunk * wm.PXW + p_lo;
    const int q = cb * wm.CQ + q_lo;
    const bool ok = pix < npx;
    f32x4 r = zero;
    int n = 0, pin = 0;
    if (ok) {
      n = (int)(pix / HW);
      pin = (int)(pix - (int64_t)n * HW);
      const int64_t o = pix * C + q * 4;
      // (the LAST reads of the output gradient, of z and of the mask bytes: non-temporal, see bn_apply_fwd_p2_kernel)
      const f32x4 gv0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gsrc + o));
      const f32x4 zv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + o));
      f32x4 ov = MM == 1 ? *reinterpret_cast<const f32x4*>(out + o) : zero;
      if (MM == 3) ov.x = __uint_as_float((unsigned)__builtin_nontemporal_load(relu_mask + (o >> 2)));
      const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
      const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
      const f32x4 bt = MM == 2 ? *reinterpret_cast<const f32x4*>(beta + q * 4) : zero;
      const f32x4 db = *reinterpret_cast<const f32x4*>(sums + q * 4);
      const f32x4 dg = *reinterpret_cast<const f32x4*>(sums + C + q * 4);
      const f32x4 gv = bwd_mask(gv0, MM, ov, zv, mu, is, g, bt);
      if constexpr (SC) {
        if (sc_g1) *reinterpret_cast<f32x4*>(sc_g1 + o) = (sc_ow & 1) ? gv : *reinterpret_cast<const f32x4*>(sc_g1 + o) + gv;
        if (sc_g2) *reinterpret_cast<f32x4*>(sc_g2 + o) = (sc_ow & 2) ? gv : *reinterpret_cast<const f32x4*>(sc_g2 + o) + gv;
      }
      const f32x4 xh = (zv - mu) * is;
      r = (g * is) * (gv - db * invM - xh * (dg * invM));
      if (gz) *reinterpret_cast<f32x4*>(gz + o) = r;
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    }
    p2_f16x4 h, l;
    p2_split(r * out_mul, h, l);
    const p2_u32x4 gran = p2_pair_granule(h, l, odd);
    if (ok) *reinterpret_cast<p2_u32x4*>(planes + (((int64_t)n * 2 * C8 + (q >> 1)) * HW + pin) * 8 + (odd ? plane_halves : 0)) = gran;
  }
  if (amax_row) tr_amax_store(amax_row, amax);
}

// Backward of out = act(bn(z) + res1 + res2) at the conv resolution (no upsample): same results as mval_bn_bwd_amax
// (dz in gz, dgamma, dbeta, residual gradients scattered), one tensor write and one to two tensor reads fewer.
extern "C" int mval_bn_bwd_fused(const float* gout, const float* out, const float* z, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, float* gres1, float* gres2, float* gz, float* dgamma,
                                 float* dbeta, double* ws, float* sums, int N, int H, int W, int C, int relu, int overwrite,
                                 uint32_t* gz_amax_row, void* stream) {
  return mval_bn_bwd_fused_mask(gout, out, nullptr, z, mean, invstd, gamma, beta, gres1, gres2, gz, dgamma, dbeta, ws, sums, N, H, W, C, relu,
                                overwrite, gz_amax_row, stream);
}
// The same with the ReLU mask taken from the bytes mval_bn_apply_fwd_mask kept (relu_mask != NULL: `out` is not read at all).
extern "C" int mval_bn_bwd_fused_mask(const float* gout, const float* out, const uint8_t* relu_mask, const float* z, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, float* gres1, float* gres2, float* gz,
                                      float* dgamma, float* dbeta, double* ws, float* sums, int N, int H, int W, int C, int relu,
                                      int overwrite, uint32_t* gz_amax_row, void* stream) {
  return mval_bn_bwd_fused_p2(gout, out, relu_mask, z, mean, invstd, gamma, beta, gres1, gres2, gz, dgamma, dbeta, ws, sums, N, H, W, C, relu,
                              overwrite, gz_amax_row, nullptr, nullptr, nullptr, nullptr, stream);
}
// The same, and (dz_planes != NULL) dz ALSO as P2 planes [n][plane][C/8][H][W][8] with rows dz_rows[n][MVAL_P2_ROW] -- the input of the
// data-gradient conv on the P2 kernels (round 4).  Its scale comes from a bound of |dz| built in the finalize step from max |masked
// gradient| (kept by the reduction, gmax_ws >= 512 floats), dbeta, dgamma and Samuelson's |xhat| <= sqrt(M - 1); bound_slot: one dword of
// scratch.  gz may be NULL then (no fp32 copy of dz).
// (round 6) The NEXT mval_bn_bwd_fused_p2 call finds its reduction done: part[nslots][C][2] and gmaxc[nslots][C] were left by the epilogue
// of the data-gradient conv that wrote gout (its only writer; ReLU, no residual), and that kernel zeroed bound_slot.
static thread_local const double* g_presum_part = nullptr;
static thread_local const float* g_presum_gmax = nullptr;
static thread_local int g_presum_slots = 0;
void mval_bn_bwd_set_presummed(const double* part, const float* gmaxc, int nslots) {
  g_presum_part = part;
  g_presum_gmax = gmaxc;
  g_presum_slots = nslots;
}

extern "C" int mval_bn_bwd_fused_p2(const float* gout, const float* out, const uint8_t* relu_mask, const float* z, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, float* gres1, float* gres2, float* gz,
                                    float* dgamma, float* dbeta, double* ws, float* sums, int N, int H, int W, int C, int relu,
                                    int overwrite, uint32_t* gz_amax_row, void* dz_planes, uint32_t* dz_rows, float* gmax_ws,
                                    uint32_t* bound_slot, void* stream) {
  // (pre-summed partials are consumed by THIS call, whatever it returns: a failed call must not leave them for the next one)
  const double* pre_part = g_presum_part;
  const float* pre_gmax = g_presum_gmax;
  const int pre_slots = g_presum_slots;
  g_presum_part = nullptr;
  g_presum_gmax = nullptr;
  g_presum_slots = 0;
  const bool p2 = dz_planes != nullptr;
  MVAL_REQUIRE(!p2 || (dz_rows && gmax_ws && bound_slot && (C & 7) == 0 && dgamma && dbeta), "mval_bn_bwd_fused_p2: the P2 form needs rows, scratch and C % 8 == 0");
  MVAL_REQUIRE(p2 || gz, "mval_bn_bwd_fused: no output for dz");
  MVAL_REQUIRE(gout && z && mean && invstd && gamma && beta && ws && sums && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0,
               "mval_bn_bwd_fused: bad arguments (C must be a multiple of 4)");
  MVAL_REQUIRE((int64_t)N * H * W * C < ((int64_t)1 << 33) && (int64_t)N * H * W < ((int64_t)1 << 31), "mval_bn_bwd_fused: more than 2^31 float4 elements");
  MVAL_REQUIRE(!gres1 || gres1 != gres2, "mval_bn_bwd_fused: the two residual gradients must be distinct buffers");
  MVAL_REQUIRE(!relu || out || relu_mask || (!gres1 && !gres2), "mval_bn_bwd_fused: ReLU with residuals needs the output activation (or its mask bytes)");
  const int64_t M = (int64_t)N * H * W;
  const int c4n = C >> 2;
  const int lanes = c4n < 256 ? c4n : 256;
  const int rows = 256 / lanes;
  int nb = (int)((M + rows - 1) / rows);
  if (nb > TR_BLOCKS) nb = TR_BLOCKS;
  const size_t sh = (size_t)rows * lanes * 4 * 2 * sizeof(double);
  hipStream_t s = mval_stream(stream);
  const int mask_mode = !relu ? 0 : (gres1 || gres2) ? (relu_mask ? 3 : 1) : 2;
  if (pre_part) {
    MVAL_REQUIRE(p2 && (mask_mode == 2 || mask_mode == 3) && pre_gmax && pre_slots > 0,
                 "mval_bn_bwd_fused_p2: pre-summed partials need the P2 form of a ReLU op (mask from z, or from the kept bits with residuals)");
    hipLaunchKernelGGL(bn_bwd_finalize_bound_c_kernel, dim3(C), dim3(256), 0, s, pre_part, pre_gmax, pre_slots, C, dbeta, dgamma, sums, gamma, invstd,
                       1.0f / (float)M, (float)sqrt(M > 1 ? (double)M - 1.0 : 1.0), bound_slot);
    MVAL_CHECK_LAUNCH("mval_bn_bwd_fused/finalize (pre-summed)");
  } else {
#ifdef MVAL_TRAIN_ABLATE  // (measurement build only: the upper bound of "the backward reduction in the producer dgrad's epilogue" -- the sums stay stale)
  if (!(g_train_ablate & 2) || gres1 || gres2)
#endif
  hipLaunchKernelGGL(bn_bwd_reduce2_kernel, dim3(nb), dim3(256), sh, s, gout, out, z, mean, invstd, gamma, beta, gres1, gres2, ws,
                     (int)M, C, mask_mode, overwrite, relu_mask, p2 ? gmax_ws : nullptr, p2 ? bound_slot : nullptr);
  MVAL_CHECK_LAUNCH("mval_bn_bwd_fused/reduce");
  if (p2)
    hipLaunchKernelGGL(bn_bwd_finalize_bound_kernel, dim3(C), dim3(256), 0, s, ws, nb, C, dbeta, dgamma, sums, gmax_ws, gamma, invstd,
                       1.0f / (float)M, (float)sqrt(M > 1 ? (double)M - 1.0 : 1.0), bound_slot);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, ws, nb, C, dbeta, dgamma, sums);
  MVAL_CHECK_LAUNCH("mval_bn_bwd_fused/finalize");
  }
  // the masked gradient again: from the residual slot this op was the first to write (it holds exactly that), else
  // from gout with the mask re-derived
  const float* gsrc = gout;
  int apply_mask = mask_mode;
  if (!pre_part) {
    if (gres1 && (overwrite & 1)) { gsrc = gres1; apply_mask = 0; }
    else if (gres2 && (overwrite & 2)) { gsrc = gres2; apply_mask = 0; }
  }
  const int64_t total = M * c4n;
  int nb2 = (int)((total + TR_APPLY_THREADS - 1) / TR_APPLY_THREADS);
  if (nb2 > TR_APPLY_BLOCKS) nb2 = TR_APPLY_BLOCKS;
  if (p2 && pre_part && (gres1 || gres2)) {
    // (round 6) the reduction pass did not run: this pass masks from the kept bits AND scatters the residual gradients
    hipLaunchKernelGGL((bn_bwd_apply2_p2_kernel<3, true>), dim3(nb2), dim3(TR_APPLY_THREADS), 0, s, gout, out, z, mean, invstd, gamma, beta, sums, gz,
                       reinterpret_cast<_Float16*>(dz_planes), dz_rows, bound_slot, N, H * W, C, 3, gz_amax_row, relu_mask, gres1, gres2, overwrite);
    MVAL_CHECK_LAUNCH("mval_bn_bwd_fused/apply p2 + scatter");
    return 0;
  }
  if (p2) {
#define BWD_P2_LAUNCH(MM_)                                                                                                              \
  hipLaunchKernelGGL(bn_bwd_apply2_p2_kernel<MM_>, dim3(nb2), dim3(TR_APPLY_THREADS), 0, s, gsrc, out, z, mean, invstd, gamma, beta, sums, gz, \
                     reinterpret_cast<_Float16*>(dz_planes), dz_rows, bound_slot, N, H * W, C, apply_mask, gz_amax_row, relu_mask)
    if (apply_mask == 0) BWD_P2_LAUNCH(0);
    else if (apply_mask == 1) BWD_P2_LAUNCH(1);
    else if (apply_mask == 2) BWD_P2_LAUNCH(2);
    else BWD_P2_LAUNCH(3);
#undef BWD_P2_LAUNCH
    MVAL_CHECK_LAUNCH("mval_bn_bwd_fused/apply p2");
    return 0;
  }
  hipLaunchKernelGGL(bn_bwd_apply2_kernel, dim3(nb2), dim3(TR_APPLY_THREADS), 0, s, gsrc, out, z, mean, invstd, gamma, beta, sums, gz, M,
                     C, apply_mask, gz_amax_row, relu_mask);
  MVAL_CHECK_LAUNCH("mval_bn_bwd_fused/apply");
  return 0;
}

// ---- measurement of one P2 tensor (the training plan's bound-slack probe, engine_train.TrainPlan.p2_slack; not on the timed path) ----
// out4 (zeroed by the caller): [0] float bits of 2^-s of image 0, [1] float bits of max |h + l| in SCALED units (the a-priori bound sits in
// [2^13, 2^14) there), [2] count of non-zero values whose scaled magnitude is below 2^-3 (their l runs into fp16's subnormal spacing: fewer
// than 22 significand bits survive, conv_p2.h), [3] count of non-zero values.
__global__ __launch_bounds__(256) void p2_plane_stats_kernel(const _Float16* __restrict__ planes, const unsigned* __restrict__ rows, int N, int C8,
                                                             int HW, unsigned* __restrict__ out4) {
  const int64_t per_img = (int64_t)C8 * HW, total = (int64_t)N * per_img;
  float mx = 0.f;
  unsigned small = 0, nz = 0;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int64_t n = g / per_img, r = g - n * per_img;
    const _Float16* hp = planes + (n * 2 * per_img + r) * 8;
    const p2_f16x8 h = *reinterpret_cast<const p2_f16x8*>(hp), l = *reinterpret_cast<const p2_f16x8*>(hp + per_img * 8);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float v = fabsf((float)h[k] + (float)l[k]);
      mx = fmaxf(mx, v);
      nz += v > 0.f;
      small += v > 0.f && v < 0.125f;
    }
  }
  mx = wave_max(mx);
  for (int o = 32; o > 0; o >>= 1) {
    small += __shfl_xor(small, o);
    nz += __shfl_xor(nz, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out4 + 1, __float_as_uint(mx));
    atomicAdd(out4 + 2, small);
    atomicAdd(out4 + 3, nz);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out4[0] = rows[P2_INV_SLOT];
}
extern "C" int mval_p2_plane_stats(const void* planes, const uint32_t* rows, int N, int C, int HW, uint32_t* out4, void* stream) {
  MVAL_REQUIRE(planes && rows && out4 && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "mval_p2_plane_stats: bad arguments (C % 8)");
  MVAL_REQUIRE((int64_t)N * C * HW < ((int64_t)1 << 32), "mval_p2_plane_stats: more than 2^32 elements (the counts are 32-bit)");
  const int64_t total = (int64_t)N * (C >> 3) * HW;
  const int nb = (int)std::min<int64_t>(1024, (total + 255) / 256);
  hipLaunchKernelGGL(p2_plane_stats_kernel, dim3(nb), dim3(256), 0, mval_stream(stream), reinterpret_cast<const _Float16*>(planes), rows, N,
                     C >> 3, HW, out4);
  MVAL_CHECK_LAUNCH("mval_p2_plane_stats");
  return 0;
}

// ---- max-pool backward (PoseResNet stem, pose_resnet.py:35: MaxPool2d(3, 2, 1)) ----------------
// gin[n, iy, ix, c] (+)= sum of gout over the output windows whose arg-max is (iy, ix).  Gather form
// (one thread per input float4, no atomics): each of the <= 4 windows covering the element is
// re-scanned in ATen's order (rows, then columns, strictly-greater update, so the FIRST maximum wins
// -- after a ReLU whole windows tie at 0) and contributes when its arg-max is this element.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                          float* __restrict__ gin, int N, int Hin, int Win, int C,
                                                          int Hout, int Wout, int k, int stride, int pad,
                                                          int accumulate) {
  const int c4n = C >> 2;
  const int64_t total = (int64_t)N * Hin * Win * c4n;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int q = (int)(t % c4n);
    int64_t p = t / c4n;
    const int ix = (int)(p % Win);
    const int iy = (int)((p / Win) % Hin);
    const int n = (int)(p / ((int64_t)Win * Hin));
    f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
    // windows (oy, ox) with oy * stride - pad <= iy < oy * stride - pad + k
    const int oy_lo = max(0, (iy + pad - k + stride) / stride), oy_hi = min(Hout - 1, (iy + pad) / stride);
    const int ox_lo = max(0, (ix + pad - k + stride) / stride), ox_hi = min(Wout - 1, (ix + pad) / stride);
    for (int oy = oy_lo; oy <= oy_hi; oy++)
      for (int ox = ox_lo; ox <= ox_hi; ox++) {
        const int y0 = max(0, oy * stride - pad), y1 = min(Hin, oy * stride - pad + k);
        const int x0 = max(0, ox * stride - pad), x1 = min(Win, ox * stride - pad + k);
        f32x4 best = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int by[4] = {y0, y0, y0, y0}, bx[4] = {x0, x0, x0, x0};
        for (int yy = y0; yy < y1; yy++)
          for (int xx = x0; xx < x1; xx++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((int64_t)n * Hin + yy) * Win + xx) * C + q * 4);
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (v[j] > best[j] || v[j] != v[j]) {
                best[j] = v[j];
                by[j] = yy;
                bx[j] = xx;
              }
          }
        const f32x4 go = *reinterpret_cast<const f32x4*>(gout + (((int64_t)n * Hout + oy) * Wout + ox) * C + q * 4);
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (by[j] == iy && bx[j] == ix) g[j] += go[j];
      }
    f32x4* dst = reinterpret_cast<f32x4*>(gin + p * C + q * 4);
    *dst = accumulate ? *dst + g : g;
  }
}

extern "C" int mval_maxpool_bwd(const float* gout, const float* x, float* gin, int N, int Hin, int Win, int C, int Hout,
                                int Wout, int k, int stride, int pad, int accumulate, void* stream) {
  MVAL_REQUIRE(gout && x && gin && N > 0 && C > 0 && (C & 3) == 0 && k > 0 && stride > 0 && k >= stride,
               "mval_maxpool_bwd: bad arguments (C must be a multiple of 4, k >= stride)");
  const int64_t total = (int64_t)N * Hin * Win * (C >> 2);
  int nb = (int)((total + 255) / 256);
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nb), dim3(256), 0, mval_stream(stream), gout, x, gin, N, Hin, Win, C, Hout,
                     Wout, k, stride, pad, accumulate);
  MVAL_CHECK_LAUNCH("mval_maxpool_bwd");
  return 0;
}

// sum of S slabs of n floats (float64 accumulation), deterministic split-K reduction
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int S, int64_t n, float* __restrict__ out,
                                   int accumulate) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = 0;
  for (int s = 0; s < S; s++) a += (double)slabs[(int64_t)s * n + i];
  out[i] = accumulate ? out[i] + (float)a : (float)a;
}

extern "C" int mval_slab_reduce(const float* slabs, int S, int64_t n, float* out, int accumulate, void* stream) {
  MVAL_REQUIRE(S > 0 && n > 0, "mval_slab_reduce: bad dims");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mval_stream(stream), slabs, S,
                     n, out, accumulate);
  MVAL_CHECK_LAUNCH("mval_slab_reduce");
  return 0;
}
