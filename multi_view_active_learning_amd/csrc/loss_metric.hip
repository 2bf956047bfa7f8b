// Masked heat-map MSE (K8, reference pose_estimators/loss.py:14-20) forward/backward and
// MKPE / MPJPE (K15, reference utils/evaluation.py:198-208).  HBM-bound streaming kernels.
#include "mval_common.h"

#define MSE_BLOCKS 1024

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ h, const float* __restrict__ g,
                                                          const uint8_t* __restrict__ valid, double* __restrict__ ws,
                                                          int64_t lead, int64_t hw) {
  __shared__ double red[4];
  double acc = 0.0;
  const int64_t total = lead * hw;
  const bool vec = (hw & 3) == 0 && (((uintptr_t)h | (uintptr_t)g) & 15) == 0;
  if (vec) {
    const int64_t n4 = total >> 2;
    const float4* h4 = reinterpret_cast<const float4*>(h);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      if (valid && !valid[(i << 2) / hw]) continue;
      float4 a = h4[i], b = g4[i];
      float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
      acc += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      if (valid && !valid[i / hw]) continue;
      float d = h[i] - g[i];
      acc += (double)(d * d);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ ws, float* __restrict__ out, int nb,
                                                        double denom) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += ws[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / denom);
}

extern "C" int mval_masked_mse_fwd(const float* h, const float* g, const uint8_t* valid, float* out, double* ws,
                                   int64_t lead, int64_t hw, double denom, void* stream) {
  MVAL_REQUIRE(lead >= 0 && hw > 0 && denom != 0.0, "mval_masked_mse_fwd: bad dims");
  int64_t total = lead * hw;
  int nb = (int)((total / 4 + 255) / 256);
  if (nb < 1) nb = 1;
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, mval_stream(stream), h, g, valid, ws, lead, hw);
  MVAL_CHECK_LAUNCH("mval_masked_mse_fwd");
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, mval_stream(stream), ws, out, nb, denom);
  MVAL_CHECK_LAUNCH("mval_masked_mse_fwd/final");
  return 0;
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ h, const float* __restrict__ g,
                                                      const uint8_t* __restrict__ valid,
                                                      const float* __restrict__ grad_out, float* __restrict__ gh,
                                                      int64_t lead, int64_t hw, float denom) {
  const float gs = grad_out[0] / denom;  // d(sum/denom)
  const int64_t total = lead * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    bool ok = !valid || valid[i / hw];
    float d = h[i] - g[i];
    gh[i] = ok ? (2.0f * d) * gs : 0.0f;
  }
}

extern "C" int mval_masked_mse_bwd(const float* h, const float* g, const uint8_t* valid, const float* grad_out,
                                   float* grad_h, int64_t lead, int64_t hw, double denom, void* stream) {
  MVAL_REQUIRE(lead >= 0 && hw > 0 && denom != 0.0, "mval_masked_mse_bwd: bad dims");
  int64_t total = lead * hw;
  if (total == 0) return 0;
  int nb = (int)((total + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(nb), dim3(256), 0, mval_stream(stream), h, g, valid, grad_out, grad_h, lead,
                     hw, (float)denom);
  MVAL_CHECK_LAUNCH("mval_masked_mse_bwd");
  return 0;
}

// ---- MKPE ------------------------------------------------------------------------------
// d_sj = sqrt(sum_c valid ? (pred[s,j,c] - gt[s,c,j])^2 : 0)   (float32, c = 0,1,2 in order)
__device__ __forceinline__ float mkpe_d(const float* pred, const float* gt, const float* valid, int64_t s, int j,
                                        int J, int gt_rows) {
  bool ok = valid[s * J + j] != 0.0f;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float d = pred[(s * J + j) * 3 + c] - gt[(s * gt_rows + c) * J + j];
    acc += ok ? d * d : 0.f;
  }
  return sqrtf(acc);
}

__global__ void mkpe_per_sample_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                       const float* __restrict__ valid, float* __restrict__ per_sample, int64_t S,
                                       int J, int gt_rows) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float acc = 0.f;
  for (int j = 0; j < J; j++) acc += mkpe_d(pred, gt, valid, s, j, J, gt_rows) / valid[s * J + j];  // 0/0 -> NaN
  per_sample[s] = acc / (float)J;
}

__global__ void mkpe_total_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                  const float* __restrict__ valid, float* __restrict__ out, int64_t S, int J,
                                  int gt_rows) {
  __shared__ float mk[1024];
  int j = threadIdx.x;
  if (j < J) {
    float kpe = 0.f, cnt = 0.f;  // kpe = kpe + d ; count = count + valid  (sample order)
    for (int64_t s = 0; s < S; s++) {
      kpe += mkpe_d(pred, gt, valid, s, j, J, gt_rows);
      cnt += valid[s * J + j];
    }
    mk[j] = kpe / cnt;
  }
  __syncthreads();
  if (j == 0) {
    float acc = 0.f;
    for (int k = 0; k < J; k++) acc += mk[k];
    out[0] = acc / (float)J;
  }
}

extern "C" int mval_mkpe(const float* pred, const float* gt, const float* valid, float* out, float* per_sample,
                         int64_t S, int J, int gt_rows, void* stream) {
  MVAL_REQUIRE(S >= 0 && J > 0 && J <= 1024 && gt_rows >= 3, "mval_mkpe: bad dims");
  if (S > 0 && per_sample) {
    hipLaunchKernelGGL(mkpe_per_sample_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, mval_stream(stream), pred,
                       gt, valid, per_sample, S, J, gt_rows);
    MVAL_CHECK_LAUNCH("mval_mkpe/per_sample");
  }
  if (out) {
    hipLaunchKernelGGL(mkpe_total_kernel, dim3(1), dim3(((J + 63) / 64) * 64), 0, mval_stream(stream), pred, gt, valid,
                       out, S, J, gt_rows);
    MVAL_CHECK_LAUNCH("mval_mkpe/total");
  }
  return 0;
}

// ---- 3-D PCK / PCKh --------------------------------------------------------------------------
// utils/evaluation.py:150-195 (compute_3d_pckh / compute_3d_pck), the counters of _evaluate_all
// (strategy.py:638-649).  Distances are float32 with torch's operation order -- ((dx^2 + dy^2) + dz^2),
// every operation rounded separately (no FMA contraction), then sqrt -- because the result is an
// integer count of strict `<` comparisons:
//   mode 0 (PCK):  valid joints only; hit when (double)dis < threshold[t]; counts[j] = #valid samples
//   mode 1 (PCKh): every joint; hit when dis < (float32)(head * (float)threshold[t]) with head = the
//                  distance between gt joints 0 and 1; counts[j] = S
// One workgroup per (threshold, joint); integer sums, so the result is order-independent.
__device__ __forceinline__ float pck_dist(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
}

__global__ __launch_bounds__(256) void pck3d_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                    const float* __restrict__ valid,
                                                    const double* __restrict__ thresholds, int mode,
                                                    long long* __restrict__ hits, long long* __restrict__ counts,
                                                    int64_t S, int J, int gt_rows) {
  __shared__ long long sh[2][256];
  const int j = blockIdx.x, t = blockIdx.y;
  const double thr = thresholds[t];
  long long h = 0, c = 0;
  for (int64_t s = threadIdx.x; s < S; s += 256) {
    const float* g = gt + s * gt_rows * J;
    const float* pr = pred + (s * J + j) * 3;
    if (mode == 0 && valid[s * J + j] == 0.0f) continue;
    const float dis = pck_dist(pr[0], pr[1], pr[2], g[j], g[J + j], g[2 * J + j]);
    c++;
    if (mode == 0) {
      h += (double)dis < thr;
    } else {
      const float head = pck_dist(g[0], g[J], g[2 * J], g[1], g[J + 1], g[2 * J + 1]);
      h += dis < __fmul_rn(head, (float)thr);
    }
  }
  sh[0][threadIdx.x] = h;
  sh[1][threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hits[(int64_t)t * J + j] = sh[0][0];
    if (t == 0) counts[j] = sh[1][0];
  }
}

extern "C" int mval_pck3d(const float* pred, const float* gt, const float* valid, const double* thresholds, int T,
                          int mode, long long* hits, long long* counts, int64_t S, int J, int gt_rows, void* stream) {
  MVAL_REQUIRE(pred && gt && thresholds && hits && counts && S >= 0 && J >= 2 && T > 0 && gt_rows >= 3 &&
                   (mode == 1 || (mode == 0 && valid)),
               "mval_pck3d: bad arguments");
  hipLaunchKernelGGL(pck3d_kernel, dim3(J, T), dim3(256), 0, mval_stream(stream), pred, gt, valid, thresholds, mode, hits,
                     counts, S, J, gt_rows);
  MVAL_CHECK_LAUNCH("mval_pck3d");
  return 0;
}
