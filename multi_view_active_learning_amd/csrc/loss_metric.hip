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
