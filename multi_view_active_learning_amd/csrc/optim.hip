// Adam over every parameter of the model in ONE launch (reference: strategy.py:405-407 builds torch.optim.Adam, :479 steps it).
//
// HBM-bound streaming: 16 bytes read (p, g, m, v) and 12 written (p, m, v) per parameter -- 800 MB for HRNet-W32's 28.5 M
// parameters, 0.16 ms at the HBM rate.  torch's default (foreach) implementation walks the 300 tensors in five passes of ~19
// multi-tensor launches each: 1.8 ms of a 70 ms training step (profiles/r04/bench_c3_kernel_stats_r4.csv).
//
// The arithmetic is torch/optim/adam.py's _single_tensor_adam, op for op in float32 (python scalars become float32 there too):
//     g  = grad (+ weight_decay * p)
//     m  = m + (1 - beta1) * (g - m)                      exp_avg.lerp_(grad, 1 - beta1)
//     v  = v * beta2 + (1 - beta2) * g * g                exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//     p  = p - step_size * (m / (sqrt(v) / bc2_sqrt + eps))   param.addcdiv_(exp_avg, denom, value = -step_size)
// with step_size = lr / (1 - beta1^t) and bc2_sqrt = sqrt(1 - beta2^t) evaluated by the host in double, as python does.
#include "mval_common.h"

struct AdamJob {  // = mval_adam_job (include/mval_hip.h)
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t count;
};

#define ADAM_BLOCK_ELEMS 4096  // per workgroup: 256 threads x 4 rounds x float4

struct AdamHyper {
  float w1, beta2, w2, eps, weight_decay, neg_step_size, inv_dummy, bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, const float g0, float& m, float& v, const AdamHyper& h) {
  const float g = h.weight_decay != 0.f ? g0 + h.weight_decay * p : g0;
  m = m + h.w1 * (g - m);
  v = v * h.beta2;
  v = v + h.w2 * g * g;
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p = p + h.neg_step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamJob* __restrict__ jobs, const int* __restrict__ first_block, int n_jobs,
                                                        AdamHyper h) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {  // the job this workgroup belongs to (first_block is ascending)
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const AdamJob jb = jobs[lo];
  const int64_t base = (int64_t)((int)blockIdx.x - first_block[lo]) * ADAM_BLOCK_ELEMS;
  const int64_t end = min(jb.count, base + ADAM_BLOCK_ELEMS);
  const bool vec = ((reinterpret_cast<uintptr_t>(jb.p) | reinterpret_cast<uintptr_t>(jb.g) | reinterpret_cast<uintptr_t>(jb.m) |
                     reinterpret_cast<uintptr_t>(jb.v)) & 15) == 0;
  if (vec) {
    float4 P[4], G[4], M[4], V[4];
    int64_t idx[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {  // every load of the workgroup's 64 KB is requested before the first use
      idx[r] = base + (int64_t)(r * 256 + (int)threadIdx.x) * 4;
      if (idx[r] + 4 <= end) {
        P[r] = *reinterpret_cast<const float4*>(jb.p + idx[r]);
        G[r] = *reinterpret_cast<const float4*>(jb.g + idx[r]);
        M[r] = *reinterpret_cast<const float4*>(jb.m + idx[r]);
        V[r] = *reinterpret_cast<const float4*>(jb.v + idx[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (idx[r] + 4 <= end) {
        adam_one(P[r].x, G[r].x, M[r].x, V[r].x, h);
        adam_one(P[r].y, G[r].y, M[r].y, V[r].y, h);
        adam_one(P[r].z, G[r].z, M[r].z, V[r].z, h);
        adam_one(P[r].w, G[r].w, M[r].w, V[r].w, h);
        *reinterpret_cast<float4*>(jb.p + idx[r]) = P[r];
        *reinterpret_cast<float4*>(jb.m + idx[r]) = M[r];
        *reinterpret_cast<float4*>(jb.v + idx[r]) = V[r];
      } else {
        for (int64_t i = idx[r]; i < end; i++) {  // the tensor's last, partial quad
          float p = jb.p[i], m = jb.m[i], v = jb.v[i];
          adam_one(p, jb.g[i], m, v, h);
          jb.p[i] = p; jb.m[i] = m; jb.v[i] = v;
        }
      }
    }
  } else {
    for (int64_t i = base + threadIdx.x; i < end; i += 256) {
      float p = jb.p[i], m = jb.m[i], v = jb.v[i];
      adam_one(p, jb.g[i], m, v, h);
      jb.p[i] = p; jb.m[i] = m; jb.v[i] = v;
    }
  }
}

extern "C" int mval_adam_block_elems(void) { return ADAM_BLOCK_ELEMS; }

extern "C" int mval_adam_step(const mval_adam_job* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, float one_minus_beta1,
                              float beta2, float one_minus_beta2, float eps, float weight_decay, float step_size, float bias_correction2_sqrt,
                              void* stream) {
  MVAL_REQUIRE(n_jobs >= 0 && total_blocks >= 0, "mval_adam_step: negative counts");
  if (n_jobs == 0 || total_blocks == 0) return 0;
  MVAL_REQUIRE(jobs_dev && first_block_dev, "mval_adam_step: null job table");
  MVAL_REQUIRE(bias_correction2_sqrt > 0.f, "mval_adam_step: bias_correction2_sqrt must be positive");
  static_assert(sizeof(AdamJob) == sizeof(mval_adam_job), "job layout");
  AdamHyper h = {one_minus_beta1, beta2, one_minus_beta2, eps, weight_decay, -step_size, 0.f, bias_correction2_sqrt};
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)total_blocks), dim3(256), 0, mval_stream(stream),
                     reinterpret_cast<const AdamJob*>(jobs_dev), first_block_dev, n_jobs, h);
  MVAL_CHECK_LAUNCH("mval_adam_step");
  return 0;
}
