// Core-set selection: greedy k-center (K14, reference utils/coreset.py:35-95).
//
// HBM/MALL-bound: every greedy step streams the whole feature table once
// (n_obs * D * 8 B; 22.9 MB at the BASELINE pool) plus the running-minimum vector.
// The reference does this with sklearn on the CPU, redundantly on every rank.
//
// Device design
//   * features are transposed once to [D][n_obs] so that the per-step distance pass is a
//     perfectly coalesced stream (thread = row, 8-byte lanes contiguous), the centre vector
//     sits in LDS and is broadcast;
//   * ONE launch per greedy step: each workgroup first reduces the previous step's
//     per-workgroup (max, index) partials to the global arg-max (redundantly, <= 1024
//     pairs), then does its slice of  min_d = minimum(min_d, dist(., centre))  and emits its
//     own partial for the next step -- no grid barrier, no host round trip, graph-capturable;
//   * arithmetic is sklearn's expanded form in float64, in sklearn's operation order:
//     d = sqrt(max(0, (-2 x.c + |x|^2) + |c|^2)); first maximum wins ties; NaN is a maximum
//     (np.argmax) and propagates through minimum (np.minimum); nothing is masked.
#include "mval_common.h"

#define KC_THREADS 256
#define KC_MAX_BLOCKS 1024
#define KC_MAX_D 512

struct KcPartial {
  double val;
  int64_t idx;
};

__device__ __forceinline__ bool kc_better(double v, int64_t i, double bv, int64_t bi) {
  bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (vn) return i < bi;
  return (v > bv) || (v == bv && i < bi);
}
__device__ __forceinline__ double np_minimum(double a, double b) {
  if (a != a) return a;
  if (b != b) return b;
  return a < b ? a : b;
}

__device__ __forceinline__ KcPartial kc_block_reduce(KcPartial p, KcPartial* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(p.val, o, 64);
    long long oi = __shfl_xor((long long)p.idx, o, 64);
    if (kc_better(ov, oi, p.val, p.idx)) { p.val = ov; p.idx = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = p;
  __syncthreads();
  KcPartial r = sh[0];
  for (int w = 1; w < KC_THREADS / 64; w++)
    if (kc_better(sh[w].val, sh[w].idx, r.val, r.idx)) r = sh[w];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(KC_THREADS) void kc_transpose_norm_kernel(const double* __restrict__ feat,
                                                                       double* __restrict__ featT,
                                                                       double* __restrict__ norms, int64_t n, int D) {
  int64_t i = (int64_t)blockIdx.x * KC_THREADS + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int d = 0; d < D; d++) {
    double x = feat[i * D + d];
    featT[(int64_t)d * n + i] = x;
    s = fma(x, x, s);
  }
  norms[i] = s;
}

// min over the labeled centres (coreset.py:64-67), 4 centres per pass over a row
__global__ __launch_bounds__(KC_THREADS) void kc_init_kernel(const double* __restrict__ feat,
                                                             const double* __restrict__ featT,
                                                             const double* __restrict__ norms,
                                                             const int64_t* __restrict__ labeled, int64_t n_labeled,
                                                             double* __restrict__ min_d, int have_min,
                                                             KcPartial* __restrict__ part, int64_t n, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* cen = reinterpret_cast<double*>(smem_raw);  // [4][D]
  __shared__ KcPartial sh[KC_THREADS / 64];
  __shared__ double cn[4];
  KcPartial best;
  best.val = -INFINITY;
  best.idx = INT64_MAX;
  for (int64_t i0 = (int64_t)blockIdx.x * KC_THREADS; i0 < n; i0 += (int64_t)gridDim.x * KC_THREADS) {
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < n;
    double md = (have_min && live) ? min_d[i] : INFINITY;
    bool first = !have_min;
    const double xx = live ? norms[i] : 0.0;
    for (int64_t c0 = 0; c0 < n_labeled; c0 += 4) {
      int nc = (int)min((int64_t)4, n_labeled - c0);
      __syncthreads();
      for (int t = threadIdx.x; t < nc * D; t += KC_THREADS) cen[t] = feat[labeled[c0 + t / D] * D + t % D];
      if (threadIdx.x < nc) cn[threadIdx.x] = norms[labeled[c0 + threadIdx.x]];
      __syncthreads();
      if (live) {
        double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        for (int d = 0; d < D; d++) {
          double x = featT[(int64_t)d * n + i];
          d0 = fma(x, cen[d], d0);
          if (nc > 1) d1 = fma(x, cen[D + d], d1);
          if (nc > 2) d2 = fma(x, cen[2 * D + d], d2);
          if (nc > 3) d3 = fma(x, cen[3 * D + d], d3);
        }
        double dd[4] = {d0, d1, d2, d3};
        for (int k = 0; k < nc; k++) {
          double t = -2.0 * dd[k];
          t += xx;
          t += cn[k];
          t = sqrt(t > 0.0 ? t : (t != t ? t : 0.0));
          md = first ? t : np_minimum(md, t);
          first = false;
        }
      }
    }
    if (live) {
      min_d[i] = md;
      if (kc_better(md, i, best.val, best.idx)) { best.val = md; best.idx = i; }
    }
  }
  KcPartial r = kc_block_reduce(best, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(KC_THREADS) void kc_step_kernel(const double* __restrict__ featT,
                                                             const double* __restrict__ norms,
                                                             double* __restrict__ min_d,
                                                             const KcPartial* __restrict__ part_in, int n_part_in,
                                                             KcPartial* __restrict__ part_out,
                                                             int64_t* __restrict__ picks, int step, int64_t n, int D) {
  __shared__ KcPartial sh[KC_THREADS / 64];
  __shared__ double cen[KC_MAX_D];
  // (1) global arg-max of the previous pass, redundantly per workgroup
  KcPartial b;
  b.val = -INFINITY;
  b.idx = INT64_MAX;
  for (int t = threadIdx.x; t < n_part_in; t += KC_THREADS) {
    KcPartial q = part_in[t];
    if (kc_better(q.val, q.idx, b.val, b.idx)) b = q;
  }
  b = kc_block_reduce(b, sh);
  const int64_t ind = b.idx;
  if (blockIdx.x == 0 && threadIdx.x == 0) picks[step] = ind;
  for (int d = threadIdx.x; d < D; d += KC_THREADS) cen[d] = featT[(int64_t)d * n + ind];
  __syncthreads();
  const double yy = norms[ind];
  // (2) min_d = minimum(min_d, dist(., centre)) on this workgroup's rows + next partial
  KcPartial best;
  best.val = -INFINITY;
  best.idx = INT64_MAX;
  for (int64_t i = (int64_t)blockIdx.x * KC_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * KC_THREADS) {
    double dot = 0.0;
    for (int d = 0; d < D; d++) dot = fma(featT[(int64_t)d * n + i], cen[d], dot);
    double t = -2.0 * dot;
    t += norms[i];
    t += yy;
    t = sqrt(t > 0.0 ? t : (t != t ? t : 0.0));
    double md = np_minimum(min_d[i], t);
    min_d[i] = md;
    if (kc_better(md, i, best.val, best.idx)) { best.val = md; best.idx = i; }
  }
  KcPartial r = kc_block_reduce(best, sh);
  if (threadIdx.x == 0) part_out[blockIdx.x] = r;
}

static int kc_blocks(int64_t n) {
  int64_t nb = (n + KC_THREADS - 1) / KC_THREADS;
  if (nb > KC_MAX_BLOCKS) nb = KC_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" size_t mval_kcenter_workspace_bytes(int64_t n_obs, int D) {
  return (size_t)n_obs * D * sizeof(double) + 2 * KC_MAX_BLOCKS * sizeof(KcPartial) + 256;
}

extern "C" int mval_kcenter_select(const double* feat, int64_t n_obs, int D, const int64_t* labeled, int64_t n_labeled,
                                   int n_select, int have_min_dist, double* row_norms, double* min_dist, int64_t* picks,
                                   void* ws, void* stream) {
  MVAL_REQUIRE(n_obs > 0 && D > 0 && D <= KC_MAX_D && n_select >= 0 && n_labeled >= 0, "mval_kcenter_select: bad dims");
  hipStream_t s = mval_stream(stream);
  double* featT = reinterpret_cast<double*>(ws);
  KcPartial* part = reinterpret_cast<KcPartial*>(featT + (size_t)n_obs * D);
  part = reinterpret_cast<KcPartial*>(((uintptr_t)part + 15) & ~(uintptr_t)15);
  const int nb = kc_blocks(n_obs);
  hipLaunchKernelGGL(kc_transpose_norm_kernel, dim3((unsigned)((n_obs + KC_THREADS - 1) / KC_THREADS)),
                     dim3(KC_THREADS), 0, s, feat, featT, row_norms, n_obs, D);
  MVAL_CHECK_LAUNCH("mval_kcenter_select/transpose");
  hipLaunchKernelGGL(kc_init_kernel, dim3(nb), dim3(KC_THREADS), (size_t)4 * D * sizeof(double), s, feat, featT,
                     row_norms, labeled, n_labeled, min_dist, have_min_dist, part, n_obs, D);
  MVAL_CHECK_LAUNCH("mval_kcenter_select/init");
  for (int t = 0; t < n_select; t++) {
    KcPartial* pin = part + (t & 1) * KC_MAX_BLOCKS;
    KcPartial* pout = part + ((t + 1) & 1) * KC_MAX_BLOCKS;
    hipLaunchKernelGGL(kc_step_kernel, dim3(nb), dim3(KC_THREADS), 0, s, featT, row_norms, min_dist, pin, nb, pout,
                       picks, t, n_obs, D);
  }
  MVAL_CHECK_LAUNCH("mval_kcenter_select/step");
  return 0;
}

// utils/coreset.py:40-46: (pose^T)[0:3,:] - (pose^T)[0:3, root]  flattened coord-major
__global__ void coreset_features_kernel(const double* __restrict__ pose, double* __restrict__ feat, int64_t n, int J,
                                        int rows, int root) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = n * 3 * J;
  if (t >= total) return;
  int64_t i = t / (3 * J);
  int r = (int)(t % (3 * J));
  int c = r / J, j = r % J;
  const double* p = pose + i * (int64_t)J * rows;
  feat[t] = p[j * rows + c] - p[root * rows + c];
}

extern "C" int mval_coreset_features(const double* pose, double* feat, int64_t n, int J, int rows, int root,
                                     void* stream) {
  MVAL_REQUIRE(n >= 0 && J > 0 && rows >= 3 && root >= 0 && root < J, "mval_coreset_features: bad dims (root=%d J=%d)",
               root, J);
  if (n == 0) return 0;
  int64_t total = n * 3 * J;
  hipLaunchKernelGGL(coreset_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mval_stream(stream),
                     pose, feat, n, J, rows, root);
  MVAL_CHECK_LAUNCH("mval_coreset_features");
  return 0;
}

// ---- nearest cluster centre (strategy.py:981-989: ``self.kmeans.predict([kp])`` per pseudo-label candidate)
// label[i] = argmin_k ||feat_i - centre_k||^2 (float64, first minimum), evaluated like sklearn's KMeans.predict
// as ||c_k||^2 - 2 feat_i . c_k (the ||feat_i||^2 term does not change the argmin).  One thread per sample.
__global__ __launch_bounds__(256) void nearest_center_kernel(const double* __restrict__ feat,
                                                             const double* __restrict__ centers, int64_t n, int D, int K,
                                                             int* __restrict__ label) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* x = feat + i * D;
  double best = 0.0;
  int bk = 0;
  for (int k = 0; k < K; k++) {
    const double* c = centers + (int64_t)k * D;
    double cc = 0.0, xc = 0.0;
    for (int d = 0; d < D; d++) {
      cc += c[d] * c[d];
      xc += x[d] * c[d];
    }
    const double v = cc - 2.0 * xc;
    if (k == 0 || v < best) {
      best = v;
      bk = k;
    }
  }
  label[i] = bk;
}

extern "C" int mval_nearest_center(const double* feat, const double* centers, int64_t n, int D, int K, int* label,
                                   void* stream) {
  MVAL_REQUIRE(feat && centers && label && n >= 0 && D > 0 && K > 0, "mval_nearest_center: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(nearest_center_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mval_stream(stream), feat,
                     centers, n, D, K, label);
  MVAL_CHECK_LAUNCH("mval_nearest_center");
  return 0;
}
