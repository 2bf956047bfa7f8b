// Batched pairwise-RANSAC DLT triangulation + reprojection error (K11), XE metric (K12).
//
// Replaces the per-sample / per-joint / per-pair NumPy loop of the reference
// (utils/triangulation.py:209-233, 260-338, 341-384): C(V,2)+1 LAPACK SVD calls per joint.
//
// Mapping: one (frame, joint) problem per group of PG lanes (PG = pow2 >= C(V,2), <= 64);
// lane p of a group triangulates view pair p and votes inliers over all V views; a
// butterfly picks the first pair with the largest inlier set; the final DLT over the
// sorted inlier views is solved redundantly by the group and written by its lane 0.
//
// The DLT null vector is computed WITHOUT an SVD library and without forming A^T A (which
// squares the condition number: entries of A reach 1e5): rows of the 2n x 4 system are
// folded into a 4x4 upper-triangular R with Givens rotations (backward stable, streaming,
// no 2n-sized array), then a one-sided Jacobi (Hestenes) iteration on R yields the right
// singular vector of the smallest singular value to high relative accuracy.  All float64.
// Latency/ALU-bound; bytes are negligible (V*J*2 ints + V*12 doubles per frame).
#include "mval_common.h"

#define MAXV 11  // C(11,2) = 55 <= 64 = n_iters: beyond that the reference samples pairs randomly

struct R4 {
  double r[4][4];  // upper triangular (lower part kept zero)
};

__device__ __forceinline__ void r4_zero(R4& m) {
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) m.r[i][j] = 0.0;
}

// fold one row a[0..3] into R with 4 Givens rotations
__device__ __forceinline__ void r4_add_row(R4& m, double a0, double a1, double a2, double a3) {
  double a[4] = {a0, a1, a2, a3};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double x = m.r[i][i], y = a[i];
    if (y != 0.0) {
      double h = hypot(x, y);
      double c = x / h, s = y / h;
      m.r[i][i] = h;
      a[i] = 0.0;
#pragma unroll
      for (int j = i + 1; j < 4; j++) {
        double rj = m.r[i][j], aj = a[j];
        m.r[i][j] = c * rj + s * aj;
        a[j] = c * aj - s * rj;
      }
    }
  }
}

// right singular vector of the smallest singular value of the 4x4 matrix R
__device__ __forceinline__ void r4_null_vector(const R4& m, double x[4]) {
  double g[4][4], v[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      g[i][j] = m.r[i][j];
      v[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          al = fma(g[k][p], g[k][p], al);
          be = fma(g[k][q], g[k][q], be);
          ga = fma(g[k][p], g[k][q], ga);
        }
        if (fabs(ga) > 1e-17 * sqrt(al * be) && ga != 0.0) {
          rotated = true;
          double zeta = (be - al) / (2.0 * ga);
          double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            double gp = g[k][p], gq = g[k][q];
            g[k][p] = c * gp - s * gq;
            g[k][q] = s * gp + c * gq;
            double vp = v[k][p], vq = v[k][q];
            v[k][p] = c * vp - s * vq;
            v[k][q] = s * vp + c * vq;
          }
        }
      }
    }
    if (!rotated) break;
  }
  double best = INFINITY;
  int bi = 3;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double n = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) n = fma(g[k][j], g[k][j], n);
    if (n < best) { best = n; bi = j; }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = (bi == 0) ? v[k][0] : (bi == 1) ? v[k][1] : (bi == 2) ? v[k][2] : v[k][3];
}

__device__ __forceinline__ void add_view_rows(R4& m, const double* __restrict__ P, double px, double py) {
  // utils/triangulation.py:357-361: x*P[2,:] - P[0,:] ; y*P[2,:] - P[1,:]
  r4_add_row(m, px * P[8] - P[0], px * P[9] - P[1], px * P[10] - P[2], px * P[11] - P[3]);
  r4_add_row(m, py * P[8] - P[4], py * P[9] - P[5], py * P[10] - P[6], py * P[11] - P[7]);
}

__device__ __forceinline__ void dehomogenise(const double h[4], double X[3]) {
  double w = (h[3] == 0.0) ? 1.0 : h[3];  // utils/triangulation.py:397-399
  X[0] = h[0] / w;
  X[1] = h[1] / w;
  X[2] = h[2] / w;
}

// utils/triangulation.py:371-384,459-477: 1/2 * || pt - pi(P [X;1]) ||
__device__ __forceinline__ double reproj_err(const double* __restrict__ P, const double X[3], double px, double py) {
  double u = fma(X[2], P[2], fma(X[1], P[1], X[0] * P[0])) + P[3];
  double v = fma(X[2], P[6], fma(X[1], P[5], X[0] * P[4])) + P[7];
  double w = fma(X[2], P[10], fma(X[1], P[9], X[0] * P[8])) + P[11];
  if (w == 0.0) w = 1.0;
  double dx = px - u / w, dy = py - v / w;
  return 0.5 * sqrt(dx * dx + dy * dy);
}

// numpy's pairwise summation (np.add.reduce on a contiguous double vector), so that
// np.mean(...) in the reference is reproduced bit for bit given equal addends.
__device__ double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; i++) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

template <typename KP>
__global__ __launch_bounds__(64) void ransac_dlt_kernel(const KP* __restrict__ kp2d, const double* __restrict__ proj,
                                                         const uint8_t* __restrict__ valid, double* __restrict__ kp3d,
                                                         double* __restrict__ joint_err,
                                                         int32_t* __restrict__ joint_inliers, int64_t n_prob, int V,
                                                         int J, int n_pairs, int PG, double eps) {
  const int lane = threadIdx.x;
  const int grp = lane / PG, p = lane % PG;
  const int64_t prob = (int64_t)blockIdx.x * (64 / PG) + grp;
  const bool live = prob < n_prob;
  const int64_t pr = live ? prob : 0;
  const int64_t b = pr / J;
  const int j = (int)(pr % J);
  const bool is_valid = live && (!valid || valid[b * J + j]);
  const double* Pb = proj + b * V * 12;
  const KP* kb = kp2d + (b * V * (int64_t)J + j) * 2;  // + v * J * 2

  // ---- stage 1: lane p triangulates pair p and votes ----------------------------------
  unsigned mask = 0;
  int count = 0;
  if (p < n_pairs) {
    int a = 0, rem = p;
    while (rem >= V - 1 - a) { rem -= V - 1 - a; a++; }
    int c = a + 1 + rem;
    R4 m;
    r4_zero(m);
    add_view_rows(m, Pb + a * 12, (double)kb[(int64_t)a * J * 2], (double)kb[(int64_t)a * J * 2 + 1]);
    add_view_rows(m, Pb + c * 12, (double)kb[(int64_t)c * J * 2], (double)kb[(int64_t)c * J * 2 + 1]);
    double h[4], X[3];
    r4_null_vector(m, h);
    dehomogenise(h, X);
    mask = (1u << a) | (1u << c);
    for (int v = 0; v < V; v++) {
      double e = reproj_err(Pb + v * 12, X, (double)kb[(int64_t)v * J * 2], (double)kb[(int64_t)v * J * 2 + 1]);
      if (e < eps) mask |= 1u << v;
    }
    count = __popc(mask);
  }
  // first pair with the strictly largest set: key = count * 64 + (63 - p), maximise
  int key = (p < n_pairs) ? count * 64 + (63 - p) : -1;
  int best = key;
  for (int o = PG >> 1; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, 64));
  unsigned win = (key == best) ? mask : 0u;
  for (int o = PG >> 1; o > 0; o >>= 1) win |= __shfl_xor(win, o, 64);

  // ---- stage 2: final DLT on the sorted inlier views ----------------------------------
  if (p != 0 || !live) return;
  double* X3 = kp3d + pr * 3;
  if (!is_valid) {
    X3[0] = X3[1] = X3[2] = 0.0;
    joint_err[pr] = 0.0;
    joint_inliers[pr] = 0;
    return;
  }
  R4 m;
  r4_zero(m);
  for (int v = 0; v < V; v++)
    if (win >> v & 1) add_view_rows(m, Pb + v * 12, (double)kb[(int64_t)v * J * 2], (double)kb[(int64_t)v * J * 2 + 1]);
  double h[4], X[3];
  r4_null_vector(m, h);
  dehomogenise(h, X);
  double errs[MAXV];
  int n = 0;
  for (int v = 0; v < V; v++)
    if (win >> v & 1)
      errs[n++] = reproj_err(Pb + v * 12, X, (double)kb[(int64_t)v * J * 2], (double)kb[(int64_t)v * J * 2 + 1]);
  X3[0] = X[0];
  X3[1] = X[1];
  X3[2] = X[2];
  joint_err[pr] = np_pairwise_sum(errs, n) / (double)n;
  joint_inliers[pr] = n;
}

// per frame: metric = np.mean(errs of valid joints), inlier_count = min (utils/triangulation.py:226,231)
__global__ void frame_reduce_kernel(const double* __restrict__ joint_err, const int32_t* __restrict__ joint_inliers,
                                    const uint8_t* __restrict__ valid, double* __restrict__ metric,
                                    int32_t* __restrict__ inlier_count, int B, int J) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double buf[512];
  int n = 0, mn = 0x7fffffff;
  for (int j = 0; j < J; j++) {
    if (valid && !valid[(int64_t)b * J + j]) continue;
    buf[n++] = joint_err[(int64_t)b * J + j];
    mn = min(mn, joint_inliers[(int64_t)b * J + j]);
  }
  metric[b] = n ? np_pairwise_sum(buf, n) / (double)n : NAN;
  inlier_count[b] = n ? mn : -1;
}

extern "C" int mval_triangulate_ransac(const void* kp2d, int kp_is_f32, const double* proj, const uint8_t* valid,
                                       double* kp3d, double* joint_err, int32_t* joint_inliers, double* metric,
                                       int32_t* inlier_count, int B, int V, int J, double eps, void* stream) {
  MVAL_REQUIRE(V >= 2, "mval_triangulate_ransac: need >= 2 views (reference asserts len(points) >= 2)");
  MVAL_REQUIRE(V <= MAXV, "mval_triangulate_ransac: V > 11 makes the reference sample pairs from python's RNG (unsupported)");
  MVAL_REQUIRE(J >= 1 && J <= 512 && B >= 0, "mval_triangulate_ransac: bad dims");
  if (B == 0) return 0;
  int n_pairs = V * (V - 1) / 2;
  int PG = 1;
  while (PG < n_pairs) PG <<= 1;
  int64_t n_prob = (int64_t)B * J;
  int per_block = 64 / PG;
  dim3 grid((unsigned)((n_prob + per_block - 1) / per_block));
  if (kp_is_f32)
    hipLaunchKernelGGL(ransac_dlt_kernel<float>, grid, dim3(64), 0, mval_stream(stream), (const float*)kp2d, proj,
                       valid, kp3d, joint_err, joint_inliers, n_prob, V, J, n_pairs, PG, eps);
  else
    hipLaunchKernelGGL(ransac_dlt_kernel<int64_t>, grid, dim3(64), 0, mval_stream(stream), (const int64_t*)kp2d, proj,
                       valid, kp3d, joint_err, joint_inliers, n_prob, V, J, n_pairs, PG, eps);
  MVAL_CHECK_LAUNCH("mval_triangulate_ransac");
  hipLaunchKernelGGL(frame_reduce_kernel, dim3((B + 63) / 64), dim3(64), 0, mval_stream(stream), joint_err,
                     joint_inliers, valid, metric, inlier_count, B, J);
  MVAL_CHECK_LAUNCH("mval_triangulate_ransac/reduce");
  return 0;
}

// ---- XE metric (utils/triangulation.py:236-257) ---------------------------------------
// one wave per (frame, view, joint) map; float64 like the reference's rendered target.
__global__ __launch_bounds__(256) void xe_kernel(const double* __restrict__ kp3d, const double* __restrict__ proj,
                                                 const float* __restrict__ hm, double* __restrict__ per_map,
                                                 int64_t n_maps, int V, int J, int hh, int wh, double sigma) {
  const int lane = threadIdx.x & 63;
  const int64_t map = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (map >= n_maps) return;
  const int j = (int)(map % J);
  const int v = (int)((map / J) % V);
  const int64_t b = map / ((int64_t)V * J);
  const double* P = proj + (b * V + v) * 12;
  const double* X = kp3d + (b * J + j) * 3;
  double u = X[0] * P[0] + X[1] * P[1] + X[2] * P[2] + P[3];
  double vv = X[0] * P[4] + X[1] * P[5] + X[2] * P[6] + P[7];
  double w = X[0] * P[8] + X[1] * P[9] + X[2] * P[10] + P[11];
  if (w == 0.0) w = 1.0;
  const double kx = u / w, ky = vv / w;
  const double inv = 1.0 / (2.0 * sigma * sigma);
  const float* p = hm + map * (int64_t)hh * wh;
  double acc = 0.0;
  for (int i = lane; i < hh * wh; i += 64) {
    int y = i / wh, x = i - y * wh;
    double dx = (double)x - kx, dy = (double)y - ky;
    double t = exp(-(dx * dx + dy * dy) * inv);
    double d = (double)p[i] - t;
    acc = fma(d, d, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) per_map[map] = acc / (double)(hh * wh);
}

__global__ void xe_sum_kernel(const double* __restrict__ per_map, double* __restrict__ out, int B, int VJ) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;  // the reference accumulates view-major, joint-minor, left to right
  for (int i = 0; i < VJ; i++) s += per_map[(int64_t)b * VJ + i];
  out[b] = s;
}

extern "C" int mval_reprojection_xe(const double* kp3d, const double* proj, const float* heatmaps, double* out,
                                    double* ws, int B, int V, int J, int hh, int wh, double sigma, void* stream) {
  MVAL_REQUIRE(B >= 0 && V > 0 && J > 0 && hh > 0 && wh > 0 && sigma > 0, "mval_reprojection_xe: bad dims");
  if (B == 0) return 0;
  int64_t n_maps = (int64_t)B * V * J;
  double* g_xe_ws = ws;
  hipLaunchKernelGGL(xe_kernel, dim3((unsigned)((n_maps + 3) / 4)), dim3(256), 0, mval_stream(stream), kp3d, proj,
                     heatmaps, g_xe_ws, n_maps, V, J, hh, wh, sigma);
  MVAL_CHECK_LAUNCH("mval_reprojection_xe");
  hipLaunchKernelGGL(xe_sum_kernel, dim3((B + 63) / 64), dim3(64), 0, mval_stream(stream), g_xe_ws, out, B, V * J);
  MVAL_CHECK_LAUNCH("mval_reprojection_xe/sum");
  return 0;
}
