// Per-view input pipeline on the device: the pixel work of the reference's
// ``ActiveLearningDataset.prepare_single_view`` (dataset/dataset.py:158-220) -- BGR flip, zero-filled
// crop to the square box (utils/triangulation.py:77-93), PIL LANCZOS resize to the network input
// (dataset.py:208-211), ImageNet normalisation (utils/triangulation.py:137-145) -- and the Gaussian
// ground-truth heat-maps (dataset.py:198-207).  JPEG decode and the 3x3 camera bookkeeping stay on
// the host.
//
// The resize follows Pillow's ImagingResample for 8-bit images exactly (see oracle/preprocess.py,
// which is pinned bit-for-bit against PIL): separable, horizontal pass first, float64 Lanczos-3
// weights normalised per output pixel and converted to 22-bit fixed point, integer accumulation
// from 2^21, >> 22 and clamp after EACH pass.  The crop is virtual: the horizontal pass reads the raw
// image through the box and substitutes zeros outside it.
//   kernel 1  coefficients: one thread per (view, axis, output index): source range + <= 64 weights
//   kernel 2  horizontal pass: raw u8 (H0, W0, 3) -> temp u8 [crop rows][in_w][3]
//   kernel 3  vertical pass + flip + normalise (float64, as numpy does) -> float32 (3, in_h, in_w)
// HBM-bound streaming: a view moves crop_h * crop_w * 3 bytes in and in_h * in_w * 12 bytes out.
#include "mval_common.h"

#define PP_BITS 22
#define PP_KMAX 64

struct PpCoeff {  // per (view, axis, output index)
  int lo, n;
  int k[PP_KMAX];
};

__device__ __forceinline__ double pp_sinc(double x) {
  if (x == 0.0) return 1.0;
  x = x * 3.14159265358979323846;
  return sin(x) / x;
}
__device__ __forceinline__ double pp_lanczos(double x) {
  return (-3.0 <= x && x < 3.0) ? pp_sinc(x) * pp_sinc(x / 3) : 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size)
__global__ void pp_coeff_kernel(const mval_view_desc* __restrict__ views, int n_views, int in_w, int in_h,
                                PpCoeff* __restrict__ co) {
  const int omax = max(in_w, in_h);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_views * 2 * omax) return;
  const int xx = i % omax, axis = (i / omax) & 1, v = i / (2 * omax);
  const int out_size = axis ? in_h : in_w;
  if (xx >= out_size) return;
  const mval_view_desc d = views[v];
  const int in_size = axis ? d.bottom - d.top : d.right - d.left;
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 3.0 * filterscale, ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > PP_KMAX) xmax = PP_KMAX;  // the launcher rejects boxes that need more taps
  PpCoeff& c = co[i];
  double w[PP_KMAX];
  double ww = 0.0;
  for (int x = 0; x < xmax; x++) {
    w[x] = pp_lanczos((x + xmin - center + 0.5) * ss);
    ww += w[x];
  }
  for (int x = 0; x < xmax; x++) {
    const double k = ww != 0.0 ? w[x] / ww : w[x];
    c.k[x] = k < 0 ? (int)(-0.5 + k * (1 << PP_BITS)) : (int)(0.5 + k * (1 << PP_BITS));
  }
  c.lo = xmin;
  c.n = xmax;
}

__device__ __forceinline__ int pp_clip8(int v) {
  v >>= PP_BITS;
  return v < 0 ? 0 : v > 255 ? 255 : v;
}

// temp[row][xx][c] for every crop row; reads the raw image through the (zero-filled) box
__global__ void pp_horizontal_kernel(const mval_view_desc* __restrict__ views, int in_w, int in_h, int max_crop_h,
                                     const PpCoeff* __restrict__ co, unsigned char* __restrict__ tmp) {
  const int v = blockIdx.z;
  const mval_view_desc d = views[v];
  const int crop_h = d.bottom - d.top;
  const int r = blockIdx.y * blockDim.y + threadIdx.y, xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= crop_h || xx >= in_w) return;
  const int omax = max(in_w, in_h);
  const PpCoeff& c = co[(v * 2 + 0) * omax + xx];
  const int y = r + d.top;
  int s0 = 1 << (PP_BITS - 1), s1 = s0, s2 = s0;
  if (y >= 0 && y < d.h0) {
    const unsigned char* row = d.img + (int64_t)y * d.w0 * 3;
    for (int t = 0; t < c.n; t++) {
      const int x = c.lo + t + d.left;
      if (x < 0 || x >= d.w0) continue;
      const int k = c.k[t];
      s0 += row[x * 3] * k;
      s1 += row[x * 3 + 1] * k;
      s2 += row[x * 3 + 2] * k;
    }
  }
  unsigned char* o = tmp + d.tmp_off + ((int64_t)r * in_w + xx) * 3;
  o[0] = (unsigned char)pp_clip8(s0);
  o[1] = (unsigned char)pp_clip8(s1);
  o[2] = (unsigned char)pp_clip8(s2);
}

// out[v][c][yy][xx] = (bgr[c] / 255.0 - mean[c]) / std[c], bgr[c] = resized raw channel 2 - c
__global__ void pp_vertical_kernel(const mval_view_desc* __restrict__ views, int in_w, int in_h,
                                   const PpCoeff* __restrict__ co, const unsigned char* __restrict__ tmp,
                                   float* __restrict__ out) {
  const int v = blockIdx.z;
  const mval_view_desc d = views[v];
  const int yy = blockIdx.y * blockDim.y + threadIdx.y, xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (yy >= in_h || xx >= in_w) return;
  const int omax = max(in_w, in_h);
  const PpCoeff& c = co[(v * 2 + 1) * omax + yy];
  int s[3] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
  const unsigned char* col = tmp + d.tmp_off + (int64_t)xx * 3;
  for (int t = 0; t < c.n; t++) {
    const unsigned char* px = col + (int64_t)(c.lo + t) * in_w * 3;
    const int k = c.k[t];
    s[0] += px[0] * k;
    s[1] += px[1] * k;
    s[2] += px[2] * k;
  }
  const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
    const double u = (double)pp_clip8(s[2 - ch]);
    out[(((int64_t)v * 3 + ch) * in_h + yy) * in_w + xx] = (float)((u / 255.0 - mean[ch]) / stdv[ch]);
  }
}

extern "C" size_t mval_prepare_views_workspace_bytes(int n_views, int64_t total_crop_rows, int in_w, int in_h) {
  const size_t omax = in_w > in_h ? in_w : in_h;
  return (size_t)n_views * 2 * omax * sizeof(PpCoeff) + (size_t)total_crop_rows * in_w * 3 + 256;
}

// views: n_views descriptors in DEVICE memory (square boxes; tmp_off = byte offset of the view's
// [crop_h][in_w][3] slab inside the temp part of ws); max_crop_h / max_crop_w bound the grid and the
// filter support (<= 64 taps: boxes up to ~10x the input size).
extern "C" int mval_prepare_views(const mval_view_desc* views, int n_views, int max_crop_h, int max_crop_w, int in_w,
                                  int in_h, float* out, void* ws, void* stream) {
  MVAL_REQUIRE(views && out && ws && n_views > 0 && in_w > 0 && in_h > 0 && max_crop_h > 0 && max_crop_w > 0,
               "mval_prepare_views: bad arguments");
  const double sx = (double)max_crop_w / in_w, sy = (double)max_crop_h / in_h;
  const double smax = sx > sy ? sx : sy;
  MVAL_REQUIRE((int)ceil(3.0 * (smax < 1.0 ? 1.0 : smax)) * 2 + 1 <= PP_KMAX,
               "mval_prepare_views: a %d x %d box needs more than %d filter taps for a %d x %d input", max_crop_w,
               max_crop_h, PP_KMAX, in_w, in_h);
  hipStream_t s = mval_stream(stream);
  const int omax = in_w > in_h ? in_w : in_h;
  PpCoeff* co = reinterpret_cast<PpCoeff*>(ws);
  unsigned char* tmp = reinterpret_cast<unsigned char*>(ws) + (((size_t)n_views * 2 * omax * sizeof(PpCoeff) + 255) & ~(size_t)255);
  const int nco = n_views * 2 * omax;
  hipLaunchKernelGGL(pp_coeff_kernel, dim3((nco + 127) / 128), dim3(128), 0, s, views, n_views, in_w, in_h, co);
  MVAL_CHECK_LAUNCH("mval_prepare_views/coeff");
  hipLaunchKernelGGL(pp_horizontal_kernel, dim3((in_w + 63) / 64, (max_crop_h + 3) / 4, n_views), dim3(64, 4), 0, s, views,
                     in_w, in_h, max_crop_h, co, tmp);
  MVAL_CHECK_LAUNCH("mval_prepare_views/horizontal");
  hipLaunchKernelGGL(pp_vertical_kernel, dim3((in_w + 63) / 64, (in_h + 3) / 4, n_views), dim3(64, 4), 0, s, views, in_w,
                     in_h, co, tmp, out);
  MVAL_CHECK_LAUNCH("mval_prepare_views/vertical");
  return 0;
}

// Gaussian ground-truth heat-maps (dataset.py:198-207): pt [n, 2] float64 in heat-map pixels ->
// out [n, h, w] float32 = (float) exp(-((x - px)^2 + (y - py)^2) / (2 sigma^2)), all in float64.
__global__ void pp_gt_heatmap_kernel(const double* __restrict__ pt, double two_s2, int h, int w, int64_t total,
                                     float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % w), y = (int)((i / w) % h);
  const int64_t n = i / ((int64_t)w * h);
  const double dx = (double)x - pt[n * 2], dy = (double)y - pt[n * 2 + 1];
  // each float64 operation rounded separately, as torch evaluates sum((grid - labels) ** 2) / (2 sigma^2)
  out[i] = (float)exp(-__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)) / two_s2);
}

extern "C" int mval_gt_heatmaps(const double* pt, int64_t n, double sigma, int h, int w, float* out, void* stream) {
  MVAL_REQUIRE(pt && out && n > 0 && h > 0 && w > 0 && sigma > 0, "mval_gt_heatmaps: bad arguments");
  const int64_t total = n * h * w;
  hipLaunchKernelGGL(pp_gt_heatmap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mval_stream(stream), pt,
                     2.0 * (sigma * sigma), h, w, total, out);
  MVAL_CHECK_LAUNCH("mval_gt_heatmaps");
  return 0;
}
