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

// The same pass with the tile's operands in LDS (round 5; the kernel above stays as the fallback for spans that do not fit).  A workgroup
// = 64 output columns x 4 rows at a time (one wave per row) over PP_HROWS crop rows: the 64 columns' weights are staged ONCE, tap-major
// (k_s[t][column]: the lanes of a wave read consecutive words), and per row the source span of the tile -- the columns' windows overlap
// heavily: 64 columns at scale 2 read 140 source pixels -- is fetched with consecutive byte loads (zero-filled outside the image, as
// the box demands) instead of <= 64 x 3 strided byte loads and as many 4-byte weight loads from a 264-byte-strided table per thread.
// Integer sums are order-free: same bytes out.  128 views 512^2 -> 256^2: 439 -> see profiles/r05/pp_lds.log.
#define PP_HROWS 64
#define PP_SPAN 1024  // source pixels per tile row that fit (64 columns at scale <= ~15)
__global__ __launch_bounds__(256) void pp_horizontal_lds_kernel(const mval_view_desc* __restrict__ views, int in_w, int in_h,
                                                                const PpCoeff* __restrict__ co, unsigned char* __restrict__ tmp) {
  __shared__ int lo_s[64], n_s[64];
  __shared__ int k_s[PP_KMAX][64];
  __shared__ unsigned char px_s[4][PP_SPAN * 3];
  const int v = blockIdx.z;
  const mval_view_desc d = views[v];
  const int crop_h = d.bottom - d.top;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int xx = blockIdx.x * 64 + tx;
  const int omax = max(in_w, in_h);
  const bool col_ok = xx < in_w;
  const PpCoeff* c = co + (int64_t)(v * 2 + 0) * omax + (col_ok ? xx : in_w - 1);
  if (ty == 0) {
    lo_s[tx] = c->lo;
    n_s[tx] = col_ok ? c->n : 0;
  }
  const int n = col_ok ? c->n : 0;
  for (int t = ty; t < n; t += 4) k_s[t][tx] = c->k[t];
  __syncthreads();
  // the tile's source span: lo and lo + n do not decrease with the column
  const int last = min(63, in_w - 1 - (int)blockIdx.x * 64);
  const int span_lo = lo_s[0], span = lo_s[last] + n_s[last] - span_lo;
  const int my_off = (lo_s[tx] - span_lo) * 3;
  const int r_end = min(crop_h, ((int)blockIdx.y + 1) * PP_HROWS);
  if (span > PP_SPAN) {
    // The launcher sizes the staged row from the HOST's max_crop_w; the span comes from the view's DEVICE descriptor.  A view whose box is
    // wider than the host said must not write past px_s: its tiles read their taps straight from the image (pp_horizontal_kernel's loop:
    // same integer sums, same bytes out).  Uniform per workgroup (span is a function of shared values), so no barrier is skipped.
    for (int r = blockIdx.y * PP_HROWS + ty; r < r_end; r += 4) {
      if (!col_ok) continue;
      const int y = r + d.top;
      int s0 = 1 << (PP_BITS - 1), s1 = s0, s2 = s0;
      if (y >= 0 && y < d.h0) {
        const unsigned char* row = d.img + (int64_t)y * d.w0 * 3;
        for (int t = 0; t < n; t++) {
          const int x = lo_s[tx] + t + d.left;
          if (x < 0 || x >= d.w0) continue;
          const int k = k_s[t][tx];
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
    return;
  }
  for (int r0 = blockIdx.y * PP_HROWS; r0 < r_end; r0 += 4) {
    const int r = r0 + ty, y = r + d.top;
    const bool row_ok = r < crop_h && y >= 0 && y < d.h0;
    const unsigned char* row = d.img + (int64_t)(row_ok ? y : 0) * d.w0 * 3;
    const int x0 = span_lo + d.left;  // source column of the span's first pixel
    // eight loads in flight per lane and round trip (one at a time, the 7 dependent round trips of a 420-byte span were the kernel: 4 us per row)
    for (int j0 = tx; j0 < span * 3; j0 += 8 * 64) {
      unsigned char b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u * 64, x = x0 + j / 3;
        b[u] = (j < span * 3 && row_ok && x >= 0 && x < d.w0) ? row[(int64_t)x0 * 3 + j] : (unsigned char)0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (j0 + u * 64 < span * 3) px_s[ty][j0 + u * 64] = b[u];
    }
    // (no workgroup barrier in this loop: row `ty` of px_s belongs to ONE wave -- blockDim.x = 64 -- and a wave's LDS operations execute
    // in order, so its reads below see its stores above and the next trip's stores come after these reads; with barriers the four waves
    // waited for each other's global loads every trip: 226 vs see pp_lds.log)
    __builtin_amdgcn_wave_barrier();
    if (r < crop_h && col_ok) {
      int s0 = 1 << (PP_BITS - 1), s1 = s0, s2 = s0;
      const unsigned char* p = px_s[ty] + my_off;
      for (int t = 0; t < n; t++) {
        const int k = k_s[t][tx];
        s0 += p[t * 3] * k;
        s1 += p[t * 3 + 1] * k;
        s2 += p[t * 3 + 2] * k;
      }
      unsigned char* o = tmp + d.tmp_off + ((int64_t)r * in_w + xx) * 3;
      o[0] = (unsigned char)pp_clip8(s0);
      o[1] = (unsigned char)pp_clip8(s1);
      o[2] = (unsigned char)pp_clip8(s2);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// out[v][c][yy][xx] = (bgr[c] / 255.0 - mean[c]) / std[c], bgr[c] = resized raw channel 2 - c
__global__ void pp_lut_kernel(float* __restrict__ lut_g) {  // <<<3, 256>>>
  const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
  const int ch = blockIdx.x, u = threadIdx.x;
  lut_g[ch * 256 + u] = (float)(((double)u / 255.0 - mean[ch]) / stdv[ch]);
}

__global__ __launch_bounds__(256) void pp_vertical_kernel(const mval_view_desc* __restrict__ views, int in_w, int in_h,
                                   const PpCoeff* __restrict__ co, const unsigned char* __restrict__ tmp,
                                   float* __restrict__ out, const float* __restrict__ lut_g) {
  // the normalisation of a byte has 256 values per channel: evaluated once per call with numpy's float64 expression (pp_lut_kernel; three
  // float64 divisions per output pixel before), then looked up
  __shared__ float lut[3][256];
  {
    const int u = threadIdx.y * blockDim.x + threadIdx.x;  // 256 threads
#pragma unroll
    for (int ch = 0; ch < 3; ch++) lut[ch][u] = lut_g[ch * 256 + u];
  }
  __syncthreads();
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
#pragma unroll
  for (int ch = 0; ch < 3; ch++) out[(((int64_t)v * 3 + ch) * in_h + yy) * in_w + xx] = lut[ch][pp_clip8(s[2 - ch])];
}

// The vertical pass with 4-byte loads (in_w * 3 a multiple of 4: every network input size in use).  The weights of an output row are the
// same for all its bytes, so a thread owns FOUR consecutive bytes of the temp rows -- 12 dword loads per four output values instead of 36
// byte loads per three (a byte load costs the texture path what a dword load does) -- and scatters them to their (channel, column).
__global__ __launch_bounds__(256) void pp_vertical4_kernel(const mval_view_desc* __restrict__ views, int in_w, int in_h,
                                                           const PpCoeff* __restrict__ co, const unsigned char* __restrict__ tmp,
                                                           float* __restrict__ out, const float* __restrict__ lut_g) {
  __shared__ float lut[3][256];
  {
    const int u = threadIdx.y * blockDim.x + threadIdx.x;  // 256 threads
#pragma unroll
    for (int ch = 0; ch < 3; ch++) lut[ch][u] = lut_g[ch * 256 + u];
  }
  __syncthreads();
  const int v = blockIdx.z;
  const mval_view_desc d = views[v];
  const int row_b = in_w * 3, q = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y * blockDim.y + threadIdx.y;
  if (yy >= in_h || q * 4 >= row_b) return;
  const int omax = max(in_w, in_h);
  const PpCoeff& c = co[(v * 2 + 1) * omax + yy];
  int s[4] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
  const unsigned char* col = tmp + d.tmp_off + (int64_t)q * 4;
  const bool aligned = (reinterpret_cast<uintptr_t>(tmp + d.tmp_off) & 3) == 0;  // (a caller's own slab offsets need not be)
  for (int t = 0; t < c.n; t++) {
    const unsigned char* pw = col + (int64_t)(c.lo + t) * row_b;
    const unsigned w = aligned ? *reinterpret_cast<const unsigned*>(pw) : (unsigned)pw[0] | ((unsigned)pw[1] << 8) | ((unsigned)pw[2] << 16) | ((unsigned)pw[3] << 24);
    const int k = c.k[t];
    s[0] += (int)(w & 255u) * k;
    s[1] += (int)((w >> 8) & 255u) * k;
    s[2] += (int)((w >> 16) & 255u) * k;
    s[3] += (int)(w >> 24) * k;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int b = q * 4 + j, xx = b / 3, ch = 2 - (b - xx * 3);  // raw channel b % 3 is the output's channel 2 - that
    out[(((int64_t)v * 3 + ch) * in_h + yy) * in_w + xx] = lut[ch][pp_clip8(s[j])];
  }
}

extern "C" size_t mval_prepare_views_workspace_bytes(int n_views, int64_t total_crop_rows, int in_w, int in_h) {
  const size_t omax = in_w > in_h ? in_w : in_h;
  return (size_t)n_views * 2 * omax * sizeof(PpCoeff) + (size_t)total_crop_rows * in_w * 3 + 256 + 256 + 3 * 256 * sizeof(float);
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
  float* lut_g = reinterpret_cast<float*>(ws);
  char* ws1 = reinterpret_cast<char*>(ws) + 3 * 256 * sizeof(float);  // (3 072 bytes: the rest stays 256-byte aligned)
  PpCoeff* co = reinterpret_cast<PpCoeff*>(ws1);
  unsigned char* tmp = reinterpret_cast<unsigned char*>(ws1) + (((size_t)n_views * 2 * omax * sizeof(PpCoeff) + 255) & ~(size_t)255);
  const int nco = n_views * 2 * omax;
  // (the normalisation table sits in front of the coefficients: the first 3 KB of ws; the temp part's size depends on the caller's rows)
  hipLaunchKernelGGL(pp_lut_kernel, dim3(3), dim3(256), 0, s, lut_g);
  hipLaunchKernelGGL(pp_coeff_kernel, dim3((nco + 127) / 128), dim3(128), 0, s, views, n_views, in_w, in_h, co);
  MVAL_CHECK_LAUNCH("mval_prepare_views/coeff");
  // the source span of 64 output columns: 64 * scale + the filter support on both sides (+ rounding)
  if ((int)ceil(64.0 * sx + 6.0 * (sx < 1.0 ? 1.0 : sx)) + 4 <= PP_SPAN)
    hipLaunchKernelGGL(pp_horizontal_lds_kernel, dim3((in_w + 63) / 64, (max_crop_h + PP_HROWS - 1) / PP_HROWS, n_views), dim3(64, 4), 0, s, views,
                       in_w, in_h, co, tmp);
  else
    hipLaunchKernelGGL(pp_horizontal_kernel, dim3((in_w + 63) / 64, (max_crop_h + 3) / 4, n_views), dim3(64, 4), 0, s, views,
                       in_w, in_h, max_crop_h, co, tmp);
  MVAL_CHECK_LAUNCH("mval_prepare_views/horizontal");
  // (tmp_off of every view is a multiple of 4 when in_w * 3 is: the caller packs the views' [crop_h][in_w][3] slabs back to back)
  if ((in_w * 3) % 4 == 0)
    hipLaunchKernelGGL(pp_vertical4_kernel, dim3((in_w * 3 / 4 + 63) / 64, (in_h + 3) / 4, n_views), dim3(64, 4), 0, s, views, in_w, in_h, co, tmp, out,
                       lut_g);
  else
    hipLaunchKernelGGL(pp_vertical_kernel, dim3((in_w + 63) / 64, (in_h + 3) / 4, n_views), dim3(64, 4), 0, s, views, in_w,
                       in_h, co, tmp, out, lut_g);
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
