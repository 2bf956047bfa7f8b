// Heat-map network runner: weight packing, eval-mode BatchNorm folding, the generic
// direct (VALU) operators, and the op-list executor behind mval_net_forward.
//
// The direct kernels cover the operators that are not GEMM-shaped enough for the matrix
// cores (3-channel stems, max-pool) or not yet ported to them (transposed conv); they are
// also the on-device cross-check for the MFMA kernels (MVAL_FORCE_DIRECT=1).
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "conv_common.h"
#include "conv_p2.h"

// ---- weight packing ---------------------------------------------------------------------
extern "C" size_t mval_packed_weight_floats(int pack, int cout, int cin, int k) {
  if (pack == MVAL_PACK_HWIO) return (size_t)k * k * cin * cout;
  if (pack == MVAL_PACK_MFMA16_BF3) return (size_t)k * k * ((cin + 31) / 32) * ((cout + 15) / 16) * 768;
  // two fp16 planes per block + the 4-float trailer [2^-s, bits of max |w|, 0, 0] (conv_mfma_split.hip)
  if (pack == MVAL_PACK_MFMA16_H2) return (size_t)k * k * ((cin + 31) / 32) * ((cout + 15) / 16) * 512 + 4;
  size_t g = (cin + 15) / 16, ns = (cout + 15) / 16;
  return (size_t)k * k * g * ns * 256;
}

__device__ __forceinline__ float w_at(const float* w, int transposed, int cout, int cin, int k, int co, int ci, int t) {
  // 0: Conv2d [cout][cin][k][k] ; 1: ConvTranspose2d [cin][cout][k][k] ;
  // 2: data-gradient form of a Conv2d whose weight is [cin'=cout][cout'=cin][k][k]: swap the
  //    channel roles and flip the taps (dx = conv(dz, W^T flipped))
  if (transposed == 2) return w[((int64_t)ci * cout + co) * k * k + (k * k - 1 - t)];
  return transposed ? w[((int64_t)ci * cout + co) * k * k + t] : w[((int64_t)co * cin + ci) * k * k + t];
}

__global__ void pack_hwio_kernel(const float* __restrict__ w, float* __restrict__ p, int transposed, int cout, int cin,
                                 int k) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)k * k * cin * cout;
  if (i >= total) return;
  int co = (int)(i % cout);
  int ci = (int)((i / cout) % cin);
  int t = (int)(i / ((int64_t)cout * cin));
  p[i] = w_at(w, transposed, cout, cin, k, co, ci, t);
}

__global__ void pack_mfma16_kernel(const float* __restrict__ w, float* __restrict__ p, int transposed, int cout,
                                   int cin, int k) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int G = (cin + 15) / 16, NS = (cout + 15) / 16;
  int64_t total = (int64_t)k * k * G * NS * 256;
  if (i >= total) return;
  int j = (int)(i & 3);
  int lane = (int)((i >> 2) & 63);
  int64_t r = i >> 8;
  int ns = (int)(r % NS);
  int g = (int)((r / NS) % G);
  int t = (int)(r / ((int64_t)NS * G));
  int co = ns * 16 + (lane & 15);
  int ci = g * 16 + (lane >> 4) * 4 + j;
  p[i] = (co < cout && ci < cin) ? w_at(w, transposed, cout, cin, k, co, ci, t) : 0.f;
}

extern "C" int mval_pack_bf3_jobs(const mval_pack_job* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks,
                                 void* stream) {
  MVAL_REQUIRE(jobs_dev && first_block_dev && n_jobs > 0 && total_blocks > 0, "mval_pack_bf3_jobs: bad arguments");
  mval_pack_bf3_batch(jobs_dev, first_block_dev, n_jobs, total_blocks, mval_stream(stream));
  MVAL_CHECK_LAUNCH("mval_pack_bf3_jobs");
  return 0;
}

extern "C" int mval_pack_split_jobs(const mval_pack_job* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks,
                                   void* stream) {
  MVAL_REQUIRE(jobs_dev && first_block_dev && n_jobs > 0 && total_blocks > 0, "mval_pack_split_jobs: bad arguments");
  mval_pack_split_batch(jobs_dev, first_block_dev, n_jobs, total_blocks, mval_stream(stream));
  MVAL_CHECK_LAUNCH("mval_pack_split_jobs");
  return 0;
}

extern "C" int mval_pack_conv_weights(int pack, int transposed, const float* w, float* packed, int cout, int cin, int k,
                                      void* stream) {
  MVAL_REQUIRE(cout > 0 && cin > 0 && k > 0, "mval_pack_conv_weights: bad dims");
  if (pack == MVAL_PACK_MFMA16_BF3) {
    mval_pack_bf3(transposed, w, packed, cout, cin, k, mval_stream(stream));
    MVAL_CHECK_LAUNCH("mval_pack_conv_weights/bf3");
    return 0;
  }
  if (pack == MVAL_PACK_MFMA16_H2) {
    mval_pack_h2(transposed, w, packed, cout, cin, k, mval_stream(stream));
    MVAL_CHECK_LAUNCH("mval_pack_conv_weights/h2");
    return 0;
  }
  int64_t total = (int64_t)mval_packed_weight_floats(pack, cout, cin, k);
  dim3 grid((unsigned)((total + 255) / 256));
  if (pack == MVAL_PACK_HWIO)
    hipLaunchKernelGGL(pack_hwio_kernel, grid, dim3(256), 0, mval_stream(stream), w, packed, transposed, cout, cin, k);
  else if (pack == MVAL_PACK_MFMA16)
    hipLaunchKernelGGL(pack_mfma16_kernel, grid, dim3(256), 0, mval_stream(stream), w, packed, transposed, cout, cin,
                       k);
  else
    MVAL_REQUIRE(false, "mval_pack_conv_weights: unknown packing %d", pack);
  MVAL_CHECK_LAUNCH("mval_pack_conv_weights");
  return 0;
}

// torch's inference formula (aten batch_norm_cpu_transform_input):
//   alpha = gamma / sqrt(var + eps) ; y = x * alpha + (beta - mean * alpha)
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               float* scale, float* shift, int c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  float invstd = 1.0f / sqrtf(var[i] + eps);
  float a = gamma[i] * invstd;
  scale[i] = a;
  shift[i] = beta[i] - mean[i] * a;
}

extern "C" int mval_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                            float* scale, float* shift, int c, void* stream) {
  MVAL_REQUIRE(c > 0, "mval_bn_fold: bad dims");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, mval_stream(stream), gamma, beta, mean, var,
                     eps, scale, shift, c);
  MVAL_CHECK_LAUNCH("mval_bn_fold");
  return 0;
}

// ---- direct operators -------------------------------------------------------------------
#define DC_CO 4  // couts per thread

// thread = (output pixel, group of DC_CO couts); weights HWIO so a wave reads contiguous couts
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  const int cog = (a.Cout + DC_CO - 1) / DC_CO;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = (int64_t)a.N * a.Hout * a.Wout * cog;
  if (t >= total) return;
  const int cg = (int)(t % cog);
  int64_t p = t / cog;
  const int x = (int)(p % a.Wout);
  const int y = (int)((p / a.Wout) % a.Hout);
  const int n = (int)(p / ((int64_t)a.Wout * a.Hout));
  const int c0 = cg * DC_CO;
  float acc[DC_CO];
#pragma unroll
  for (int q = 0; q < DC_CO; q++) acc[q] = 0.f;
  for (int ky = 0; ky < a.k; ky++) {
    const int iy = y * a.stride - a.pad + ky;
    if (iy < 0 || iy >= a.Hin) continue;
    for (int kx = 0; kx < a.k; kx++) {
      const int ix = x * a.stride - a.pad + kx;
      if (ix < 0 || ix >= a.Win) continue;
      const float* wp = a.w + ((int64_t)(ky * a.k + kx) * a.Cin) * a.Cout + c0;
      for (int ci = 0; ci < a.Cin; ci++) {
        const float v = a.in_nchw ? a.in[(((int64_t)n * a.Cin + ci) * a.Hin + iy) * a.Win + ix]
                                  : a.in[(((int64_t)n * a.Hin + iy) * a.Win + ix) * a.Cin + ci];
#pragma unroll
        for (int q = 0; q < DC_CO; q++)
          if (c0 + q < a.Cout) acc[q] = fmaf(v, wp[(int64_t)ci * a.Cout + q], acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < DC_CO; q++)
    if (c0 + q < a.Cout) conv_store(a, n, y, x, c0 + q, acc[q] * a.scale[c0 + q] + a.shift[c0 + q]);
}

// ConvTranspose2d(k, stride, pad): out[oy] gathers iy with oy = iy*stride - pad + ky
__global__ __launch_bounds__(256) void deconv_direct_kernel(ConvArgs a) {
  const int cog = (a.Cout + DC_CO - 1) / DC_CO;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = (int64_t)a.N * a.Hout * a.Wout * cog;
  if (t >= total) return;
  const int cg = (int)(t % cog);
  int64_t p = t / cog;
  const int x = (int)(p % a.Wout);
  const int y = (int)((p / a.Wout) % a.Hout);
  const int n = (int)(p / ((int64_t)a.Wout * a.Hout));
  const int c0 = cg * DC_CO;
  float acc[DC_CO];
#pragma unroll
  for (int q = 0; q < DC_CO; q++) acc[q] = 0.f;
  for (int ky = 0; ky < a.k; ky++) {
    const int ty = y + a.pad - ky;
    if (ty < 0 || ty % a.stride) continue;
    const int iy = ty / a.stride;
    if (iy >= a.Hin) continue;
    for (int kx = 0; kx < a.k; kx++) {
      const int tx = x + a.pad - kx;
      if (tx < 0 || tx % a.stride) continue;
      const int ix = tx / a.stride;
      if (ix >= a.Win) continue;
      const float* wp = a.w + ((int64_t)(ky * a.k + kx) * a.Cin) * a.Cout + c0;
      const float* ip = a.in + (((int64_t)n * a.Hin + iy) * a.Win + ix) * a.Cin;
      for (int ci = 0; ci < a.Cin; ci++) {
        const float v = ip[ci];
#pragma unroll
        for (int q = 0; q < DC_CO; q++)
          if (c0 + q < a.Cout) acc[q] = fmaf(v, wp[(int64_t)ci * a.Cout + q], acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < DC_CO; q++)
    if (c0 + q < a.Cout) conv_store(a, n, y, x, c0 + q, acc[q] * a.scale[c0 + q] + a.shift[c0 + q]);
}

// MaxPool2d(k, stride, pad) on NHWC (padding never wins: -inf)
__global__ __launch_bounds__(256) void maxpool_kernel(ConvArgs a) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = (int64_t)a.N * a.Hout * a.Wout * a.Cout;
  if (t >= total) return;
  const int c = (int)(t % a.Cout);
  int64_t p = t / a.Cout;
  const int x = (int)(p % a.Wout);
  const int y = (int)((p / a.Wout) % a.Hout);
  const int n = (int)(p / ((int64_t)a.Wout * a.Hout));
  float m = -INFINITY;
  for (int ky = 0; ky < a.k; ky++) {
    const int iy = y * a.stride - a.pad + ky;
    if (iy < 0 || iy >= a.Hin) continue;
    for (int kx = 0; kx < a.k; kx++) {
      const int ix = x * a.stride - a.pad + kx;
      if (ix < 0 || ix >= a.Win) continue;
      float v = a.in[(((int64_t)n * a.Hin + iy) * a.Win + ix) * a.Cin + c];
      m = (v > m || v != v) ? v : m;
    }
  }
  a.out[t] = m;
}

// slots[i] = max(slots[i], max |x| over image i): the per-image activation scale of the fp16-split convs for tensors
// whose producer does not keep it itself (max-pool, the generic direct kernels) and for single-op launches.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t per_image, int vec, unsigned* slots) {

  const float* xi = x + (int64_t)blockIdx.y * per_image;
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = vec ? per_image / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const conv_f32x4 v = reinterpret_cast<const conv_f32x4*>(xi)[i];
    m = conv_amax4(m, v.x, v.y, v.z, v.w);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_image; i += stride) m = fmaxf(m, fabsf(xi[i]));
  conv_amax_commit(slots + (int64_t)blockIdx.y * MVAL_AMAX_ROW, (int)blockIdx.x, (int)gridDim.x, m);
}

__global__ void zero_rows_kernel(unsigned* rows, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) rows[i] = 0u;
}
void mval_launch_zero_rows(unsigned* rows, int64_t n_dwords, hipStream_t s) {
  int64_t blocks = (n_dwords + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows, n_dwords);
}

int mval_launch_amax(const float* x, int64_t per_image, int n_images, unsigned* slots, hipStream_t s) {
  const int vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (per_image & 3) == 0;
  int64_t blocks = (per_image / 4 + 255) / 256;
  if (blocks > 64) blocks = 64;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks, (unsigned)n_images), dim3(256), 0, s, x, per_image, vec, slots);
  return 0;
}

extern "C" int mval_amax(const float* x, int64_t per_image, int n_images, uint32_t* slots, void* stream) {
  MVAL_REQUIRE(x && slots && per_image > 0 && n_images > 0, "mval_amax: bad arguments");
  mval_launch_amax(x, per_image, n_images, slots, mval_stream(stream));
  MVAL_CHECK_LAUNCH("mval_amax");
  return 0;
}

int mval_launch_conv_direct(const ConvArgs& a, int kind, hipStream_t s) {
  if (kind == MVAL_OP_MAXPOOL) {
    int64_t total = (int64_t)a.N * a.Hout * a.Wout * a.Cout;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return 0;
  }
  const int cog = (a.Cout + DC_CO - 1) / DC_CO;
  int64_t total = (int64_t)a.N * a.Hout * a.Wout * cog;
  dim3 grid((unsigned)((total + 255) / 256));
  if (kind == MVAL_OP_DECONV)
    hipLaunchKernelGGL(deconv_direct_kernel, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), 0, s, a);
  return 0;
}

// ---- op executor ------------------------------------------------------------------------
static int force_direct() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MVAL_FORCE_DIRECT");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v;
}

static void fill_geometry(ConvArgs& a, const mval_op* op, int n_images) {
  a.N = n_images;
  a.Hin = op->hin; a.Win = op->win; a.Cin = op->cin;
  a.Hout = op->hout; a.Wout = op->wout; a.Cout = op->cout;
  a.k = op->k; a.stride = op->stride; a.pad = op->pad;
  a.up = op->up; a.relu = op->relu; a.in_nchw = op->in_nchw; a.out_nchw = op->out_nchw;
  a.dil = 1;
  a.th = a.tw = a.tn = a.tw_log2 = a.thw_log2 = a.tiles_x = a.tiles_y = 0;
  a.G_total = (op->cin + 15) / 16;
  a.NS_total = (op->cout + 15) / 16;
}

// ConvTranspose2d on the matrix cores = stride-1 conv over the zero-dilated input with pad k-1-p
// (weights packed with mode 2: tap-flipped, channel-swapped)
static void deconv_as_conv(ConvArgs& a, const mval_op* op) {
  a.dil = op->stride;
  a.stride = 1;
  a.pad = op->k - 1 - op->pad;
}

extern "C" int mval_op_mfma_supported(const mval_op* op, int n_images) {
  if (!op || (op->kind != MVAL_OP_CONV && op->kind != MVAL_OP_DECONV) || n_images <= 0) return 0;
  ConvArgs a = {};
  a.in = a.w = a.scale = a.shift = a.res1 = a.res2 = nullptr;
  a.out = nullptr;
  fill_geometry(a, op, n_images);
  if (op->kind == MVAL_OP_DECONV) deconv_as_conv(a, op);
  return mval_conv_mfma_supported(a);
}

// ConvTranspose2d(k4, s2, p1) on the split-bf16 kernel: four 2x2 stride-1 convs, one per output parity
// (py, px), each over the (hin, win) grid and scattered to rows 2a + py, columns 2b + px.  Weights:
// pack mode 3 ([parity][2x2 taps]...), one parity = a quarter of the k = 4 packed buffer.
static size_t deconv_parity_floats(const mval_op* op) {  // one parity's 2x2 taps in the packed k4 buffer
  const int pack = op->algo == MVAL_ALGO_MFMA_H2 ? MVAL_PACK_MFMA16_H2 : MVAL_PACK_MFMA16_BF3;
  return (mval_packed_weight_floats(pack, op->cout, op->cin, 4) & ~(size_t)7) / 4;  // (the h2 trailer is not a parity's)
}
static void deconv_parity(ConvArgs& a, const mval_op* op, int parity, const float* w_packed) {
  a.k = 2; a.stride = 1; a.pad = 1; a.dil = 1;
  a.Hout = op->hin; a.Wout = op->win;
  a.org_dy = parity >> 1; a.org_dx = parity & 1;
  a.os_log2 = 1; a.ooy = parity >> 1; a.oox = parity & 1;
  if (w_packed) a.w = w_packed + (size_t)parity * deconv_parity_floats(op);
}

extern "C" int mval_op_algo_supported(const mval_op* op, int n_images, int algo) {
  if (op && op->kind == MVAL_OP_FUSE_UP)
    return algo == MVAL_ALGO_MFMA_P2 && n_images > 0 && !op->in_nchw && !op->out_nchw && op->res1_off >= 0 && op->res2_off < 0 &&
           mval_conv_fuse_up_p2_supported(op->cout, op->n_terms, op->t_cin, op->t_up, n_images, op->hout, op->wout);
  if (op && op->kind == MVAL_OP_STEM_P2)
    return algo == MVAL_ALGO_MFMA_P2 && n_images > 0 && op->cin == 3 && op->cout == 64 && op->in_nchw && !op->out_nchw && !op->up && op->relu &&
           op->hout * 4 == op->hin && op->wout * 4 == op->win && mval_conv_stem_p2_supported(n_images, op->hin, op->win);
  if (op && op->kind == MVAL_OP_BNECK)
    return algo == MVAL_ALGO_MFMA_P2 && n_images > 0 && op->cout == 256 && op->stride == 1 && !op->up && !op->in_nchw && !op->out_nchw &&
           op->hin == op->hout && op->win == op->wout && op->relu && mval_conv_bneck_p2_supported(op->cin, 64, n_images, op->hin, op->win);
  if (op && op->kind == MVAL_OP_BLOCK && algo == MVAL_ALGO_MFMA_P2)
    return n_images > 0 && op->cin == op->cout && op->k == 3 && op->stride == 1 && op->pad == 1 && !op->up && !op->in_nchw &&
           !op->out_nchw && op->hin == op->hout && op->win == op->wout && mval_conv_block_p2_supported(op->cin, n_images, op->hin, op->win);
  if (op && op->kind == MVAL_OP_BLOCK)
    return algo == MVAL_ALGO_MFMA_H2 && n_images > 0 && op->cin == op->cout && op->k == 3 && op->stride == 1 && op->pad == 1 &&
           !op->up && !op->in_nchw && !op->out_nchw && op->hin == op->hout && op->win == op->wout &&
           mval_conv_block_supported(op->cin, n_images, op->hin, op->win);
  if (algo == MVAL_ALGO_MFMA) return mval_op_mfma_supported(op, n_images);
  if (algo == MVAL_ALGO_MFMA_P2) {
    // (round 5) ConvTranspose2d(k4, s2, p1) = four 2 x 2 parity convs over the input grid, each scattered to its parity of the output planes
    if (op && op->kind == MVAL_OP_DECONV)
      return n_images > 0 && !op->in_nchw && !op->out_nchw && !op->up && op->k == 4 && op->stride == 2 && op->pad == 1 && (op->cin & 7) == 0 &&
             (op->cout & 7) == 0 && op->hout == 2 * op->hin && op->wout == 2 * op->win && op->res1_off < 0 && op->res2_off < 0 &&
             mval_conv_p2_parity_supported(op->cin, op->cout, op->hin, op->win, n_images, 0);
    return op && op->kind == MVAL_OP_CONV && n_images > 0 && !op->in_nchw && op->pad == op->k / 2 &&
           mval_conv_p2_supported(op->k, op->stride, op->cin, op->cout, op->hin, op->win, op->up, op->out_nchw, n_images);
  }
  if ((algo != MVAL_ALGO_MFMA_BF3 && algo != MVAL_ALGO_MFMA_H2) || !op || n_images <= 0) return 0;
  if (op->kind != MVAL_OP_CONV && op->kind != MVAL_OP_DECONV) return 0;
  ConvArgs a = {};
  fill_geometry(a, op, n_images);
  a.planes = algo == MVAL_ALGO_MFMA_H2 ? 2 : 3;
  if (op->kind == MVAL_OP_DECONV) {
    if (op->k != 4 || op->stride != 2 || op->pad != 1 || op->up || op->out_nchw) return 0;
    deconv_parity(a, op, 0, nullptr);
  }
  return mval_conv_split_supported(a);
}

// 1 when the kernel this op runs on keeps arg-max keys next to an NCHW heat-map output (the P2 and MFMA families)
static int op_keeps_argmax_keys(const mval_op* op) {
  if (!op || op->kind != MVAL_OP_CONV || !op->out_nchw || op->up || op->out_off >= 0 || op->k != 1 || op->stride != 1) return 0;
  if (op->algo == MVAL_ALGO_MFMA_P2) return 1;
  return op->algo == MVAL_ALGO_MFMA && !force_direct();  // (conv_mfma.hip's 1x1 kernels: the heat-map layer of the h2 / bf3 / fp32 plans)
}

static int op_launch(const mval_op* op, int n_images, float* workspace, const float* params, const float* net_input,
                     float* net_output, unsigned long long* argmax_keys, void* stream);

extern "C" int mval_op_launch(const mval_op* op, int n_images, float* workspace, const float* params,
                              const float* net_input, float* net_output, void* stream) {
  return op_launch(op, n_images, workspace, params, net_input, net_output, nullptr, stream);
}

static int op_launch(const mval_op* op, int n_images, float* workspace, const float* params, const float* net_input,
                     float* net_output, unsigned long long* argmax_keys, void* stream) {
  MVAL_REQUIRE(op && n_images > 0, "mval_op_launch: bad arguments");
  ConvArgs a = {};
  a.argmax_keys = argmax_keys;
  a.in = op->in_off >= 0 ? workspace + op->in_off : net_input;
  a.out = op->out_off >= 0 ? workspace + op->out_off : net_output;
  a.res1 = op->res1_off >= 0 ? workspace + op->res1_off : nullptr;
  a.res2 = op->res2_off >= 0 ? workspace + op->res2_off : nullptr;
  a.w = op->w_off >= 0 ? params + op->w_off : nullptr;
  a.scale = op->scale_off >= 0 ? params + op->scale_off : nullptr;
  a.shift = op->shift_off >= 0 ? params + op->shift_off : nullptr;
  fill_geometry(a, op, n_images);
  MVAL_REQUIRE(a.in && a.out, "mval_op_launch: missing input/output buffer");
  hipStream_t s = mval_stream(stream);
  if (op->kind == MVAL_OP_TO_P2) {
    MVAL_REQUIRE(op->in_off >= 0 && op->out_off >= 0 && op->in_amax_off > 0 && op->out_amax_off > 0 && (op->cin & 7) == 0,
                 "mval_op_launch: malformed MVAL_OP_TO_P2");
    mval_launch_nhwc_to_p2(a.in, reinterpret_cast<const unsigned*>(workspace + op->in_amax_off), reinterpret_cast<_Float16*>(a.out),
                           reinterpret_cast<unsigned*>(workspace + op->out_amax_off), n_images, op->hin * op->win, op->cin, s);
    MVAL_CHECK_LAUNCH("mval_op_launch/to_p2");
    return 0;
  }
  if (op->kind == MVAL_OP_FUSE_UP) {
    MVAL_REQUIRE(op->algo == MVAL_ALGO_MFMA_P2 && op->out_amax_off > 0 && op->res1_amax_off > 0 && a.res1 && op->out_off >= 0 && op->n_terms >= 2 &&
                     op->n_terms <= 3,
                 "mval_op_launch: malformed MVAL_OP_FUSE_UP");
    const void* tin[3] = {nullptr, nullptr, nullptr};
    const unsigned* trow[3] = {nullptr, nullptr, nullptr};
    int64_t wu[3] = {0, 0, 0};
    for (int j = 0; j < op->n_terms; j++) {
      MVAL_REQUIRE(op->t_in_off[j] >= 0 && op->t_in_amax_off[j] > 0 && op->t_w_off[j] >= 0 && op->t_scale_off[j] >= 0 && op->t_shift_off[j] >= 0 &&
                       op->t_bound_off[j] >= 0,
                   "mval_op_launch: malformed MVAL_OP_FUSE_UP term %d", j);
      tin[j] = workspace + op->t_in_off[j];
      trow[j] = reinterpret_cast<const unsigned*>(workspace + op->t_in_amax_off[j]);
      wu[j] = op->t_w_off[j] + (int64_t)mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op->cout, op->t_cin[j], 1) - 4;
    }
    int rc = mval_launch_conv_fuse_up_p2(op->cout, op->n_terms, op->relu, a.res1, reinterpret_cast<const unsigned*>(workspace + op->res1_amax_off), a.out,
                                         reinterpret_cast<unsigned*>(workspace + op->out_amax_off), params, tin, trow, op->t_cin, op->t_up, op->t_w_off, wu,
                                         op->t_scale_off, op->t_shift_off, op->t_bound_off, n_images, op->hout, op->wout, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no fused up-path kernel for c%d %dx%d", op->cout, op->hout, op->wout);
    MVAL_CHECK_LAUNCH("mval_op_launch/fuse_up_p2");
    return 0;
  }
  if (op->kind == MVAL_OP_STEM_P2) {
    MVAL_REQUIRE(op->algo == MVAL_ALGO_MFMA_P2 && op->in_amax_off > 0 && op->out_amax_off > 0 && a.w && a.scale && a.shift && op->bound_off >= 0 &&
                     op->w2_off >= 0 && op->scale2_off >= 0 && op->shift2_off >= 0 && op->bound2_off >= 0 && op->out_off >= 0 && net_input &&
                     op->in_off < 0,
                 "mval_op_launch: malformed MVAL_OP_STEM_P2");
    const size_t nw2 = mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, 64, 64, 3);
    const float* w2 = params + op->w2_off;
    int rc = mval_launch_conv_stem_p2(net_input, a.out, a.w, a.scale, a.shift, params + op->bound_off, w2, w2 + nw2 - 4, params + op->scale2_off,
                                      params + op->shift2_off, params + op->bound2_off, reinterpret_cast<unsigned*>(workspace + op->in_amax_off),
                                      reinterpret_cast<unsigned*>(workspace + op->out_amax_off), n_images, op->hin, op->win, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no fused P2 stem kernel for %dx%d", op->hin, op->win);
    MVAL_CHECK_LAUNCH("mval_op_launch/stem_p2");
    return 0;
  }
  if (op->kind == MVAL_OP_BNECK) {
    MVAL_REQUIRE(op->algo == MVAL_ALGO_MFMA_P2 && op->in_amax_off > 0 && op->out_amax_off > 0 && op->res1_amax_off > 0 && a.res1 && a.w &&
                     a.scale && a.shift && op->bound_off >= 0 && op->w2_off >= 0 && op->scale2_off >= 0 && op->shift2_off >= 0 &&
                     op->bound2_off >= 0 && op->w3_off >= 0 && op->scale3_off >= 0 && op->shift3_off >= 0 && op->bound3_off >= 0 &&
                     op->in_off >= 0 && op->out_off >= 0 && op->res2_off < 0 && op->relu,
                 "mval_op_launch: malformed MVAL_OP_BNECK");
    const size_t nw[3] = {mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, 64, op->cin, 1), mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, 64, 64, 3),
                          mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, 256, 64, 1)};
    const int64_t w[3] = {op->w_off, op->w2_off, op->w3_off};
    const int64_t wu[3] = {w[0] + (int64_t)nw[0] - 4, w[1] + (int64_t)nw[1] - 4, w[2] + (int64_t)nw[2] - 4};
    const int64_t sc[3] = {op->scale_off, op->scale2_off, op->scale3_off};
    const int64_t sh[3] = {op->shift_off, op->shift2_off, op->shift3_off};
    const int64_t bd[3] = {op->bound_off, op->bound2_off, op->bound3_off};
    int rc = mval_launch_conv_bneck_p2(op->cin, a.in, a.res1, a.out, params, w, wu, sc, sh, bd, reinterpret_cast<const unsigned*>(workspace + op->in_amax_off),
                                       reinterpret_cast<const unsigned*>(workspace + op->res1_amax_off),
                                       reinterpret_cast<unsigned*>(workspace + op->out_amax_off), n_images, op->hin, op->win, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no fused P2 Bottleneck kernel for c%d %dx%d", op->cin, op->hin, op->win);
    MVAL_CHECK_LAUNCH("mval_op_launch/bneck_p2");
    return 0;
  }
  if (op->kind == MVAL_OP_BLOCK && op->algo == MVAL_ALGO_MFMA_P2) {
    MVAL_REQUIRE(op->in_amax_off > 0 && op->out_amax_off > 0 && a.w && a.scale && a.shift && op->w2_off >= 0 && op->scale2_off >= 0 &&
                     op->shift2_off >= 0 && op->bound_off >= 0 && op->bound2_off >= 0 && op->in_off >= 0 && op->out_off >= 0,
                 "mval_op_launch: malformed P2 MVAL_OP_BLOCK");
    const size_t nw = mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op->cout, op->cin, 3);
    const float* w2 = params + op->w2_off;
    int rc = mval_launch_conv_block_p2(op->cin, a.in, a.out, a.w, a.w + nw - 4, a.scale, a.shift, params + op->bound_off, w2, w2 + nw - 4,
                                       params + op->scale2_off, params + op->shift2_off, params + op->bound2_off,
                                       reinterpret_cast<const unsigned*>(workspace + op->in_amax_off),
                                       reinterpret_cast<unsigned*>(workspace + op->out_amax_off), n_images, op->hin, op->win, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no fused P2 BasicBlock kernel for c%d %dx%d", op->cin, op->hin, op->win);
    MVAL_CHECK_LAUNCH("mval_op_launch/block_p2");
    return 0;
  }
  if (op->kind == MVAL_OP_BLOCK) {
    MVAL_REQUIRE(op->algo == MVAL_ALGO_MFMA_H2 && op->in_amax_off > 0 && a.w && a.scale && a.shift && op->w2_off >= 0 &&
                     op->scale2_off >= 0 && op->shift2_off >= 0 && (op->res1_off < 0 || op->res1_off == op->in_off) &&
                     op->res2_off < 0 && op->relu,
                 "mval_op_launch: malformed MVAL_OP_BLOCK");
    const size_t nw = mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op->cout, op->cin, 3);
    const float* w2 = params + op->w2_off;
    int rc = mval_launch_conv_block(op->cin, a.in, a.out, a.w, a.scale, a.shift, a.w + nw - 4, w2, params + op->scale2_off,
                                    params + op->shift2_off, w2 + nw - 4,
                                    reinterpret_cast<const unsigned*>(workspace + op->in_amax_off),
                                    op->out_amax_off > 0 ? reinterpret_cast<unsigned*>(workspace + op->out_amax_off) : nullptr,
                                    n_images, op->hin, op->win, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no fused BasicBlock kernel for c%d %dx%d", op->cin, op->hin, op->win);
    MVAL_CHECK_LAUNCH("mval_op_launch/block");
    return 0;
  }
  if (op->algo == MVAL_ALGO_MFMA_P2) {
    auto p2_args = [&](const mval_op* o, P2Args& p) -> int {
      const float* w = o->w_off >= 0 ? params + o->w_off : nullptr;
      MVAL_REQUIRE((o->kind == MVAL_OP_CONV || o->kind == MVAL_OP_DECONV) && w && o->scale_off >= 0 && o->shift_off >= 0 && o->in_amax_off > 0 && o->in_off >= 0 &&
                       (o->out_nchw || (o->out_amax_off > 0 && o->bound_off >= 0)) &&
                       (o->res1_off < 0 || o->res1_amax_off > 0) && (o->res2_off < 0 || o->res2_amax_off > 0),
                   "mval_op_launch: malformed MVAL_ALGO_MFMA_P2 op");
      p = {};
      p.in = reinterpret_cast<const _Float16*>(workspace + o->in_off);
      p.w = w;
      p.w_unscale = w + mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, o->cout, o->cin, o->k) - 4;  // (a transposed conv: k = 4, all 16 taps)
      p.scale = params + o->scale_off; p.shift = params + o->shift_off;
      p.bound = o->bound_off >= 0 ? params + o->bound_off : nullptr;
      p.res1 = o->res1_off >= 0 ? reinterpret_cast<const _Float16*>(workspace + o->res1_off) : nullptr;
      p.res2 = o->res2_off >= 0 ? reinterpret_cast<const _Float16*>(workspace + o->res2_off) : nullptr;
      p.in_row = reinterpret_cast<const unsigned*>(workspace + o->in_amax_off);
      p.res1_row = p.res1 ? reinterpret_cast<const unsigned*>(workspace + o->res1_amax_off) : nullptr;
      p.res2_row = p.res2 ? reinterpret_cast<const unsigned*>(workspace + o->res2_amax_off) : nullptr;
      if (o->out_nchw) {
        p.out_f32 = o->out_off >= 0 ? workspace + o->out_off : net_output;
        p.argmax_keys = argmax_keys;
      } else {
        p.out = reinterpret_cast<_Float16*>(workspace + o->out_off);
        p.out_row = reinterpret_cast<unsigned*>(workspace + o->out_amax_off);
      }
      p.N = n_images; p.Hin = o->hin; p.Win = o->win; p.Cin = o->cin;
      p.Hout = o->hout; p.Wout = o->wout; p.Cout = o->cout;
      p.k = o->k; p.stride = o->stride; p.up = o->up; p.relu = o->relu;
      return 0;
    };
    P2Args p2;
    int rc = p2_args(op, p2);
    if (rc) return rc;
    if (op->kind == MVAL_OP_DECONV) {
      // pose_resnet.py:107-137: output pixel (2a + py, 2b + px) sees 2 x 2 taps of the 4 x 4 kernel (pack mode 3: parity pp's taps are the
      // pp-th quarter of the packed buffer), input rows a - 1 + py .. a + py: four stride-1 convs over the input grid, each written to its
      // parity of the output planes; the launches share the output's rows (one scale from the one bound, a quarter of the slots each)
      const size_t quarter = (mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op->cout, op->cin, 4) & ~(size_t)7) / 4;
      for (int pp = 0; pp < 4; pp++) {
        P2Args q = p2;
        q.k = 2; q.stride = 1;
        q.Hout = op->hin; q.Wout = op->win;
        q.pad_y = 1 - (pp >> 1); q.pad_x = 1 - (pp & 1);
        q.os = 1; q.oy = pp >> 1; q.ox = pp & 1;
        q.keep_rows = pp != 0;
        q.w = p2.w + pp * quarter;
        rc = mval_launch_conv_p2(q, s);
        MVAL_REQUIRE(rc == 0, "mval_op_launch: no P2 kernel for a parity of the transposed conv cin%d cout%d %dx%d", op->cin, op->cout, op->hin, op->win);
      }
      MVAL_CHECK_LAUNCH("mval_op_launch/p2 deconv");
      return 0;
    }
    rc = mval_launch_conv_p2(p2, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no P2 kernel for conv k%d s%d cin%d cout%d %dx%d", op->k, op->stride, op->cin, op->cout, op->hin, op->win);
    MVAL_CHECK_LAUNCH("mval_op_launch/p2");
    return 0;
  }
  const bool split = op->algo == MVAL_ALGO_MFMA_BF3 || op->algo == MVAL_ALGO_MFMA_H2;
  a.out_amax = op->out_amax_off > 0 ? reinterpret_cast<unsigned*>(workspace + op->out_amax_off) : nullptr;
  bool amax_kept = false;  // does the kernel that runs keep out_amax itself?
  if (split) {
    a.planes = op->algo == MVAL_ALGO_MFMA_H2 ? 2 : 3;
    if (a.planes == 2) {
      MVAL_REQUIRE(op->in_amax_off > 0 && a.w, "mval_op_launch: the fp16-split conv needs in_amax_off (max |x| of its input)");
      a.in_amax = reinterpret_cast<const unsigned*>(workspace + op->in_amax_off);
      a.in_amax_stride = MVAL_AMAX_ROW;
      const int kk = op->kind == MVAL_OP_DECONV ? 4 : op->k;
      a.w_unscale = a.w + mval_packed_weight_floats(MVAL_PACK_MFMA16_H2, op->cout, op->cin, kk) - 4;
    }
    amax_kept = !op->out_nchw && (op->cout & 3) == 0;
  }
  if (op->kind == MVAL_OP_DECONV && split) {
    // the four parity convs in one launch (blockIdx.z): on a few images one parity alone leaves most CUs idle
    const float* w0 = a.w;
    deconv_parity(a, op, 0, w0);
    a.par_w_stride = (int)deconv_parity_floats(op);
    int rc = mval_launch_conv_split(a, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no split MFMA kernel for the transposed conv cin%d cout%d", op->cin, op->cout);
  } else if (op->kind == MVAL_OP_CONV && split) {
    int rc = mval_launch_conv_split(a, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no split MFMA kernel for conv k%d s%d cin%d cout%d", op->k, op->stride, op->cin,
                 op->cout);
  } else if ((op->kind == MVAL_OP_CONV || op->kind == MVAL_OP_DECONV) && op->algo == MVAL_ALGO_MFMA) {
    MVAL_REQUIRE(!force_direct(), "MVAL_FORCE_DIRECT=1 but the plan was packed for the MFMA kernels");
    if (op->kind == MVAL_OP_DECONV) deconv_as_conv(a, op);
    int rc = mval_launch_conv_mfma(a, s);
    MVAL_REQUIRE(rc == 0, "mval_op_launch: no MFMA kernel for conv k%d s%d cin%d cout%d", op->k, op->stride, op->cin,
                 op->cout);
    amax_kept = !op->out_nchw && (op->cout & 3) == 0;
  } else {
    MVAL_REQUIRE(op->kind == MVAL_OP_MAXPOOL || (a.w && a.scale && a.shift), "mval_op_launch: missing parameters");
    // 3-channel NCHW stems have their own store-shaped kernel; everything else is generic
    if (op->kind != MVAL_OP_CONV || force_direct() || mval_launch_conv_stem(a, s))
      mval_launch_conv_direct(a, op->kind, s);
    else
      amax_kept = true;  // the stem kernel keeps it
  }
  if (a.out_amax && !amax_kept)  // max-pool / generic direct kernels: one extra read of the output
    mval_launch_amax(a.out, (int64_t)(op->hout << op->up) * (op->wout << op->up) * op->cout, n_images, a.out_amax, s);
  MVAL_CHECK_LAUNCH("mval_op_launch");
  return 0;
}

#define MVAL_MAX_DEVICES 16
static MvalLanes g_lanes[MVAL_MAX_DEVICES];

struct MvalNet {
  std::vector<mval_op> ops;
  int n_lanes = 1;
  int lanes_override = -1;  // -1: MVAL_STREAMS decides; 0 / 1: forced off / on (mval_net_set_multi_stream)
};

static int multi_stream_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MVAL_STREAMS");
    v = (e && e[0] == '1' && e[1] == 0) ? 0 : 1;  // MVAL_STREAMS=1 forces single-stream execution
  }
  return v;
}

extern "C" void* mval_net_create(const mval_op* ops, int n_ops) {
  if (!ops || n_ops <= 0) {
    mval_set_error("mval_net_create: empty op list");
    return nullptr;
  }
  MvalNet* n = new MvalNet();
  n->ops.assign(ops, ops + n_ops);
  for (const auto& o : n->ops)
    if (o.lane + 1 > n->n_lanes) n->n_lanes = o.lane + 1;
  if (n->n_lanes > MVAL_MAX_LANES) n->n_lanes = MVAL_MAX_LANES;
  return n;
}

extern "C" int mval_net_set_multi_stream(void* net, int mode) {
  MVAL_REQUIRE(net && mode >= -1 && mode <= 1, "mval_net_set_multi_stream: bad arguments");
  reinterpret_cast<MvalNet*>(net)->lanes_override = mode;
  return 0;
}

extern "C" void mval_net_destroy(void* net) {
  delete reinterpret_cast<MvalNet*>(net);
}

// The side streams and the fork / join events are per DEVICE and shared by every plan: creation is guarded, and ONE forward
// (or training pass) may be in flight per device at a time -- two host threads driving plans on the same GPU would record
// and wait on the same events.  (One process per GPU, one stream of work per process: the package's execution model.)
static std::mutex g_lanes_mutex;

MvalLanes* mval_device_lanes() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVAL_MAX_DEVICES) return nullptr;
  MvalLanes* L = &g_lanes[dev];
  std::lock_guard<std::mutex> lock(g_lanes_mutex);
  if (L->ready) return L;
  if (hipEventCreateWithFlags(&L->fork_ev, hipEventDisableTiming) != hipSuccess) return nullptr;
  for (int l = 1; l < MVAL_MAX_LANES; l++) {
    if (hipStreamCreateWithFlags(&L->side[l], hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&L->join_ev[l], hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  L->ready = true;
  return L;
}

extern "C" int mval_net_keeps_argmax_keys(void* net) {
  if (!net) return 0;
  MvalNet* n = reinterpret_cast<MvalNet*>(net);
  int writers = 0, keepers = 0;
  for (const auto& o : n->ops)
    if (o.out_off < 0) {  // the op(s) that write the network output
      writers++;
      keepers += op_keeps_argmax_keys(&o);
    }
  return writers == 1 && keepers == 1;
}

static int net_forward(void* net, int n_images, float* workspace, const float* params, const float* input_nchw,
                       float* output_nchw, unsigned long long* argmax_keys, void* stream);

extern "C" int mval_net_forward(void* net, int n_images, float* workspace, const float* params,
                                const float* input_nchw, float* output_nchw, void* stream) {
  return net_forward(net, n_images, workspace, params, input_nchw, output_nchw, nullptr, stream);
}

extern "C" int mval_net_forward_keys(void* net, int n_images, float* workspace, const float* params, const float* input_nchw,
                                     float* output_nchw, uint64_t* argmax_keys, void* stream) {
  MVAL_REQUIRE(net && argmax_keys, "mval_net_forward_keys: null argument");
  MVAL_REQUIRE(mval_net_keeps_argmax_keys(net), "mval_net_forward_keys: the plan's heat-map layer does not run on a kernel that keeps arg-max keys");
  MvalNet* n = reinterpret_cast<MvalNet*>(net);
  int64_t maps = 0;
  for (const auto& o : n->ops)
    if (o.out_off < 0) maps = (int64_t)n_images * o.cout;
  // every stored value has a key > 0: zero = "nothing stored yet" (a memset node when captured)
  if (hipMemsetAsync(argmax_keys, 0, (size_t)maps * MVAL_ARGMAX_SLOTS * 8, mval_stream(stream)) != hipSuccess) {
    mval_set_error("mval_net_forward_keys: hipMemsetAsync failed");
    return -2;
  }
  return net_forward(net, n_images, workspace, params, input_nchw, output_nchw, reinterpret_cast<unsigned long long*>(argmax_keys), stream);
}

static int net_forward(void* net, int n_images, float* workspace, const float* params, const float* input_nchw,
                       float* output_nchw, unsigned long long* argmax_keys, void* stream) {
  MVAL_REQUIRE(net, "mval_net_forward: null net");
  MvalNet* n = reinterpret_cast<MvalNet*>(net);
  hipStream_t main_s = mval_stream(stream);
  const bool multi = n->n_lanes > 1 && (n->lanes_override < 0 ? multi_stream_enabled() : n->lanes_override != 0);
  MvalLanes* L = multi ? mval_device_lanes() : nullptr;
  if (multi) MVAL_REQUIRE(L != nullptr, "mval_net_forward: could not create the side streams");
  MvalLaneWalk walk(L, main_s);
  for (size_t i = 0; i < n->ops.size(); i++) {
    const mval_op& op = n->ops[i];
    hipStream_t s = walk.stream_for(op.phase, op.lane < n->n_lanes ? op.lane : 0);
    int rc = op_launch(&n->ops[i], n_images, workspace, params, input_nchw, output_nchw, op.out_off < 0 ? argmax_keys : nullptr, s);
    if (rc) return rc;
  }
  walk.finish();
  return 0;
}

extern "C" double mval_op_flops(const mval_op* op, int n_images) {
  if (!op || op->kind == MVAL_OP_MAXPOOL || op->kind == MVAL_OP_TO_P2) return 0.0;
  if (op->kind == MVAL_OP_FUSE_UP) {  // the terms' 1x1 convs at their own resolutions
    double f = 0.0;
    for (int j = 0; j < op->n_terms; j++) f += 2.0 * n_images * (double)(op->hout >> op->t_up[j]) * (op->wout >> op->t_up[j]) * op->t_cin[j] * op->cout;
    return f;
  }
  if (op->kind == MVAL_OP_STEM_P2)  // both convs (conv1's halo recompute is not counted)
    return 2.0 * n_images * ((double)(op->hin / 2) * (op->win / 2) * 3 * 64 * 9 + (double)op->hout * op->wout * 64 * 64 * 9);
  if (op->kind == MVAL_OP_BNECK)  // the three convs (the halo recompute of conv1 is not counted)
    return 2.0 * n_images * op->hout * op->wout * ((double)op->cin * 64 + 64.0 * 64 * 9 + 64.0 * 256);
  if (op->kind == MVAL_OP_BLOCK)  // algorithmic work of the two convs (the halo recompute is not counted)
    return 2.0 * 2.0 * n_images * op->hout * op->wout * (double)op->cin * op->cout * 9;
  if (op->kind == MVAL_OP_DECONV)  // every input pixel meets every tap once
    return 2.0 * n_images * op->hin * op->win * (double)op->cin * op->cout * op->k * op->k;
  return 2.0 * n_images * op->hout * op->wout * (double)op->cin * op->cout * op->k * op->k;
}

extern "C" int mval_net_forward_timed(void* net, int n_images, float* workspace, const float* params,
                                      const float* input_nchw, float* output_nchw, void* stream, float* ms_per_op) {
  MVAL_REQUIRE(net && ms_per_op, "mval_net_forward_timed: null argument");
  MvalNet* n = reinterpret_cast<MvalNet*>(net);
  hipStream_t s = mval_stream(stream);
  std::vector<hipEvent_t> ev(n->ops.size() + 1);
  for (auto& e : ev) {
    if (hipEventCreate(&e) != hipSuccess) {
      mval_set_error("mval_net_forward_timed: hipEventCreate failed");
      return -3;
    }
  }
  int rc = 0;
  (void)hipEventRecord(ev[0], s);
  for (size_t i = 0; i < n->ops.size() && !rc; i++) {
    rc = mval_op_launch(&n->ops[i], n_images, workspace, params, input_nchw, output_nchw, stream);
    (void)hipEventRecord(ev[i + 1], s);
  }
  (void)hipEventSynchronize(ev.back());
  for (size_t i = 0; i < n->ops.size(); i++) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
    ms_per_op[i] = ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}
