// Stem conv: 3-channel NCHW image -> NHWC feature map, k x k stride 2 (+BN+ReLU).
// HRNet conv1 (hrnet.py:303-305, 469-471: 3x3 s2) and PoseResNet conv1 (pose_resnet.py:32,
// 140-142: 7x7 s2).  K = 27 / 147 is too short and too ragged for the matrix cores and the
// layer is HBM-bound anyway (writes N*H/2*W/2*64*4 bytes: 537 MB at the BASELINE batch), so
// this is a VALU kernel shaped around the store:
//   * a workgroup computes an 8 x 16 output-pixel tile x all couts; the input patch (3 planes)
//     and the weights ([tap][cin][cout]) sit in LDS;
//   * lane = (pixel group, cout quad): 16 cout-quads x 16 groups of 8 consecutive pixels, so a
//     wave's float4 stores cover 4 pixels x 64 couts = 1 KiB contiguous NHWC bytes;
//   * per (tap, cin): one float4 weight read (16 distinct addresses per wave) + 8 broadcast
//     input reads feed 32 FMAs per lane.
#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ST_TH 8
#define ST_TW 16
#define ST_P 8  // consecutive output pixels per lane

// (waves_per_eu: left alone the compiler unrolls all 27 taps, hoists every LDS read and ends at 256
// VGPRs = one wave per SIMD -- measured 500 us / 1.3 TB/s at the BASELINE batch)
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void conv_stem_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int S = 2;
  constexpr int PH = (ST_TH - 1) * S + KS, PW = (ST_TW - 1) * S + KS;
  constexpr int PWP = PW | 1;  // odd row stride
  const int cq_n = a.Cout / 4;  // cout quads (<= 16 per pass)
  float* wl = smem;                               // [KS*KS*3][Cout]
  float* patch = smem + KS * KS * 3 * a.Cout;     // [3][PH][PWP]
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tiles_x = (a.Wout + ST_TW - 1) / ST_TW, tiles_y = (a.Hout + ST_TH - 1) / ST_TH;
  const int txi = t % tiles_x;
  t /= tiles_x;
  const int tyi = t % tiles_y;
  const int n = t / tiles_y;
  const int oy0 = tyi * ST_TH, ox0 = txi * ST_TW;
  const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;

  for (int i = tid; i < KS * KS * 3 * a.Cout; i += 256) wl[i] = a.w[i];
  for (int i = tid; i < 3 * PH * PW; i += 256) {
    const int c = i / (PH * PW), r = i % (PH * PW);
    const int py = r / PW, px = r % PW;
    const int iy = iy0 + py, ix = ix0 + px;
    float v = 0.f;
    if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) v = a.in[(((int64_t)n * 3 + c) * a.Hin + iy) * a.Win + ix];
    patch[(c * PH + py) * PWP + px] = v;
  }
  __syncthreads();

  const int cq = tid & 15;        // cout quad within a pass
  const int grp = tid >> 4;       // 16 groups: row = grp >> 1, x half = grp & 1
  const int ty = grp >> 1, tx0 = (grp & 1) * ST_P;
  float amax = 0.f;  // max |stored value| (the fp16-split consumer's activation scale, conv_common.h)
  for (int cbase = 0; cbase < cq_n; cbase += 16) {
    const int q = cbase + cq;
    if (q >= cq_n) continue;
    f32x4 acc[ST_P];
#pragma unroll
    for (int p = 0; p < ST_P; p++) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < KS; ky++) {
#pragma unroll
      for (int kx = 0; kx < KS; kx++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + ((ky * KS + kx) * 3 + c) * a.Cout + q * 4);
          const float* row = patch + (c * PH + ty * S + ky) * PWP + tx0 * S + kx;
#pragma unroll
          for (int p = 0; p < ST_P; p++) {
            // two v_pk_fma_f32 per pixel (the layer is VALU-bound before it is HBM-bound)
            const f32x2 vv = {row[p * S], row[p * S]};
            const f32x2 lo = __builtin_elementwise_fma(vv, (f32x2){w4.x, w4.y}, (f32x2){acc[p].x, acc[p].y});
            const f32x2 hi = __builtin_elementwise_fma(vv, (f32x2){w4.z, w4.w}, (f32x2){acc[p].z, acc[p].w});
            acc[p] = (f32x4){lo.x, lo.y, hi.x, hi.y};
          }
        }
      }
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + q * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + q * 4);
    const int y = oy0 + ty;
    if (y < a.Hout) {
#pragma unroll
      for (int p = 0; p < ST_P; p++) {
        const int x = ox0 + tx0 + p;
        if (x >= a.Wout) continue;
        f32x4 r = acc[p] * sc + sh;
        if (a.relu) {
          r.x = mval_relu(r.x); r.y = mval_relu(r.y); r.z = mval_relu(r.z); r.w = mval_relu(r.w);
        }
        *reinterpret_cast<f32x4*>(a.out + (((int64_t)n * a.Hout + y) * a.Wout + x) * a.Cout + q * 4) = r;
        amax = conv_amax4(amax, r.x, r.y, r.z, r.w);
      }
    }
  }
  if (a.out_amax) conv_amax_commit(a.out_amax + (int64_t)n * MVAL_AMAX_ROW, (int)(blockIdx.x % (unsigned)a.amax_tiles), a.amax_tiles, amax);
}

// returns 1 when the op is not a stem of the supported shape
int mval_launch_conv_stem(const ConvArgs& a0, hipStream_t s) {
  ConvArgs a = a0;
  if (!a.in_nchw || a.Cin != 3 || a.stride != 2 || a.up || a.res1 || a.res2 || a.out_nchw || (a.Cout & 3)) return 1;
  if (a.k != 3 && a.k != 7) return 1;
  if (a.pad != a.k / 2) return 1;
  const int PH = (ST_TH - 1) * 2 + a.k, PW = ((ST_TW - 1) * 2 + a.k) | 1;
  size_t smem = (size_t)(a.k * a.k * 3 * a.Cout + 3 * PH * PW) * sizeof(float);
  if (smem > 64 * 1024) return 1;
  const int tiles = ((a.Wout + ST_TW - 1) / ST_TW) * ((a.Hout + ST_TH - 1) / ST_TH) * a.N;
  conv_amax_prepare(a, tiles / a.N, 1, s);
  if (a.k == 3)
    hipLaunchKernelGGL(conv_stem_kernel<3>, dim3(tiles), dim3(256), smem, s, a);
  else
    hipLaunchKernelGGL(conv_stem_kernel<7>, dim3(tiles), dim3(256), smem, s, a);
  return 0;
}
