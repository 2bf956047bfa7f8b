// Shared definitions of the fused conv operators (direct and MFMA kernels).
#pragma once
#include "mval_common.h"

struct ConvArgs {
  const float* in;
  const float* w;      // packed weights
  const float* scale;  // [cout] folded BN scale (or 1)
  const float* shift;  // [cout] folded BN shift (or bias)
  const float* res1;   // optional residuals at output resolution (after upsample), NHWC
  const float* res2;
  float* out;
  int N, Hin, Win, Cin, Hout, Wout, Cout;  // Hout/Wout before the fused upsample
  int k, stride, pad;
  int up, relu, in_nchw, out_nchw;
  int dil;  // 1, or 2: the input is read as if zero-dilated by 2 (data gradient of a stride-2 conv)
  // MFMA tiling (filled by the launcher)
  int th, tw, tn, tw_log2, thw_log2;
  int tiles_x, tiles_y;
  int G_total, NS_total;
};

// out = act(((v + res1) + res2)), written to the 2^up x 2^up replicated positions.
__device__ __forceinline__ void conv_store(const ConvArgs& a, int n, int y, int x, int c, float v) {
  const int Ho = a.Hout << a.up, Wo = a.Wout << a.up;
  const int rep = 1 << a.up;
  for (int dy = 0; dy < rep; dy++) {
    for (int dx = 0; dx < rep; dx++) {
      const int Y = (y << a.up) + dy, X = (x << a.up) + dx;
      const int64_t o = (((int64_t)n * Ho + Y) * Wo + X) * a.Cout + c;
      float r = v;
      if (a.res1) r += a.res1[o];
      if (a.res2) r += a.res2[o];
      if (a.relu) r = fmaxf(r, 0.f);
      if (a.out_nchw)
        a.out[(((int64_t)n * a.Cout + c) * Ho + Y) * Wo + X] = r;
      else
        a.out[o] = r;
    }
  }
}

int mval_launch_conv_mfma(const ConvArgs& a, hipStream_t s);  // conv_mfma.hip; returns 1 if unsupported
int mval_conv_mfma_supported(const ConvArgs& a);            // same selection logic, no launch
int mval_launch_conv_bf3(const ConvArgs& a, hipStream_t s);   // conv_mfma_bf3.hip; returns 1 if unsupported
int mval_conv_bf3_supported(const ConvArgs& a);
int mval_pack_bf3(int mode, const float* w, float* packed, int cout, int cin, int k, hipStream_t s);
int mval_launch_conv_direct(const ConvArgs& a, int kind, hipStream_t s);
int mval_launch_conv_stem(const ConvArgs& a, hipStream_t s);  // conv_stem.hip; returns 1 if unsupported
