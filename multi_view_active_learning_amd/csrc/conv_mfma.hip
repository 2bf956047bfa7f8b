// Fused conv (3x3 / 1x1, stride 1 / 2) + BatchNorm + residual(s) + ReLU (+ nearest upsample,
// + NCHW store) on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Replaces the cuDNN/ATen convs + separate BN/ReLU/add/upsample framework ops of the
// reference's HRNet / PoseResNet forward (pose_estimators/hrnet.py:36-52,75-95,199-287).
//
// Implicit GEMM, M = output pixels, N = couts, K = taps x cin:
//   * a workgroup (4 waves) owns a tile of tn x th x tw output pixels x (16*NT*WN) couts of
//     one image group; the INPUT PATCH of that tile (with its halo) for a chunk of KC input
//     channels is staged once in LDS ([pixel][KC+8] floats: the +8 pad makes the per-tap
//     ds_read_b128 fragment reads bank-conflict free) and re-used by all k*k taps, so global
//     ->LDS traffic is ~1.3x the activations instead of 9x (im2col);
//   * A fragments (pixels x 4 consecutive cin) come from LDS with one ds_read_b128 per
//     16-pixel sub-tile per 16 cin; a tap is just a constant LDS offset;
//   * B fragments (weights) are pre-packed on device in fragment order
//     [tap][cin/16][cout/16][lane][4] (mval_pack_conv_weights) so each wave fetches its
//     16 cin x 16 cout block with ONE fully coalesced 1 KiB global_load_dwordx4, straight
//     to VGPRs (weights are shared by every workgroup -> L2 hits), prefetched one step ahead;
//   * waves split N first (each cout block is fetched by one wave only) and M second;
//   * epilogue in registers: y = acc*scale + shift (+res1) (+res2), ReLU, optional 2^up
//     nearest replication, NHWC or NCHW store.
// fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s peak); the kernel is MFMA-bound.
#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define KPAD 8

template <int KS, int S, int KC, int WN, int WM, int NT, int MS>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KCP = KC + KPAD;
  constexpr int GC = KC / 16;  // 16-cin groups per chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int PH = (a.th - 1) * S + KS, PW = (a.tw - 1) * S + KS;
  const int pad = (KS - 1) / 2;

  // tile origin
  int t = blockIdx.x;
  const int txi = t % a.tiles_x;
  t /= a.tiles_x;
  const int tyi = t % a.tiles_y;
  const int n0 = (t / a.tiles_y) * a.tn;
  const int oy0 = tyi * a.th, ox0 = txi * a.tw;
  const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;

  // cout blocks of this wave
  const int ns0 = (blockIdx.y * WN + wn) * NT;
  const bool wave_active = ns0 < a.NS_total;

  // per-lane LDS base of each 16-pixel sub-tile (pixel = lane & 15, cin quad = lane >> 4)
  int abase[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ms++) {
    const int p = (wm * MS + ms) * 16 + (lane & 15);
    const int tni = p >> a.thw_log2;
    const int rem = p & ((1 << a.thw_log2) - 1);
    const int ty = rem >> a.tw_log2, tx = rem & ((1 << a.tw_log2) - 1);
    abase[ms] = ((tni * PH + ty * S) * PW + tx * S) * KCP + (lane >> 4) * 4;
  }

  f32x4 acc[MS][NT];
#pragma unroll
  for (int ms = 0; ms < MS; ms++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[ms][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const f32x4* wq = reinterpret_cast<const f32x4*>(a.w) + lane;
  const int patch_px = a.tn * PH * PW;
  constexpr int C4 = KC / 4;  // float4 per pixel per chunk

  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    if (c0) __syncthreads();
    // ---- stage the input patch chunk: coalesced float4 loads (NHWC), zero outside ----
    for (int e = tid; e < patch_px * C4; e += 256) {
      const int px = e / C4, q = e - px * C4;
      int r = px;
      const int pxx = r % PW;
      r /= PW;
      const int pyy = r % PH;
      const int tni = r / PH;
      const int iy = iy0 + pyy, ix = ix0 + pxx, n = n0 + tni;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n < a.N && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
        v = *reinterpret_cast<const f32x4*>(a.in + (((int64_t)n * a.Hin + iy) * a.Win + ix) * a.Cin + c0 + q * 4);
      *reinterpret_cast<f32x4*>(smem + px * KCP + q * 4) = v;
    }
    __syncthreads();
    if (!wave_active) continue;

    const int g0 = c0 / 16;
    // weight fragment stream for this chunk: index ((tap*G + g)*NS + ns) * 64 (+lane)
    f32x4 bcur[NT], bnxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int ns = min(ns0 + nt, a.NS_total - 1);
      bcur[nt] = wq[((int64_t)(0 * a.G_total + g0) * a.NS_total + ns) * 64];
    }
#pragma unroll
    for (int it = 0; it < KS * KS * GC; it++) {
      const int tap = it / GC, gg = it % GC;
      if (it + 1 < KS * KS * GC) {
        const int tap2 = (it + 1) / GC, gg2 = (it + 1) % GC;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const int ns = min(ns0 + nt, a.NS_total - 1);
          bnxt[nt] = wq[((int64_t)(tap2 * a.G_total + g0 + gg2) * a.NS_total + ns) * 64];
        }
      }
      const int toff = ((tap / KS) * PW + (tap % KS)) * KCP + gg * 16;
      f32x4 af[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ms++) af[ms] = *reinterpret_cast<const f32x4*>(smem + abase[ms] + toff);
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int ms = 0; ms < MS; ms++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
            acc[ms][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ms][jj], bcur[nt][jj], acc[ms][nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) bcur[nt] = bnxt[nt];
    }
  }
  if (!wave_active) return;

  // ---- epilogue: C layout col(cout) = lane & 15, row(pixel) = (lane >> 4) * 4 + reg ----
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int c = (ns0 + nt) * 16 + (lane & 15);
    if (c >= a.Cout) continue;
    const float sc = a.scale[c], sh = a.shift[c];
#pragma unroll
    for (int ms = 0; ms < MS; ms++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int p = (wm * MS + ms) * 16 + (lane >> 4) * 4 + r;
        const int tni = p >> a.thw_log2;
        const int rem = p & ((1 << a.thw_log2) - 1);
        const int y = oy0 + (rem >> a.tw_log2), x = ox0 + (rem & ((1 << a.tw_log2) - 1));
        const int n = n0 + tni;
        if (n < a.N && y < a.Hout && x < a.Wout) conv_store(a, n, y, x, c, acc[ms][nt][r] * sc + sh);
      }
    }
  }
}

template <int KS, int S, int KC, int WN, int WM, int NT, int MS>
static int launch_cfg(ConvArgs a, int th, int tw, int tn, hipStream_t s) {
  a.th = th; a.tw = tw; a.tn = tn;
  a.tw_log2 = __builtin_ctz(tw);
  a.thw_log2 = __builtin_ctz(th * tw);
  a.tiles_x = (a.Wout + tw - 1) / tw;
  a.tiles_y = (a.Hout + th - 1) / th;
  const int ngroups = (a.N + tn - 1) / tn;
  const int PH = (th - 1) * S + KS, PW = (tw - 1) * S + KS;
  size_t smem = (size_t)tn * PH * PW * (KC + KPAD) * sizeof(float);
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * ngroups), (unsigned)((a.NS_total + WN * NT - 1) / (WN * NT)));
  hipLaunchKernelGGL((conv_mfma_kernel<KS, S, KC, WN, WM, NT, MS>), grid, dim3(256), smem, s, a);
  return 0;
}

// pixel-tile shape for a feature map: tiles are 64 or 128 pixels, tw in {8, 16}
static void pick_tile(int H, int W, int mt, int* th, int* tw, int* tn) {
  int w = (W > 8) ? 16 : 8;
  int h = mt / w, n = 1;
  int hh = 1;
  while (hh < H) hh <<= 1;
  if (hh < h) {  // map shorter than the tile: put several images into one tile
    n = h / hh;
    h = hh;
  }
  *th = h; *tw = w; *tn = n;
}

template <int KS, int S>
static int dispatch(const ConvArgs& a, hipStream_t s) {
  const int kc32 = (a.Cin % 32 == 0);
  int th, tw, tn;
  const int64_t px = (int64_t)a.N * a.Hout * a.Wout;
  const bool n48 = (a.NS_total % 3 == 0) && (a.NS_total % 4 != 0);  // HRNet-W48 widths
  // stride-2 patches are 4x larger: keep them at 64-pixel tiles
  const bool small = (S == 2) || px * a.NS_total < (int64_t)128 * 4 * 2048;
  if (n48) {
    pick_tile(a.Hout, a.Wout, S == 2 ? 64 : 128, &th, &tw, &tn);
    if (S == 2) return kc32 ? launch_cfg<KS, S, 32, 1, 4, 3, 1>(a, th, tw, tn, s) : launch_cfg<KS, S, 16, 1, 4, 3, 1>(a, th, tw, tn, s);
    return kc32 ? launch_cfg<KS, S, 32, 1, 4, 3, 2>(a, th, tw, tn, s) : launch_cfg<KS, S, 16, 1, 4, 3, 2>(a, th, tw, tn, s);
  }
  if (a.NS_total <= 2) {  // <= 32 couts: 2 cout blocks x 2 pixel halves
    pick_tile(a.Hout, a.Wout, small ? 64 : 128, &th, &tw, &tn);
    if (small) return kc32 ? launch_cfg<KS, S, 32, 2, 2, 1, 2>(a, th, tw, tn, s) : launch_cfg<KS, S, 16, 2, 2, 1, 2>(a, th, tw, tn, s);
    return kc32 ? launch_cfg<KS, S, 32, 2, 2, 1, 4>(a, th, tw, tn, s) : launch_cfg<KS, S, 16, 2, 2, 1, 4>(a, th, tw, tn, s);
  }
  pick_tile(a.Hout, a.Wout, small ? 64 : 128, &th, &tw, &tn);
  if (small) return kc32 ? launch_cfg<KS, S, 32, 4, 1, 1, 4>(a, th, tw, tn, s) : launch_cfg<KS, S, 16, 4, 1, 1, 4>(a, th, tw, tn, s);
  return kc32 ? launch_cfg<KS, S, 32, 4, 1, 1, 8>(a, th, tw, tn, s) : launch_cfg<KS, S, 16, 4, 1, 1, 8>(a, th, tw, tn, s);
}

int mval_launch_conv_mfma(const ConvArgs& a, hipStream_t s) {
  if (a.in_nchw || a.Cin % 16 != 0 || a.pad != (a.k - 1) / 2) return 1;
  if (a.k == 3 && a.stride == 1) return dispatch<3, 1>(a, s);
  if (a.k == 3 && a.stride == 2) return dispatch<3, 2>(a, s);
  if (a.k == 1 && a.stride == 1) return dispatch<1, 1>(a, s);
  if (a.k == 1 && a.stride == 2) return dispatch<1, 2>(a, s);
  return 1;
}
