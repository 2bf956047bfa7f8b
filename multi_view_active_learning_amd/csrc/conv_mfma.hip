// Fused conv (3x3 / 1x1, stride 1 / 2) + BatchNorm + residual(s) + ReLU (+ nearest upsample,
// + NCHW store) on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Replaces the cuDNN/ATen convs + separate BN/ReLU/add/upsample framework ops of the
// reference's HRNet / PoseResNet forward (pose_estimators/hrnet.py:36-52,75-95,199-287).
//
// Implicit GEMM, M = output pixels, N = couts, K = taps x cin:
//   * a workgroup (4 waves) owns a tile of tn x th x tw output pixels x (16*NT*WN) couts; the
//     INPUT PATCH of that tile (with its halo) for a chunk of KC input channels is staged once
//     in LDS ([pixel][KC+8] floats; the +8 pad spreads a ds_read_b128 lane group over all 64
//     banks) and re-used by all k*k taps, so global->LDS traffic is ~1.3x the activations
//     instead of 9x (im2col); a tap is a constant LDS offset;
//   * chunks are double-buffered: the global loads of chunk c+1 are issued into registers
//     before the MFMAs of chunk c and written to the other LDS buffer after them (one
//     barrier per chunk, "issue early / write late");
//   * A fragments: one ds_read_b128 per 16-pixel sub-tile per 16 cin;
//   * B fragments (weights) are pre-packed on device in fragment order
//     [tap][cin/16][cout/16][lane][4] (mval_pack_conv_weights): ONE fully coalesced 1 KiB
//     global_load_dwordx4 per wave per (tap, 16 cin), straight to VGPRs (weights are shared by
//     every workgroup -> L2 hits), prefetched one step ahead; waves split N first so each
//     weight block is fetched once per workgroup;
//   * epilogue: y = acc*scale + shift goes through LDS so that the residual loads and the
//     stores are float4 along channels (full 128-B lines) -- (+res1) (+res2), ReLU, optional
//     2^up nearest replication, NHWC or NCHW store.
// fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s peak); the kernel is MFMA-bound.
#include <stdlib.h>

#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define KPAD 8
#define OPAD 4

template <int KS, int S, int KC, int WN, int WM, int NT, int MS, int NE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KCP = KC + KPAD;
  constexpr int GC = KC / 16;       // 16-cin groups per chunk
  constexpr int C4 = KC / 4;        // float4 per pixel per chunk
  constexpr int MT = 16 * MS * WM;  // pixels per tile
  constexpr int NTILE = 16 * NT * WN;
  constexpr int LDW = NTILE + OPAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int PH = (a.th - 1) * S + KS, PW = (a.tw - 1) * S + KS;
  const int pad = a.pad;  // (KS - 1) / 2 for the 3x3 / 1x1 convs; 2 / 1 for the k4 transposed-conv forms

  // tile origin
  int t = blockIdx.x;
  const int txi = t % a.tiles_x;
  t /= a.tiles_x;
  const int tyi = t % a.tiles_y;
  const int n0 = (t / a.tiles_y) * a.tn;
  const int oy0 = tyi * a.th, ox0 = txi * a.tw;
  const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;

  const int ns0 = (blockIdx.y * WN + wn) * NT;  // first 16-cout block of this wave
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

  // staging descriptors: element e = tid + 256*i -> patch pixel e / C4, float4 e % C4
  const int patch_px = a.tn * PH * PW;
  const int patch_e = patch_px * C4;
  const int patch_floats = patch_px * KCP;
  constexpr int NEA = NE > 0 ? NE : 1;
  int goff[NEA];  // float offset of the element in the input for chunk 0, -1 outside the image
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e = tid + 256 * i;
    const int px = e / C4, q = e % C4;
    const int prow = conv_div20(px, a.pw_magic);  // multiply-shift instead of runtime divides
    const int pxx = px - prow * PW;
    const int tni = conv_div20(prow, a.ph_magic);
    const int pyy = prow - tni * PH;
    const int iy = iy0 + pyy, ix = ix0 + pxx, n = n0 + tni;
    // dil == 2: (iy, ix) index the zero-dilated input; only even positions carry data
    const int dsh = a.dil - 1;
    const int sy = iy >> dsh, sx = ix >> dsh;
    const bool ok = e < patch_e && n < a.N && iy >= 0 && ix >= 0 && ((iy | ix) & dsh) == 0 && sy < a.Hin && sx < a.Win;
    goff[i] = ok ? ((n * a.Hin + sy) * a.Win + sx) * a.Cin + q * 4 : -1;
  }

  f32x4 acc[MS][NT];
#pragma unroll
  for (int ms = 0; ms < MS; ms++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[ms][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const f32x4* wq = reinterpret_cast<const f32x4*>(a.w) + lane;
  const int nchunks = a.Cin / KC;
  f32x4 stage[NEA];

  // generic (NE == 0) staging for rare tile shapes whose patch exceeds the register budget:
  // plain load -> LDS store loop, single buffer, two barriers per chunk
  auto stage_generic = [&](int c0) {
    for (int e = tid; e < patch_e; e += 256) {
      const int px = e / C4, q = e % C4;
      int r = px;
      const int pxx = r % PW;
      r /= PW;
      const int pyy = r % PH;
      const int tni = r / PH;
      const int iy = iy0 + pyy, ix = ix0 + pxx, n = n0 + tni;
      const int dsh = a.dil - 1;
      const int sy = iy >> dsh, sx = ix >> dsh;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n < a.N && iy >= 0 && ix >= 0 && ((iy | ix) & dsh) == 0 && sy < a.Hin && sx < a.Win)
        v = *reinterpret_cast<const f32x4*>(a.in + (((int64_t)n * a.Hin + sy) * a.Win + sx) * a.Cin + c0 + q * 4);
      *reinterpret_cast<f32x4*>(smem + px * KCP + q * 4) = v;
    }
  };

  // prologue: chunk 0 -> LDS buffer 0
  if constexpr (NE > 0) {
#pragma unroll
    for (int i = 0; i < NE; i++)
      stage[i] = goff[i] >= 0 ? *reinterpret_cast<const f32x4*>(a.in + goff[i]) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const int e = tid + 256 * i;
      if (e < patch_e) *reinterpret_cast<f32x4*>(smem + (e / C4) * KCP + (e % C4) * 4) = stage[i];
    }
  } else {
    stage_generic(0);
  }
  __syncthreads();

  for (int ch = 0; ch < nchunks; ch++) {
    const float* cur = NE > 0 ? smem + (ch & 1) * patch_floats : smem;
    const bool more = ch + 1 < nchunks;
    if (NE > 0 && more) {  // issue the next chunk's global loads before the MFMAs
      const int c1 = (ch + 1) * KC;
#pragma unroll
      for (int i = 0; i < NE; i++)
        stage[i] = goff[i] >= 0 ? *reinterpret_cast<const f32x4*>(a.in + goff[i] + c1) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (wave_active) {
      const int g0 = ch * GC;
      f32x4 bcur[NT], bnxt[NT];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        const int ns = min(ns0 + nt, a.NS_total - 1);
        bcur[nt] = wq[((int64_t)g0 * a.NS_total + ns) * 64];
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
        for (int ms = 0; ms < MS; ms++) af[ms] = *reinterpret_cast<const f32x4*>(cur + abase[ms] + toff);
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
    if (NE == 0 && more) {
      __syncthreads();
      stage_generic((ch + 1) * KC);
    }
    if (NE > 0 && more) {  // write the staged chunk into the other buffer (last read two barriers ago)
      float* nxt = smem + ((ch + 1) & 1) * patch_floats;
#pragma unroll
      for (int i = 0; i < NE; i++) {
        const int e = tid + 256 * i;
        if (e < patch_e) *reinterpret_cast<f32x4*>(nxt + (e / C4) * KCP + (e % C4) * 4) = stage[i];
      }
    }
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------
  // C layout: col(cout) = lane & 15, row(pixel) = (lane >> 4) * 4 + reg.  BN in registers,
  // then through LDS ([pixel][NTILE+4]) so that global accesses run along channels.
  if (wave_active) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int cl = (wn * NT + nt) * 16 + (lane & 15);
      const int c = blockIdx.y * NTILE + cl;
      const float sc = c < a.Cout ? a.scale[c] : 0.f, sh = c < a.Cout ? a.shift[c] : 0.f;
#pragma unroll
      for (int ms = 0; ms < MS; ms++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int p = (wm * MS + ms) * 16 + (lane >> 4) * 4 + r;
          smem[p * LDW + cl] = acc[ms][nt][r] * sc + sh;
        }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int cl = (wn * NT + nt) * 16 + (lane & 15);
#pragma unroll
      for (int ms = 0; ms < MS; ms++)
#pragma unroll
        for (int r = 0; r < 4; r++) smem[((wm * MS + ms) * 16 + (lane >> 4) * 4 + r) * LDW + cl] = 0.f;
    }
  }
  __syncthreads();
  conv_tile_store<MT, NTILE, 256, KS == 1 && S == 1>(a, smem, tid, n0, oy0, ox0, blockIdx.y * NTILE);  // (1x1: the heat-map layers, with arg-max keys)
}

static thread_local int g_dry_run = 0;  // feasibility query: run the selection logic, launch nothing

template <int KS, int S, int KC, int WN, int WM, int NT, int MS>
static int launch_cfg(ConvArgs a, int th, int tw, int tn, hipStream_t s) {
  a.th = th; a.tw = tw; a.tn = tn;
  a.tw_log2 = __builtin_ctz(tw);
  a.thw_log2 = __builtin_ctz(th * tw);
  a.tiles_x = (a.Wout + tw - 1) / tw;
  a.tiles_y = (a.Hout + th - 1) / th;
  const int ngroups = (a.N + tn - 1) / tn;
  const int PH = (th - 1) * S + KS, PW = (tw - 1) * S + KS;
  a.pw_magic = ((1u << 20) + PW - 1) / PW;
  a.ph_magic = ((1u << 20) + PH - 1) / PH;
  if (tn * PH * PW >= 4096 || PW >= 256 || PH >= 256) return 1;  // conv_div20 range
  constexpr int MT = 16 * MS * WM, NTILE = 16 * NT * WN;
  const int patch_floats = tn * PH * PW * (KC + KPAD);
  const int ne = (tn * PH * PW * (KC / 4) + 255) / 256;
  const int nbuf = (a.Cin / KC > 1 && ne <= 10) ? 2 : 1;  // the generic staging path is single-buffered
  size_t smem = (size_t)patch_floats * nbuf * sizeof(float);
  const size_t otile = (size_t)CONV_OTILE_FLOATS(MT, NTILE) * sizeof(float);  // (with the max |x| reduction scratch)
  if (otile > smem) smem = otile;
  if (smem > 128 * 1024) return 1;  // LDS is 160 KiB per CU on gfx950
  if (g_dry_run) return 0;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * ngroups), (unsigned)((a.NS_total + WN * NT - 1) / (WN * NT)));
  conv_amax_prepare(a, a.tiles_x * a.tiles_y, (int)grid.y, s);
  conv_bn_part_prepare<MT, NTILE, 256>(a, grid.x, 1);
  // staging registers are sized at compile time (runtime-indexed arrays would go to scratch)
  if (ne <= 4)
    hipLaunchKernelGGL((conv_mfma_kernel<KS, S, KC, WN, WM, NT, MS, 4>), grid, dim3(256), smem, s, a);
  else if (ne <= 6)
    hipLaunchKernelGGL((conv_mfma_kernel<KS, S, KC, WN, WM, NT, MS, 6>), grid, dim3(256), smem, s, a);
  else if (ne <= 10)
    hipLaunchKernelGGL((conv_mfma_kernel<KS, S, KC, WN, WM, NT, MS, 10>), grid, dim3(256), smem, s, a);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<KS, S, KC, WN, WM, NT, MS, 0>), grid, dim3(256), smem, s, a);
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
  // stride-2 3x3 patches are ~4x larger per output pixel: 16-channel chunks keep two LDS
  // buffers under 64 KB so that two workgroups still fit on a CU
  const int kc32 = (a.Cin % 32 == 0) && !(S == 2 && KS == 3);
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

// k = 4 forms of ConvTranspose2d(k4, s2, p1) (PoseResNet head, pose_resnet.py:88-117): the forward
// is a stride-1 conv over the zero-dilated input with pad 2 (tap-flipped, channel-swapped weights,
// pack mode 2), its data gradient a plain stride-2 conv with pad 1.  One tile shape each: 64 pixels
// x 64 couts, 16-channel chunks.
template <int S>
static int dispatch_k4(const ConvArgs& a, hipStream_t s) {
  int th, tw, tn;
  pick_tile(a.Hout, a.Wout, 64, &th, &tw, &tn);
  return launch_cfg<4, S, 16, 4, 1, 1, 4>(a, th, tw, tn, s);
}

int mval_launch_conv_mfma(const ConvArgs& a, hipStream_t s) {
  if (a.in_nchw || a.Cin % 16 != 0) return 1;
  // 32-bit element offsets inside the staging loop
  if ((int64_t)a.N * a.Hin * a.Win * a.Cin >= (int64_t)1 << 31) return 1;  // 32-bit element offsets: the engine slices larger batches
  if (a.k == 4 && a.stride == 1 && a.dil == 2 && a.pad == 2) return dispatch_k4<1>(a, s);
  if (a.k == 4 && a.stride == 2 && a.dil == 1 && a.pad == 1) return dispatch_k4<2>(a, s);
  if (a.pad != (a.k - 1) / 2) return 1;
  if (a.k == 3 && a.stride == 1) return dispatch<3, 1>(a, s);
  if (a.k == 3 && a.stride == 2) return dispatch<3, 2>(a, s);
  if (a.k == 1 && a.stride == 1) return dispatch<1, 1>(a, s);
  if (a.k == 1 && a.stride == 2) return dispatch<1, 2>(a, s);
  return 1;
}

int mval_conv_mfma_supported(const ConvArgs& a) {
  g_dry_run = 1;
  int rc = mval_launch_conv_mfma(a, nullptr);
  g_dry_run = 0;
  return rc == 0;
}
