// Weight gradients of the fused conv operators (the wgrad half of K7; autograd of the
// reference's nn.Conv2d under strategy.py:478 ``batch_loss.backward()``).
//
//   dW[co][ci][tap] = sum over (n, oy, ox) of  x[n, oy*s - p + ty, ox*s - p + tx, ci] * dz[n, oy, ox, co]
//
// GEMM with M = cin, N = cout, K = output pixels (hundreds of thousands) on
// v_mfma_f32_16x16x4_f32 (exact fp32):
//   * a workgroup owns a (32 cin x 32 cout) block for ALL k*k taps -- 4 waves = 2 cin tiles x
//     2 cout tiles, each wave keeps k*k accumulator tiles in registers -- and walks a strided
//     share of the pixel tiles (split-K over workgroups); the next tile's operands are fetched
//     global -> registers during the current tile's MFMA loop;
//   * per pixel tile the x patch (with halo, 32 cin) and the dz tile (32 cout) are staged in LDS
//     with row stride 48 floats (== 16 mod 32 banks, so the 4 pixel-quads of a ds_read_b32 do
//     not collide); a tap is a constant LDS offset into the patch, so each 4-pixel step costs
//     one B read + k*k A reads for k*k MFMAs;
//   * partial sums go to per-split slabs [split][tap][cin][cout]; a second kernel reduces the
//     slabs in float64 and writes torch's [cout][cin][kh][kw] layout (deterministic, no atomics).
// Tiny operators (3-channel stem, 19-joint final layer) use a direct VALU kernel.
#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int mval_launch_wgrad_bf3(const float* x, const float* dz, float* slabs, int N, int Hin, int Win, int Cin, int Hout,
                          int Wout, int Cout, int k, int stride, int max_slabs, const unsigned* x_amax,
                          const unsigned* dz_amax, hipStream_t s);
int mval_launch_wgrad_bf3_p2(const float* x, const void* x_p2, const unsigned* x_p2_rows, const float* dz, float* slabs, int N, int Hin,
                             int Win, int Cin, int Hout, int Wout, int Cout, int k, int stride, int max_slabs, const unsigned* x_amax,
                             const unsigned* dz_amax, hipStream_t s, const void* dz_p2 = nullptr, const unsigned* dz_p2_rows = nullptr);
int mval_wgrad_bf3_covers(int Cin, int Cout, int k, int stride);
extern "C" int mval_conv_wgrad_split_covers(int cin, int cout, int k, int stride) { return mval_wgrad_bf3_covers(cin, cout, k, stride); }
extern "C" int mval_conv_wgrad_p2_covers(int cin, int cout, int k, int stride) { return mval_wgrad_bf3_covers(cin, cout, k, stride) && (cin & 7) == 0; }
// x as P2 planes for the next mval_conv_wgrad_on call of this thread (net_train.hip sets it per operator; nullptr = fp32 x)
static thread_local const void* g_wgrad_x_p2 = nullptr;
static thread_local const unsigned* g_wgrad_x_p2_rows = nullptr;
void mval_conv_wgrad_set_p2_x(const void* planes, const unsigned* rows) {
  g_wgrad_x_p2 = planes;
  g_wgrad_x_p2_rows = rows;
}
static thread_local const void* g_wgrad_dz_p2 = nullptr;
static thread_local const unsigned* g_wgrad_dz_p2_rows = nullptr;
void mval_conv_wgrad_set_p2_dz(const void* planes, const unsigned* rows) {
  g_wgrad_dz_p2 = planes;
  g_wgrad_dz_p2_rows = rows;
}
  // conv_wgrad_bf3.hip

#define WG_CB 32   // cin / cout block
#define WG_LD 48   // LDS row stride (floats)

struct WgradArgs {
  const float* x;    // NHWC (N, Hin, Win, Cin)
  const float* dz;   // NHWC (N, Hout, Wout, Cout)
  float* slabs;      // [PS][T][Cin][Cout]
  int N, Hin, Win, Cin, Hout, Wout, Cout;
  int pad;
  int th, tw, tn, tw_log2, thw_log2, tiles_x, tiles_y, ntiles, PS;
};

template <int KS, int S, int MT, int NEX>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = KS * KS;
  constexpr int NEZ = MT * (WG_CB / 4) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci_t = wave & 1, co_t = wave >> 1;
  const int ci0 = blockIdx.y * WG_CB, co0 = blockIdx.z * WG_CB;
  const int PH = (a.th - 1) * S + KS, PW = (a.tw - 1) * S + KS;
  const int patch_px = a.tn * PH * PW;
  const int patch_e = patch_px * (WG_CB / 4);
  float* patch = smem;                    // [patch_px][WG_LD]
  float* dzt = smem + patch_px * WG_LD;   // [MT][WG_LD]

  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // tile-invariant position of every staged element (row << 20 | column << 8 | image)
  int pkx[NEX], pkz[NEZ];
#pragma unroll
  for (int i = 0; i < NEX; i++) {
    const int e = tid + 256 * i;
    int r = e >> 3;
    const int pxx = r % PW;
    r /= PW;
    pkx[i] = e < patch_e ? ((r % PH) << 20) | (pxx << 8) | (r / PH) : -1;
  }
#pragma unroll
  for (int i = 0; i < NEZ; i++) {
    const int p = (tid + 256 * i) >> 3;
    const int rem = p & ((1 << a.thw_log2) - 1);
    pkz[i] = ((rem >> a.tw_log2) << 20) | ((rem & ((1 << a.tw_log2) - 1)) << 8) | (p >> a.thw_log2);
  }
  const int q4 = (tid & 7) * 4;
  const bool cx_ok = ci0 + q4 < a.Cin, cz_ok = co0 + q4 < a.Cout;

  // next tile's x patch and dz tile travel global -> registers while the current one is multiplied
  f32x4 xr[NEX], zr[NEZ];
  auto load_tile = [&](int t) {
    const int txi = t % a.tiles_x;
    t /= a.tiles_x;
    const int oy0 = (t % a.tiles_y) * a.th, ox0 = txi * a.tw, n0 = (t / a.tiles_y) * a.tn;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
#pragma unroll
    for (int i = 0; i < NEX; i++) {
      const int iy = iy0 + (pkx[i] >> 20), ix = ix0 + ((pkx[i] >> 8) & 0xfff), n = n0 + (pkx[i] & 0xff);
      const bool ok = pkx[i] >= 0 && cx_ok && n < a.N && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      xr[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + (((int64_t)n * a.Hin + iy) * a.Win + ix) * a.Cin + ci0 + q4)
                 : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NEZ; i++) {
      const int y = oy0 + (pkz[i] >> 20), x = ox0 + ((pkz[i] >> 8) & 0xfff), n = n0 + (pkz[i] & 0xff);
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (cz_ok && n < a.N && y < a.Hout && x < a.Wout) {
        const float* src = a.dz + (((int64_t)n * a.Hout + y) * a.Wout + x) * a.Cout + co0 + q4;
        if ((a.Cout & 3) == 0) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {  // 19-joint final layer: rows are not 16-byte aligned
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (co0 + q4 + j < a.Cout) v[j] = src[j];
        }
      }
      zr[i] = v;
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) load_tile(tile);
  for (; tile < a.ntiles; tile += a.PS) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NEX; i++) {
      const int e = tid + 256 * i;
      if (e < patch_e) *reinterpret_cast<f32x4*>(patch + (e >> 3) * WG_LD + q4) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < NEZ; i++) *reinterpret_cast<f32x4*>(dzt + ((tid + 256 * i) >> 3) * WG_LD + q4) = zr[i];
    __syncthreads();
    if (tile + a.PS < a.ntiles) load_tile(tile + a.PS);
    // K loop over the tile's pixels, 4 per MFMA (pixel = 4*step + (lane >> 4))
#pragma unroll 2
    for (int st = 0; st < MT / 4; st++) {
      const int p = st * 4 + (lane >> 4);
      const int tni = p >> a.thw_log2;
      const int rem = p & ((1 << a.thw_log2) - 1);
      const int ty = rem >> a.tw_log2, tx = rem & ((1 << a.tw_log2) - 1);
      const float b = dzt[p * WG_LD + co_t * 16 + (lane & 15)];
      const float* ap = patch + ((tni * PH + ty * S) * PW + tx * S) * WG_LD + ci_t * 16 + (lane & 15);
#pragma unroll
      for (int t2 = 0; t2 < T; t2++) {
        const float av = ap[((t2 / KS) * PW + (t2 % KS)) * WG_LD];
        acc[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[t2], 0, 0, 0);
      }
    }
  }
  // slab[ps][t][ci][co]; C layout: row (cin) = (lane >> 4) * 4 + r, col (cout) = lane & 15
  const int co = co0 + co_t * 16 + (lane & 15);
#pragma unroll
  for (int t2 = 0; t2 < T; t2++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int ci = ci0 + ci_t * 16 + (lane >> 4) * 4 + r;
      if (ci < a.Cin && co < a.Cout)
        a.slabs[(((int64_t)blockIdx.x * T + t2) * a.Cin + ci) * a.Cout + co] = acc[t2][r];
    }
}

// dW[co][ci][t] = sum_ps slab[ps][t][ci][co].  A workgroup owns 64 consecutive outputs and walks the
// slabs with `parts` = blockDim / 64 lanes per output (up to 16, four loads in flight each): the
// 32-channel layers have only 9216 outputs but 512 slabs, and with 4 lanes per output their
// reduction was a 128-deep dependent chain on 144 workgroups (45 us for 19 MB).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slabs, int PS, int T, int Cin,
                                                            int Cout, float* __restrict__ dw) {
  __shared__ double red[1024];
  const int64_t n = (int64_t)T * Cin * Cout;
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6, parts = blockDim.x >> 6;
  double s = 0;
  if (i < n) {
    int p = part;
    for (; p + 7 * parts < PS; p += 8 * parts) {  // eight slabs per round trip (the 32-channel layers: 768 slabs over 16 lanes per output)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = slabs[(int64_t)(p + u * parts) * n + i];
      s += (((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3])) + (((double)v[4] + (double)v[5]) + ((double)v[6] + (double)v[7]));
    }
    for (; p + 3 * parts < PS; p += 4 * parts) {
      const float v0 = slabs[(int64_t)p * n + i], v1 = slabs[(int64_t)(p + parts) * n + i];
      const float v2 = slabs[(int64_t)(p + 2 * parts) * n + i], v3 = slabs[(int64_t)(p + 3 * parts) * n + i];
      s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; p < PS; p += parts) s += (double)slabs[(int64_t)p * n + i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && i < n) {
    s = 0;
    for (int k = 0; k < parts; k++) s += red[k * 64 + threadIdx.x];
    const int co = (int)(i % Cout);
    const int ci = (int)((i / Cout) % Cin);
    const int t = (int)(i / ((int64_t)Cout * Cin));
    dw[((int64_t)co * Cin + ci) * T + t] = (float)s;
  }
}

// ---- stem conv1 (3 NCHW input channels, 3x3 or 7x7 stride 2) on the matrix cores ---------------
// GEMM rows = (cin, tap) = 27 / 147 (two / ten 16-row tiles), columns = cout (one 16-wide tile
// per wave), K = output pixels.  The NCHW patch sits in LDS as [ci][PH][PW]; a row's (ci, ky, kx)
// is a constant offset, the pixel another, so A operands are plain ds_read_b32 gathers.
#define WS_LDZ 80  // dz tile row stride (floats), 64 cout + 16: == 16 mod 32 banks

template <int KS, int MT>
__global__ __launch_bounds__(256) void conv_wgrad_stem_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = KS * KS, ROWS = 3 * T, NR = (ROWS + 15) / 16;  // (cin, tap) rows: 27 -> 2 tiles, 147 -> 10
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int PH = (a.th - 1) * 2 + KS, PW = (a.tw - 1) * 2 + KS;
  float* patch = smem;                 // [3][PH][PW]
  float* dzt = smem + ((3 * PH * PW + 3) & ~3);  // [MT][WS_LDZ]
  const int co0 = wave * 16;
  const bool wave_active = co0 < a.Cout;
  int roff[NR];
  bool rok[NR];
#pragma unroll
  for (int mt = 0; mt < NR; mt++) {
    const int i = mt * 16 + (lane & 15);
    rok[mt] = i < ROWS;
    const int ci = i / T, t = i % T;
    roff[mt] = rok[mt] ? (ci * PH + t / KS) * PW + t % KS : 0;
  }
  f32x4 acc[NR];
#pragma unroll
  for (int mt = 0; mt < NR; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int c4n = a.Cout >> 2;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += a.PS) {
    int t = tile;
    const int txi = t % a.tiles_x;
    t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oy0 = tyi * a.th, ox0 = txi * a.tw;
    const int iy0 = oy0 * 2 - a.pad, ix0 = ox0 * 2 - a.pad;
    __syncthreads();
    for (int e = tid; e < 3 * PH * PW; e += 256) {
      const int pxx = e % PW;
      const int r = e / PW;
      const int pyy = r % PH, ci = r / PH;
      const int iy = iy0 + pyy, ix = ix0 + pxx;
      patch[e] = (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                     ? a.x[(((int64_t)n * 3 + ci) * a.Hin + iy) * a.Win + ix]
                     : 0.f;
    }
    for (int e = tid; e < MT * c4n; e += 256) {
      const int p = e / c4n, q = e % c4n;
      const int y = oy0 + (p >> a.tw_log2), x = ox0 + (p & ((1 << a.tw_log2) - 1));
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (y < a.Hout && x < a.Wout)
        v = *reinterpret_cast<const f32x4*>(a.dz + (((int64_t)n * a.Hout + y) * a.Wout + x) * a.Cout + q * 4);
      *reinterpret_cast<f32x4*>(dzt + p * WS_LDZ + q * 4) = v;
    }
    __syncthreads();
    if (wave_active) {
#pragma unroll 2
      for (int st = 0; st < MT / 4; st++) {
        const int p = st * 4 + (lane >> 4);
        const int poff = ((p >> a.tw_log2) * 2) * PW + (p & ((1 << a.tw_log2) - 1)) * 2;
        const float b = dzt[p * WS_LDZ + co0 + (lane & 15)];
#pragma unroll
        for (int mt = 0; mt < NR; mt++) {
          const float av = rok[mt] ? patch[roff[mt] + poff] : 0.f;
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[mt], 0, 0, 0);
        }
      }
    }
  }
  // slab[ps][t][ci][co]
  if (wave_active) {
    const int co = co0 + (lane & 15);
#pragma unroll
    for (int mt = 0; mt < NR; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = mt * 16 + (lane >> 4) * 4 + r;
        if (i < ROWS && co < a.Cout)
          a.slabs[(((int64_t)blockIdx.x * T + i % T) * 3 + i / T) * a.Cout + co] = acc[mt][r];
      }
  }
}

// ---- direct wgrad for whatever is left (odd shapes; not on the HRNet / PoseResNet hot path) ------
struct WgradDirectArgs {
  const float* x;
  const float* dz;
  float* slabs;  // [PS][T*Cin*Cout] in [t][ci][co] order
  int N, Hin, Win, Cin, Hout, Wout, Cout, k, stride, pad, x_nchw, PS;
};

#define WD_EPT 8

__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(WgradDirectArgs a) {
  const int E = a.k * a.k * a.Cin * a.Cout;
  const int64_t M = (int64_t)a.N * a.Hout * a.Wout;
  float acc[WD_EPT];
  int xo[WD_EPT], co[WD_EPT], dy[WD_EPT], dx[WD_EPT], ci[WD_EPT];
#pragma unroll
  for (int i = 0; i < WD_EPT; i++) {
    const int e = threadIdx.x + 256 * i;
    acc[i] = 0.f;
    co[i] = e % a.Cout;
    ci[i] = (e / a.Cout) % a.Cin;
    const int t = e / (a.Cout * a.Cin);
    dy[i] = t / a.k - a.pad;
    dx[i] = t % a.k - a.pad;
    xo[i] = e < E;
  }
  const int64_t per = (M + a.PS - 1) / a.PS;
  const int64_t p0 = blockIdx.x * per, p1 = min(M, p0 + per);
  for (int64_t p = p0; p < p1; p++) {
    const int x = (int)(p % a.Wout);
    const int y = (int)((p / a.Wout) % a.Hout);
    const int n = (int)(p / ((int64_t)a.Wout * a.Hout));
#pragma unroll
    for (int i = 0; i < WD_EPT; i++) {
      if (!xo[i]) continue;
      const int iy = y * a.stride + dy[i], ix = x * a.stride + dx[i];
      if (iy < 0 || iy >= a.Hin || ix < 0 || ix >= a.Win) continue;
      const float xv = a.x_nchw ? a.x[(((int64_t)n * a.Cin + ci[i]) * a.Hin + iy) * a.Win + ix]
                                : a.x[(((int64_t)n * a.Hin + iy) * a.Win + ix) * a.Cin + ci[i]];
      acc[i] = fmaf(xv, a.dz[p * a.Cout + co[i]], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < WD_EPT; i++) {
    const int e = threadIdx.x + 256 * i;
    if (e < E) a.slabs[(int64_t)blockIdx.x * E + e] = acc[i];
  }
}

static void wg_pick_tile(int H, int W, int mt, int* th, int* tw, int* tn) {
  int w = (W > 8) ? 16 : 8;
  int h = mt / w, n = 1;
  int hh = 1;
  while (hh < H) hh <<= 1;
  if (hh < h) {
    n = h / hh;
    h = hh;
  }
  *th = h; *tw = w; *tn = n;
}

static int wg_splits(int cin, int cout, int target) {
  const int cb = ((cin + WG_CB - 1) / WG_CB) * ((cout + WG_CB - 1) / WG_CB);
  int ps = target / cb;
  if (ps < 1) ps = 1;
  if (ps > 512) ps = 512;
  return ps;
}

int mval_conv_wgrad_on(const float* x, const float* dz, float* dw, float* ws, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                       int k, int stride, int pad, int x_nchw, const uint32_t* x_amax_row, const uint32_t* dz_amax_row, hipStream_t s);

extern "C" size_t mval_conv_wgrad_workspace_floats(int cin, int cout, int k) {
  return (size_t)wg_splits(cin, cout, 1024) * k * k * cin * cout;  // PS slabs of [tap][cin][cout]
}

// x NHWC (or NCHW when x_nchw), dz NHWC, dw [cout][cin][k][k]; ws >= mval_conv_wgrad_workspace_floats
extern "C" int mval_conv_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int Hin, int Win, int Cin,
                               int Hout, int Wout, int Cout, int k, int stride, int pad, int x_nchw, void* stream) {
  return mval_conv_wgrad_scaled(x, dz, dw, ws, N, Hin, Win, Cin, Hout, Wout, Cout, k, stride, pad, x_nchw, nullptr, nullptr, stream);
}

extern "C" int mval_conv_wgrad_scaled(const float* x, const float* dz, float* dw, float* ws, int N, int Hin, int Win, int Cin,
                                      int Hout, int Wout, int Cout, int k, int stride, int pad, int x_nchw,
                                      const uint32_t* x_amax_row, const uint32_t* dz_amax_row, void* stream) {
  return mval_conv_wgrad_on(x, dz, dw, ws, N, Hin, Win, Cin, Hout, Wout, Cout, k, stride, pad, x_nchw, x_amax_row, dz_amax_row, mval_stream(stream));
}

// ---- (round 6) the slab reductions of a whole backward call in ONE launch ------------------------------------------------------------
// Every op of a call writes its slabs into its own region of the workspace; mval_conv_wgrad_defer(&job) makes the NEXT
// mval_conv_wgrad_on record its reduction there instead of launching it; mval_wgrad_reduce_jobs runs up to 64 recorded reductions as one
// kernel (a job's outputs are walked by the same `parts` lanes per output, in the same order, as its own launch would: same bits).
struct WgradReduceJob {
  const float* slabs;
  float* dw;
  int PS, T, Cin, Cout, parts;
};
struct WgradReduceJobs {
  int n;
  int blk0[65];  // first block of job j; blk0[n] = total
  WgradReduceJob j[64];
};
static thread_local WgradReduceJob* g_wgrad_defer = nullptr;
void mval_conv_wgrad_defer(WgradReduceJob* out) { g_wgrad_defer = out; }

__global__ __launch_bounds__(1024) void wgrad_reduce_jobs_kernel(const WgradReduceJobs J) {
  __shared__ double red[1024];
  int jn = 0;
  while (jn + 1 < J.n && (int)blockIdx.x >= J.blk0[jn + 1]) jn++;  // (uniform: <= 63 scalar steps)
  const WgradReduceJob job = J.j[jn];
  const int64_t n = (int64_t)job.T * job.Cin * job.Cout;
  const int64_t i = (int64_t)((int)blockIdx.x - J.blk0[jn]) * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6, parts = job.parts, PS = job.PS;
  const float* __restrict__ slabs = job.slabs;
  double s = 0;
  if (i < n && part < parts) {
    int p = part;
    for (; p + 7 * parts < PS; p += 8 * parts) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = slabs[(int64_t)(p + u * parts) * n + i];
      s += (((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3])) + (((double)v[4] + (double)v[5]) + ((double)v[6] + (double)v[7]));
    }
    for (; p + 3 * parts < PS; p += 4 * parts) {
      const float v0 = slabs[(int64_t)p * n + i], v1 = slabs[(int64_t)(p + parts) * n + i];
      const float v2 = slabs[(int64_t)(p + 2 * parts) * n + i], v3 = slabs[(int64_t)(p + 3 * parts) * n + i];
      s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; p < PS; p += parts) s += (double)slabs[(int64_t)p * n + i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && i < n) {
    s = 0;
    for (int k = 0; k < parts; k++) s += red[k * 64 + threadIdx.x];
    const int co = (int)(i % job.Cout);
    const int ci = (int)((i / job.Cout) % job.Cin);
    const int t = (int)(i / ((int64_t)job.Cout * job.Cin));
    job.dw[((int64_t)co * job.Cin + ci) * job.T + t] = (float)s;
  }
}

// jobs[0 .. n): recorded by mval_conv_wgrad_defer; launched in chunks of 64 on `s`
int mval_wgrad_reduce_jobs(const WgradReduceJob* jobs, int n, hipStream_t s) {
  for (int j0 = 0; j0 < n; j0 += 64) {
    WgradReduceJobs J;
    J.n = n - j0 < 64 ? n - j0 : 64;
    int b = 0;
    for (int j = 0; j < J.n; j++) {
      J.j[j] = jobs[j0 + j];
      J.blk0[j] = b;
      b += (int)(((int64_t)J.j[j].T * J.j[j].Cin * J.j[j].Cout + 63) / 64);
    }
    J.blk0[J.n] = b;
    hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3((unsigned)b), dim3(1024), 0, s, J);
    MVAL_CHECK_LAUNCH("mval_wgrad_reduce_jobs");
  }
  return 0;
}

// (net_train.hip's entry: the HIP stream as such.  Round 4 also ran the slab reduction on a side stream beside the op's data gradient:
// two event hand-overs per operator cost more than the 7 us reduction they hid, C3 79.9 vs 77.7 ms -- removed in round 5.)
int mval_conv_wgrad_on(const float* x, const float* dz, float* dw, float* ws, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                       int k, int stride, int pad, int x_nchw, const uint32_t* x_amax_row, const uint32_t* dz_amax_row, hipStream_t s) {
  WgradReduceJob* defer = g_wgrad_defer;  // (consumed by THIS call, whatever it returns)
  g_wgrad_defer = nullptr;
  MVAL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && k > 0 && stride > 0, "mval_conv_wgrad: bad dims");
  const int T = k * k;
  const int64_t n_out = (int64_t)T * Cin * Cout;
  // k = 4, stride 2, pad 1: the weight gradient of ConvTranspose2d(k4, s2, p1) with the roles of the
  // activations swapped (x := dz of the transposed conv, dz := its input)
  const bool mfma = !x_nchw && (Cin & 3) == 0 && Cin >= 16 &&
                    (((k == 1 || k == 3) && (stride == 1 || stride == 2) && pad == k / 2) || (k == 4 && stride == 2 && pad == 1));
  const bool stem = x_nchw && Cin == 3 && (k == 3 || k == 7) && stride == 2 && pad == k / 2 && (Cout & 15) == 0 && Cout <= 64;
  int PS = 0;
  const void* xp2 = g_wgrad_x_p2;
  const unsigned* xp2_rows = g_wgrad_x_p2_rows;
  g_wgrad_x_p2 = nullptr;
  g_wgrad_x_p2_rows = nullptr;
  const void* zp2 = g_wgrad_dz_p2;
  const unsigned* zp2_rows = g_wgrad_dz_p2_rows;
  g_wgrad_dz_p2 = nullptr;
  g_wgrad_dz_p2_rows = nullptr;
  if (!x_nchw && pad == k / 2)  // split-bf16 kernel (conv_wgrad_bf3.hip); 0 = shape not covered
    PS = mval_launch_wgrad_bf3_p2(x, xp2, xp2_rows, dz, ws, N, Hin, Win, Cin, Hout, Wout, Cout, k, stride, wg_splits(Cin, Cout, 1024),
                                  x_amax_row, dz_amax_row, s, zp2, zp2_rows);
  MVAL_REQUIRE(PS >= 0, "mval_conv_wgrad: x given as its producer's raw z (mval_conv_wgrad_set_z_x) needs the split kernel's 3x3 stride-1 form with dz as planes (k%d s%d cin%d cout%d)", k, stride, Cin, Cout);
  MVAL_REQUIRE(!(xp2 || zp2) || PS > 0, "mval_conv_wgrad: the P2 form of x needs the split kernel (k%d s%d cin%d cout%d) and dz's magnitude row", k, stride, Cin, Cout);
  if (PS > 0) {
    MVAL_CHECK_LAUNCH("mval_conv_wgrad/bf3");
  } else if (stem) {
    WgradArgs a;
    a.x = x; a.dz = dz; a.slabs = ws;
    a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout; a.pad = pad;
    a.th = 8; a.tw = 16; a.tn = 1;
    a.tw_log2 = 4; a.thw_log2 = 7;
    a.tiles_x = (Wout + 15) / 16;
    a.tiles_y = (Hout + 7) / 8;
    a.ntiles = a.tiles_x * a.tiles_y * N;
    PS = a.ntiles < 512 ? a.ntiles : 512;
    a.PS = PS;
    const int PH = 14 + k, PW = 30 + k;
    const size_t smem = (size_t)(((3 * PH * PW + 3) & ~3) + 128 * WS_LDZ) * sizeof(float);
    if (k == 3)
      hipLaunchKernelGGL((conv_wgrad_stem_kernel<3, 128>), dim3(PS), dim3(256), smem, s, a);
    else
      hipLaunchKernelGGL((conv_wgrad_stem_kernel<7, 128>), dim3(PS), dim3(256), smem, s, a);
    MVAL_CHECK_LAUNCH("mval_conv_wgrad/stem");
  } else if (mfma) {
    WgradArgs a;
    a.x = x; a.dz = dz; a.slabs = ws;
    a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout; a.pad = pad;
    const int mt = (stride == 2 || (int64_t)N * Hout * Wout < 65536) ? 64 : 128;
    wg_pick_tile(Hout, Wout, mt, &a.th, &a.tw, &a.tn);
    a.tw_log2 = __builtin_ctz(a.tw);
    a.thw_log2 = __builtin_ctz(a.th * a.tw);
    a.tiles_x = (Wout + a.tw - 1) / a.tw;
    a.tiles_y = (Hout + a.th - 1) / a.th;
    a.ntiles = a.tiles_x * a.tiles_y * ((N + a.tn - 1) / a.tn);
    // workgroups in flight: 128-pixel tiles hold 59 KB of LDS (2 per CU -> 512 resident: one round and
    // half the slab traffic of 1024), 64-pixel tiles fit 4 per CU
    PS = wg_splits(Cin, Cout, mt == 128 ? 512 : 1024);
    if (PS > a.ntiles) PS = a.ntiles;
    a.PS = PS;
    const int PH = (a.th - 1) * stride + k, PW = (a.tw - 1) * stride + k;
    size_t smem = (size_t)(a.tn * PH * PW + mt) * WG_LD * sizeof(float);
    MVAL_REQUIRE(smem <= 128 * 1024, "mval_conv_wgrad: tile does not fit LDS");
    dim3 grid(PS, (Cin + WG_CB - 1) / WG_CB, (Cout + WG_CB - 1) / WG_CB);
#define WG_LAUNCH(KS_, S_, MT_, NEX_)                                                                      \
  do {                                                                                                     \
    MVAL_REQUIRE(a.tn * PH * PW * 8 <= 256 * NEX_, "mval_conv_wgrad: patch of %d pixels exceeds the staging registers", \
                 a.tn * PH * PW);                                                                          \
    hipLaunchKernelGGL((conv_wgrad_kernel<KS_, S_, MT_, NEX_>), grid, dim3(256), smem, s, a);              \
  } while (0)
    if (k == 3 && stride == 1 && mt == 128) WG_LAUNCH(3, 1, 128, 6);
    else if (k == 3 && stride == 1) WG_LAUNCH(3, 1, 64, 6);
    else if (k == 3 && stride == 2) WG_LAUNCH(3, 2, 64, 12);
    else if (k == 4) WG_LAUNCH(4, 2, 64, 12);
    else if (k == 1 && stride == 1 && mt == 128) WG_LAUNCH(1, 1, 128, 4);
    else if (k == 1 && stride == 1) WG_LAUNCH(1, 1, 64, 2);
    else WG_LAUNCH(1, 2, 64, 8);
#undef WG_LAUNCH
    MVAL_CHECK_LAUNCH("mval_conv_wgrad/mfma");
  } else {
    MVAL_REQUIRE(n_out <= 256 * WD_EPT, "mval_conv_wgrad: operator too large for the direct kernel (%lld outputs)",
                 (long long)n_out);
    WgradDirectArgs a;
    a.x = x; a.dz = dz; a.slabs = ws;
    a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
    a.k = k; a.stride = stride; a.pad = pad; a.x_nchw = x_nchw;
    int64_t M = (int64_t)N * Hout * Wout;
    PS = (int)(M < 512 ? M : 512);
    a.PS = PS;
    hipLaunchKernelGGL(conv_wgrad_direct_kernel, dim3(PS), dim3(256), 0, s, a);
    MVAL_CHECK_LAUNCH("mval_conv_wgrad/direct");
  }
  int parts = PS / 8;  // >= 8 slabs per lane
  parts = parts < 1 ? 1 : parts > 16 ? 16 : parts;
  if (defer) {  // the caller runs this reduction with the others of its backward call (mval_wgrad_reduce_jobs)
    defer->slabs = ws; defer->dw = dw; defer->PS = PS; defer->T = T; defer->Cin = Cin; defer->Cout = Cout; defer->parts = parts;
    return 0;
  }
#ifdef MVAL_TRAIN_ABLATE  // (measurement build: MVAL_TRAIN_ABL bit 2 -- no slab reductions at all: the upper bound of batching them; gradients are garbage)
  extern int g_train_ablate;
  if (!(g_train_ablate & 4))
#endif
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n_out + 63) / 64)), dim3(64 * parts), 0, s, ws, PS, T, Cin, Cout, dw);
  MVAL_CHECK_LAUNCH("mval_conv_wgrad/reduce");
  return 0;
}
