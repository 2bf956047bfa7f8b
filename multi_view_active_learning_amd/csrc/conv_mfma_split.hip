// fp32-level conv on the 16-bit matrix cores: every fp32 value is split into a few 16-bit planes and the fp32 product
// is rebuilt from the plane products (exact in the MFMA, fp32 accumulate, small terms first).  The three-plane bf16
// split represents every fp32 value exactly; the two-plane fp16 split does NOT: it keeps 11 + 11 (+ sign) of the 24
// significand bits, a <= 2^-23 relative representation error per operand, and drops the lo x lo product.
//
// The exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, conv_mfma.hip) runs at 1/16 of the 16-bit MFMA rate.  Two splits:
//
//   PL = 2 ("h2", inference default): x * 2^s = xh + xl with xh, xl fp16 (11 + 11 significand bits; the
//     power-of-two scale 2^s puts the tensor's largest magnitude in [2^14, 2^15), so nothing overflows and the
//     low parts of every value that matters stay normal numbers):
//         a*b = ah*bh + (ah*bl + al*bh) + O(2^-22 |a||b|)
//     THREE v_mfma_f32_16x16x32_f16 per 32-deep k-step: 5.3x the fp32-MFMA rate.  Not bit-faithful to fp32
//     operands: per product the dropped terms are <= 2^-22 |a||b| worst case / ~2^-24 rms -- the size of one fp32
//     rounding.  What the tests hold it to is an OUTPUT bound: error against float64 no larger than 1.25x (rms) /
//     2.5x (max) the exact-fp32 MFMA chain's on the same problem (tests/test_gpu_models.py::
//     test_fused_conv_vs_torch_cpu), and no arg-max flips over 19 456 heat-maps against the exact-fp32 plan
//     (tests/test_gpu_p2.py census).  bench.py's exact_modes gives the cost of the bit-faithful kernels.
//     The activation scale comes from the producer: every kernel that writes an activation keeps max|x| of
//     the tensor in a 4-byte slot (wave maximum, then one conditional atomicMax per wave; order-independent,
//     so deterministic) and the consumer turns its exponent into 2^s while staging.  Weights are scaled by
//     their own maximum at pack time; both scales are powers of two and are undone exactly by the epilogue
//     (folded into the BatchNorm scale factor: acc * (2^-s * scale) rounds like (acc * 2^-s) * scale).
//   PL = 3 ("bf3", training plans and MVAL_CONV=bf3): x = xh + xm + xl with three bf16 values (3 x 8 bits, no
//     scaling needed: bf16 has the fp32 exponent range):
//         a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm) + O(2^-24 |a||b|)
//     SIX v_mfma_f32_16x16x32_bf16: 2.67x the fp32-MFMA rate.
//
// Same dataflow as conv_mfma.hip (patch with halo in LDS, taps = LDS offsets, weights pre-packed in fragment
// order and streamed straight to VGPRs one step ahead, BN / residual / ReLU epilogue through LDS) with these
// differences:
//   * the activation split happens ONCE per element while staging: PL 16-bit planes
//     [plane][pixel][32 ch + pad] in LDS (row stride 96 B: a 16-lane ds_read_b128 group covers all 64 banks);
//   * weights are split at pack time (mval_pack_conv_weights): [tap][cin/32][cout/16][plane][lane][8 x 16 bit];
//   * one LDS buffer + register prefetch of the next chunk (two barriers per 32-channel chunk).
// The measurement log of the variants that were tried and dropped is DESIGN.md appendix A.
// Used for 3x3 / 1x1 / 2x2-parity convs with cin % 32 == 0 (or cin = 48) when the plan selects
// MVAL_ALGO_MFMA_H2 / MVAL_ALGO_MFMA_BF3.
#include <stdlib.h>

#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BF_KC 32
// bf16 elements per LDS pixel row: 32 + 16 pad = 96 bytes, or 32 + 8 pad = 80 bytes for the stride-2 kernels (both
// conflict-free for 16 consecutive pixels; 80 bytes cost the 8-wide stride-1 tiles 5-10 % but let a fourth
// stride-2 workgroup -- 40 KB of planes -- fit a CU: 10-16 % on those layers)
#define BF_ROW_OF(S) ((S) == 2 ? 40 : 48)
#define OPAD 4

// One MFMA of the split product; fragments travel as untyped 128-bit registers (8 x 16 bit).
template <int PL>
__device__ __forceinline__ f32x4 mfma_split(const u32x4 a, const u32x4 b, const f32x4 c) {
  if constexpr (PL == 3)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// Products of a k-step in accumulation order (small terms first; the last one is the leading h x h product).
// Planes: 0 = h, 1 = m (PL = 3) / l (PL = 2), 2 = l (PL = 3).
template <int PL> __device__ __forceinline__ constexpr int split_np() { return PL == 3 ? 6 : 3; }
template <int PL> __device__ __forceinline__ constexpr int split_pa(int t) {  // activation plane of product t
  return PL == 3 ? (t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0) : (t == 0 ? 1 : 0);
}
template <int PL> __device__ __forceinline__ constexpr int split_pb(int t) {  // weight plane of product t
  return PL == 3 ? ((t == 0 || t == 3 || t == 5) ? 0 : t == 1 ? 2 : 1) : (t == 1 ? 1 : 0);
}

// fp16 two-way split of an already scaled value: h = RNE(v), l = RNE(v - h) (the subtraction is exact)
__device__ __forceinline__ void split2(const f32x4 v, f16x4& h, f16x4& l) {
  h = __builtin_convertvector(v, f16x4);
  const f32x4 r = v - __builtin_convertvector(h, f32x4);
  l = __builtin_convertvector(r, f16x4);
}

// 2^s as a float for the activation scale: the tensor's largest magnitude (its bits in *amax) lands in [2^14, 2^15);
// returns the factor and its exact inverse.  An all-zero tensor, or one that holds inf / NaN, is left unscaled.
__device__ __forceinline__ void split_act_scale(const unsigned* amax_row, float& mul, float& inv) {
  const int e = (int)((conv_amax_read(amax_row) >> 23) & 0xff);  // biased exponent of max |x| (denormal maxima: 0 -> unscaled)
  int s = (e == 0 || e == 255) ? 0 : 14 - (e - 127);
  s = max(-110, min(110, s));
  mul = __uint_as_float((unsigned)(127 + s) << 23);
  inv = __uint_as_float((unsigned)(127 - s) << 23);
}

__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const __bf16 hi = (__bf16)v[i];
    const float r1 = v[i] - (float)hi;
    const __bf16 mi = (__bf16)r1;
    const float r2 = r1 - (float)mi;
    h[i] = hi;
    m[i] = mi;
    l[i] = (__bf16)r2;
  }
}

// RS ("row sharing", 3x3 stride 1, 16-pixel-wide tiles): the wave's MS sub-tiles are consecutive
// tile rows, so the A fragment of patch row pr at column tap kx serves every (sub-tile ms, row tap
// ky) with ms + ky == pr.  Looping (kx, pr) instead of (tap, ms) reads (MS + 2) * 3 fragments per
// chunk instead of MS * 9 -- half the LDS traffic at MS = 4 -- with the three row taps' weights
// of one column live at a time.
// G > 1 (1x1 convs only): G 32-channel chunks are staged per barrier pair instead of one -- a 1x1 conv has a single
// tap per chunk, i.e. only MS * NT * 6 MFMAs (~0.2 us) between barriers and 8 KB of loads in flight per workgroup;
// the sub-chunks take the place of the taps in the inner loop.
template <int PL, int KS, int S, int WN, int WM, int NT, int MS, int NE, bool RS = false, bool PA = false, int G = 1>
__global__ __launch_bounds__(64 * WN * WM)
    __attribute__((amdgpu_waves_per_eu((KS == 1 && G == 2 && NT == 2) ? (PA ? 3 : 4) : (PA && KS >= 2 && NE <= 6) ? 3 : (PL == 2 && NE <= 6 && !PA) ? (NT * WM >= 4 ? 2 : WM > 1 ? 3 : 4) : 1, 8))) void conv_split_kernel(ConvArgs a) {
  static_assert(G == 1 || (KS == 1 && S == 1), "multi-chunk staging is for 1x1 convs");
  constexpr int NP = split_np<PL>();
  constexpr int TAPS = G > 1 ? G : KS * KS;
  constexpr int BF_ROW = BF_ROW_OF(S);
  constexpr int NTH = 64 * WN * WM;  // 256 threads, or 192 for the 48-channel-granular (HRNet-W48) tiles
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int MT = 16 * MS * WM;
  constexpr int NTILE = 16 * NT * WN;
  constexpr int LDW = NTILE + OPAD;
  constexpr int pad = KS / 2;  // 3x3: 1, 1x1: 0; the 2x2 parity kernels of a transposed conv: 1 (+ org_dy / org_dx)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int PH = (a.th - 1) * S + KS, PW = (a.tw - 1) * S + KS;
  if constexpr (KS == 2) {
    if (a.par_w_stride) {  // the four parity convs of a transposed conv share one launch
      const int parity = blockIdx.z;
      a.org_dy = a.ooy = parity >> 1;
      a.org_dx = a.oox = parity & 1;
      a.w += (int64_t)parity * a.par_w_stride;
    }
  }

  int t = blockIdx.x;
  const int txi = t % a.tiles_x;
  t /= a.tiles_x;
  const int tyi = t % a.tiles_y;
  const int n0 = (t / a.tiles_y) * a.tn;
  const int oy0 = tyi * a.th, ox0 = txi * a.tw;
  const int iy0 = oy0 * S - pad + a.org_dy, ix0 = ox0 * S - pad + a.org_dx;

  const int ns0 = (blockIdx.y * WN + wn) * NT;
  const bool wave_active = ns0 < a.NS_total;

  const int patch_px = a.tn * PH * PW;
  const int patch_e = patch_px * (BF_KC / 4);
  const int sub_bytes = patch_px * BF_ROW * 2;
  const int plane_bytes = sub_bytes * G;
  char* planes = smem_raw;  // [PL][G][patch_px][BF_ROW] 16-bit
  float in_mul = 1.f, unscale = 1.f;

  int abase[MS];  // byte offset inside a plane: pixel row + k quarter
#pragma unroll
  for (int ms = 0; ms < MS; ms++) {
    const int p = (wm * MS + ms) * 16 + (lane & 15);
    int tni, ty, tx;
    if (!conv_tile_decode(a, p, tni, ty, tx)) ty = tx = 0;  // padding slot of an odd tile: any in-range address
    abase[ms] = ((tni * PH + ty * S) * PW + tx * S) * (BF_ROW * 2) + (lane >> 4) * 16;
  }

  int goff[NE];
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e0 = tid + NTH * i;
    const int g = G > 1 ? e0 / (MT * 8) : 0;  // sub-chunk (1x1: the patch is the MT-pixel tile)
    const int e = e0 - g * (MT * 8);
    const int px = e >> 3, q = e & 7;
    const int prow = conv_div20(px, a.pw_magic);  // patch row over all images of the tile
    const int pxx = px - prow * PW;
    const int tni = conv_div20(prow, a.ph_magic);
    const int pyy = prow - tni * PH;
    const int iy = iy0 + pyy, ix = ix0 + pxx, n = n0 + tni;
    const int dsh = a.dil - 1;
    const int sy = iy >> dsh, sx = ix >> dsh;
    const int ay = sy << a.in_sub_log2, ax = sx << a.in_sub_log2;  // subsampled input view (stride-2 1x1 convs)
    const bool ok = e < patch_e && g < G && n < a.N && iy >= 0 && ix >= 0 && ((iy | ix) & dsh) == 0 && ay < a.Hin && ax < a.Win;
    goff[i] = ok ? ((n * a.Hin + ay) * a.Win + ax) * a.Cin + g * BF_KC + q * 4 : -1;
  }

  // PA ("precise accumulate", the training plans): two accumulators per output tile, the leading product
  // (h x h) and the five correction products.  The bf16 MFMA aligns its dot product to the accumulator and
  // drops (floors) what falls below a few guard bits; corrections are 2^-8 .. 2^-16 of the leading term, so
  // adding them to the big accumulator biases every output by about -1 ulp -- invisible in forward values
  // (same max / mean error as the exact-fp32 MFMA chain) but coherent, and amplified 25-100x in HRNet-W48's
  // training gradients (BatchNorm over a dozen samples per channel).  Summed separately and joined once in
  // the epilogue the bias is gone and the mean error drops 2.6x below the fp32 chain's; it costs 16
  // accumulator registers (~10 % time), so inference plans keep the single accumulator.
  f32x4 acc[MS][NT], accl[PA ? MS : 1][PA ? NT : 1];
#pragma unroll
  for (int ms = 0; ms < MS; ms++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      acc[ms][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (PA) accl[ms][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  // Weight fragments through a buffer descriptor: uniform base and (tap, chunk) block offset in SGPRs (soffset), a
  // 32-bit per-lane offset, the plane as immediate -- no 64-bit vector address arithmetic per fragment (it was
  // three v_mad_u64_u32 per fragment triple; these kernels are bound by vector-instruction issue).
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const int blk_bytes = a.NS_total * (PL * 64 * 16);
  int wlane[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) wlane[nt] = (min(ns0 + nt, a.NS_total - 1) * (PL * 64) + lane) * 16;
  auto bfrag = [&](int blk, int nt, int p) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane[nt] + p * 1024, blk * blk_bytes, 0);
  };
  const int nchunks = (a.Cin + BF_KC - 1) / BF_KC;  // the last chunk may be half empty (cin = 48)
  const int q4 = (tid & 7) * 4;                     // NTH % 8 == 0: a thread always stages the same channel quad
  f32x4 stage[NE];

  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const int gc = G > 1 ? ((tid + NTH * i) / (MT * 8)) * BF_KC : 0;
      stage[i] = (goff[i] >= 0 && c0 + gc + q4 < a.Cin) ? *reinterpret_cast<const f32x4*>(a.in + goff[i] + c0)
                                                        : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const int e0 = tid + NTH * i;
      const int g = G > 1 ? e0 / (MT * 8) : 0;
      const int e = e0 - g * (MT * 8);
      if (e < patch_e && g < G) {
        const int off = g * sub_bytes + (e >> 3) * (BF_ROW * 2) + (e & 7) * 8;
        if constexpr (PL == 3) {
          bf16x4 h, m, l;
          split3(stage[i], h, m, l);
          *reinterpret_cast<bf16x4*>(planes + off) = h;
          *reinterpret_cast<bf16x4*>(planes + plane_bytes + off) = m;
          *reinterpret_cast<bf16x4*>(planes + 2 * plane_bytes + off) = l;
        } else {
          f16x4 h, l;
          split2(stage[i] * in_mul, h, l);
          *reinterpret_cast<f16x4*>(planes + off) = h;
          *reinterpret_cast<f16x4*>(planes + plane_bytes + off) = l;
        }
      }
    }
  };

  load_chunk(0);
  if constexpr (PL == 2) {
    // one image per tile (the launcher refuses tn > 1): the image's own scale.  Read AFTER the first chunk's loads
    // are in flight: the two dependent L2 round trips (header, partials) then hide behind them (issued first they
    // cost 10 us on the 32-channel layers)
    split_act_scale(a.in_amax + (int64_t)n0 * a.in_amax_stride, in_mul, unscale);
    unscale *= *a.w_unscale;
  }
  store_chunk();
  __syncthreads();

  const int nstages = nchunks / G;  // the launcher only picks G > 1 when it divides the chunk count
  for (int ch = 0; ch < nstages; ch++) {
    const bool more = ch + 1 < nstages;
    if (more) load_chunk((ch + 1) * BF_KC * G);
    if constexpr (RS) {
      if (wave_active) {
        static_assert(!RS || (KS == 3 && S == 1), "row sharing is for 3x3 stride 1");
        u32x4 B[2][3][NT][PL];  // [column parity][row tap][cout sub-tile][plane]: the next column loads into the other half
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int p = 0; p < PL; p++) B[0][ky][nt][p] = bfrag((ky * 3) * nchunks + ch, nt, p);
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
#pragma unroll
          for (int pr = 0; pr < MS + 2; pr++) {
            if (kx + 1 < 3) {
              // next column's weights, each row tap fetched once the registers of the column before the
              // previous one's same-or-earlier tap are dead
              const int kyl = pr == 0 ? 0 : pr == MS ? 1 : pr == MS + 1 ? 2 : -1;
              if (kyl >= 0) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                  for (int p = 0; p < PL; p++) B[(kx + 1) & 1][kyl][nt][p] = bfrag((kyl * 3 + kx + 1) * nchunks + ch, nt, p);
              }
            }
            const char* ap = planes + abase[0] + (pr * PW + kx) * (BF_ROW * 2);
            u32x4 av[PL];
#pragma unroll
            for (int p = 0; p < PL; p++) av[p] = *reinterpret_cast<const u32x4*>(ap + p * plane_bytes);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
              // the products in small-to-large order, interleaved over the (up to three)
              // independent accumulators this fragment feeds
#pragma unroll
              for (int t = 0; t < NP; t++) {
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                  const int ms = pr - ky;
                  if (ms < 0 || ms >= MS) continue;
                  const int ai = split_pa<PL>(t), bi = split_pb<PL>(t);
                  if (!PA || t == NP - 1)
                    acc[ms][nt] = mfma_split<PL>(av[ai], B[kx & 1][ky][nt][bi], acc[ms][nt]);
                  else
                    accl[PA ? ms : 0][PA ? nt : 0] = mfma_split<PL>(av[ai], B[kx & 1][ky][nt][bi], accl[PA ? ms : 0][PA ? nt : 0]);
                }
              }
            }
          }
        }
      }
    } else if (wave_active) {
      // weight fragment blocks: ((tap * G32 + g32) * NS + ns) * 3 planes * 64 lanes (16-byte units)
      u32x4 b[2][NT][PL];  // [tap parity][cout sub-tile][plane]
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int p = 0; p < PL; p++) b[0][nt][p] = bfrag(ch * G, nt, p);
#pragma unroll
      for (int tap = 0; tap < TAPS; tap++) {
        if (tap + 1 < TAPS) {
          const int blk = G > 1 ? ch * G + tap + 1 : (tap + 1) * nchunks + ch;
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int p = 0; p < PL; p++) b[(tap + 1) & 1][nt][p] = bfrag(blk, nt, p);
        }
        const int toff = G > 1 ? tap * sub_bytes : ((tap / KS) * PW + (tap % KS)) * (BF_ROW * 2);
#pragma unroll
        for (int ms = 0; ms < MS; ms++) {
          const char* ap = planes + abase[ms] + toff;
          u32x4 av[PL];
#pragma unroll
          for (int p = 0; p < PL; p++) av[p] = *reinterpret_cast<const u32x4*>(ap + p * plane_bytes);
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            const u32x4* bc = b[tap & 1][nt];
            f32x4 c = PA ? accl[PA ? ms : 0][PA ? nt : 0] : acc[ms][nt];  // corrections, small terms first
#pragma unroll
            for (int t = 0; t < NP - 1; t++) c = mfma_split<PL>(av[split_pa<PL>(t)], bc[split_pb<PL>(t)], c);
            if constexpr (PA) {
              accl[ms][nt] = c;
              c = acc[ms][nt];
            }
            acc[ms][nt] = mfma_split<PL>(av[0], bc[0], c);
          }
        }
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- epilogue (as conv_mfma.hip) ----------------------------------------------------------
  float* ot = reinterpret_cast<float*>(smem_raw);
  if (wave_active) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int cl = (wn * NT + nt) * 16 + (lane & 15);
      const int c = blockIdx.y * NTILE + cl;
      // the power-of-two scales of the fp16 split are undone here: acc * (scale * 2^-s) rounds like (acc * 2^-s) * scale
      const float sc = c < a.Cout ? a.scale[c] * unscale : 0.f, sh = c < a.Cout ? a.shift[c] : 0.f;
#pragma unroll
      for (int ms = 0; ms < MS; ms++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int p = (wm * MS + ms) * 16 + (lane >> 4) * 4 + r;
          ot[p * LDW + cl] = (PA ? acc[ms][nt][r] + accl[PA ? ms : 0][PA ? nt : 0][r] : acc[ms][nt][r]) * sc + sh;
        }
    }
  }
  __syncthreads();
  conv_tile_store<MT, NTILE, NTH>(a, ot, tid, n0, oy0, ox0, blockIdx.y * NTILE);
}

static thread_local int g_bf3_dry = 0;

template <int PL, int KS, int S, int WN, int WM, int NT, int MS, int G = 1>
static int launch_split(ConvArgs a, int th, int tw, int tn, hipStream_t s) {
  a.th = th; a.tw = tw; a.tn = tn;
  a.tw_log2 = __builtin_ctz(tw);
  a.thw_log2 = __builtin_ctz(th * tw);
  a.tw_magic = 0;
  if ((tw & (tw - 1)) || (th & (th - 1))) {  // odd tile: th x tw of the MT slots, one image per tile
    if (tn != 1 || th * tw > 16 * MS * WM || G > 1) return 1;
    a.tw_magic = ((1u << 20) + tw - 1) / tw;
  }
  a.tiles_x = (a.Wout + tw - 1) / tw;
  a.tiles_y = (a.Hout + th - 1) / th;
  const int ngroups = (a.N + tn - 1) / tn;
  const int PH = (th - 1) * S + KS, PW = (tw - 1) * S + KS;
  a.pw_magic = ((1u << 20) + PW - 1) / PW;
  a.ph_magic = ((1u << 20) + PH - 1) / PH;
  constexpr int MT = 16 * MS * WM, NTILE = 16 * NT * WN;
  const int patch_px = tn * PH * PW;
  constexpr int BF_ROW = BF_ROW_OF(S);
  size_t smem = (size_t)PL * G * patch_px * BF_ROW * 2;
  const size_t otile = (size_t)CONV_OTILE_FLOATS(MT, NTILE) * sizeof(float);  // (with the max |x| reduction scratch)
  if (otile > smem) smem = otile;
  if (smem > 128 * 1024) return 1;
  constexpr int NTH = 64 * WN * WM;
  const int ne = (G * patch_px * 8 + NTH - 1) / NTH;
  if (ne > 10 || patch_px >= 4096 || tn * PH >= 4096) return 1;
  constexpr bool NE10 = MT >= 64 || S == 2;  // the small-problem tiles only exist with 6 staging slots
  if (!NE10 && ne > 6) return 1;
  if (G > 1 && (patch_px != MT || ((a.Cin + BF_KC - 1) / BF_KC) % G != 0)) return 1;
  if (PL == 2 && tn != 1) return 1;  // the fp16 split scales per image: one image per tile (maps under 8 rows use bf16x3)
  if (g_bf3_dry) return 0;
  constexpr bool PAOK = true;  // (both splits have the separate-correction-accumulator form)
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * ngroups), (unsigned)((a.NS_total + WN * NT - 1) / (WN * NT)),
            (KS == 2 && a.par_w_stride) ? 4u : 1u);
  conv_amax_prepare(a, a.tiles_x * a.tiles_y, (int)(grid.y * grid.z), s);
  conv_bn_part_prepare<MT, NTILE, NTH>(a, grid.x, grid.z);
  if constexpr (G > 1) {
    constexpr int NEG = (G * MT * 8 + NTH - 1) / NTH;
    if constexpr (PAOK)
      if (a.precise) {
        hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, NEG, false, true, G>), grid, dim3(NTH), smem, s, a);
        return 0;
      }
    hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, NEG, false, false, G>), grid, dim3(NTH), smem, s, a);
    return 0;
  }
  if constexpr (KS == 3 && S == 1) {
    if (tw == 16 && tn == 1 && ne <= 6) {  // row sharing
      if constexpr (PAOK)
        if (a.precise) {
          hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, 6, true, true>), grid, dim3(NTH), smem, s, a);
          return 0;
        }
      hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, 6, true>), grid, dim3(NTH), smem, s, a);
      return 0;
    }
  }
  if constexpr (PAOK)
    if (a.precise) {
      if (ne <= 6)
        hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, 6, false, true>), grid, dim3(NTH), smem, s, a);
      else if constexpr (NE10)
        hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, 10, false, true>), grid, dim3(NTH), smem, s, a);
      return 0;
    }
  if (ne <= 6)
    hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, 6>), grid, dim3(NTH), smem, s, a);
  else if constexpr (NE10)
    hipLaunchKernelGGL((conv_split_kernel<PL, KS, S, WN, WM, NT, MS, 10>), grid, dim3(NTH), smem, s, a);
  return 0;
}

static void split_pick_tile(int H, int W, int mt, int* th, int* tw, int* tn) {
  int w = (W > 8) ? 16 : 8;
  // widths that are no multiple of 16 (72 / 36 / 18 for HRNet-W48 at 384 x 288, 24 for PoseResNet at 256 x 192):
  // 8-wide tiles waste fewer columns than 16-wide ones, which outweighs the row-sharing kernel they give up
  // (HRNet-W48 forward 36.4 -> 32.0 ms on 64 images)
  if (W > 8 && mt >= 32 && ((W + 7) / 8) * 8 < ((W + 15) / 16) * 16) w = 8;
  int h = mt / w, n = 1;
  int hh = 1;
  while (hh < H) hh <<= 1;
  if (hh < h) {
    n = h / hh;
    h = hh;
  }
  *th = h; *tw = w; *tn = n;
}

// Maps like 24 x 18 or 12 x 9 (HRNet-W48 at 384 x 288) fill power-of-two tiles badly (25 % / 44 % padding): an odd
// tile -- tw = W, W / 2 or W / 3 columns by floor(mt / tw) rows, decoded with a magic divide -- replaces the
// power-of-two one in (th, tw, tn) when it needs at least 10 % fewer workgroups.
static void split_odd_tile(int H, int W, int N, int mt, int* th, int* tw, int* tn) {
  const int64_t wgs2 = (int64_t)((W + *tw - 1) / *tw) * ((H + *th - 1) / *th) * ((N + *tn - 1) / *tn);
  int64_t best = wgs2;
  int bth = 0, btw = 0;
  for (int d = 1; d <= 3; d++) {
    const int ctw = (W + d - 1) / d;
    if (ctw < 3 || ctw > mt / 2) continue;
    int cth = mt / ctw;
    if (cth > H) cth = H;
    const int64_t w = (int64_t)((W + ctw - 1) / ctw) * ((H + cth - 1) / cth) * N;
    if (w < best) {
      best = w;
      bth = cth; btw = ctw;
    }
  }
  if (btw && best * 10 <= wgs2 * 9) {
    *th = bth; *tw = btw; *tn = 1;
  }
}

// Tile pixels for a small problem: `px` output pixels x `groups` workgroups per pixel tile (see dispatch_split).
static int split_small_tile(int ks, int64_t px, int64_t groups) {
  const int64_t wgs64 = ((px + 63) / 64) * groups;
  if (ks == 3) return wgs64 <= 64 ? 16 : 64;
  int mt = 64;
  while (mt > 16 && ((px + mt - 1) / mt) * groups < 384) mt >>= 1;
  return mt;
}

template <int PL, int KS, int S>
static int dispatch_split(const ConvArgs& a, hipStream_t s) {
  int th, tw, tn;
  const int64_t px = (int64_t)a.N * a.Hout * a.Wout;
  // 4 sub-tiles per wave: 8 would need 220 VGPRs (one wave per SIMD) and measured slower
  if (a.NS_total <= 2) {
    const bool small = px < (int64_t)128 * 1024;
    split_pick_tile(a.Hout, a.Wout, small ? 64 : 128, &th, &tw, &tn);
    if (small) return launch_split<PL, KS, S, 2, 2, 1, 2>(a, th, tw, tn, s);
    return launch_split<PL, KS, S, 2, 2, 1, 4>(a, th, tw, tn, s);
  }
  // Small problems (a few images, or the deep low-resolution layers): with 64-pixel tiles fewer workgroups than
  // CUs exist while each walks all of cin serially.  1x1 convs and the 2x2 parity convs of a transposed conv go
  // down to 32- and 16-pixel tiles until there are a few hundred workgroups (PoseResNet-50 on 8 images:
  // 2048 -> 512 on 8x6 maps 78 -> 60 us, the 2048 -> 256 transposed conv 637 -> 147 us with its four parities in
  // one launch); the row-sharing 3x3 kernel gains only from 16-pixel tiles and only when fewer than a quarter of
  // the CUs had work (512 -> 512 on 8x6: 78 -> 64 us; 32-pixel tiles measured 1.5x SLOWER than 64-pixel ones).
  const int wn = (a.NS_total % 3 == 0 && a.NS_total % 4 != 0) ? 3 : 4;  // 48 / 96 output channels: three cout waves
  const int64_t cgroups = (a.NS_total + wn - 1) / wn, par = (KS == 2 && a.par_w_stride) ? 4 : 1;
  const int mt = split_small_tile(KS, px, cgroups * par);
  split_pick_tile(a.Hout, a.Wout, mt, &th, &tw, &tn);
  if constexpr (KS == 3)
    if (mt == 64) split_odd_tile(a.Hout, a.Wout, a.N, 64, &th, &tw, &tn);
  const int nch = (a.Cin + BF_KC - 1) / BF_KC;
  if constexpr (KS == 1) {
    // 1x1: two 32-channel chunks per barrier pair (16 KB of loads in flight per workgroup, twice the MFMAs
    // between barriers) and, from 128 output channels on, 128-cout tiles (half the staging redundancy):
    // 5-19 % on the bottleneck shapes of HRNet's layer1 / PoseResNet-50 (tools/conv1x1_sweep.py; four chunks
    // per stage cost a workgroup per CU in LDS and measured 20-30 % slower)
    if (wn == 4 && a.NS_total % 4 == 0 && nch % 2 == 0) {
      // 128-cout tiles only while they still give every CU a few workgroups
      const int64_t wgs128 = ((px + 63) / 64) * (a.NS_total / 8);
      if (a.NS_total % 8 == 0 && wgs128 >= 1024) return launch_split<PL, KS, S, 4, 1, 2, 4, 2>(a, th, tw, tn, s);
      if (mt == 16) return launch_split<PL, KS, S, 4, 1, 1, 1, 2>(a, th, tw, tn, s);
      if (mt == 32) return launch_split<PL, KS, S, 4, 1, 1, 2, 2>(a, th, tw, tn, s);
      return launch_split<PL, KS, S, 4, 1, 1, 4, 2>(a, th, tw, tn, s);
    }
  }
  if (wn == 3) {
    if (mt == 16) return launch_split<PL, KS, S, 3, 1, 1, 1>(a, th, tw, tn, s);
    if constexpr (KS != 3)
      if (mt == 32) return launch_split<PL, KS, S, 3, 1, 1, 2>(a, th, tw, tn, s);
    return launch_split<PL, KS, S, 3, 1, 1, 4>(a, th, tw, tn, s);
  }
  if (mt == 16) return launch_split<PL, KS, S, 4, 1, 1, 1>(a, th, tw, tn, s);
  if constexpr (KS != 3)
    if (mt == 32) return launch_split<PL, KS, S, 4, 1, 1, 2>(a, th, tw, tn, s);
  return launch_split<PL, KS, S, 4, 1, 1, 4>(a, th, tw, tn, s);
}

template <int PL>
static int launch_conv_split(const ConvArgs& a, hipStream_t s) {
  // cin a multiple of 32, or 48 (HRNet-W48's first branch: second chunk half empty, 25 % padding)
  if (a.in_nchw || (a.Cin % BF_KC != 0 && a.Cin != 48)) return 1;
  if ((int64_t)a.N * a.Hin * a.Win * a.Cin >= (int64_t)1 << 31) return 1;
  if (PL == 2 && !g_bf3_dry && (!a.in_amax || !a.w_unscale)) return 1;  // the fp16 split needs both scales
  // 1x1 channel GEMMs (bottleneck blocks, fuse up-paths): at the fp32-MFMA rate they are as
  // MFMA-bound as HBM-bound; here only HBM is left (202 vs 242 us for 256->64 on 128 64x64 maps, 3.3 TB/s;
  // keeping two chunks in flight instead of one measured slower, 226 us)
  if (a.k == 1 && a.pad == 0 && a.stride == 1 && a.dil == 1 && !a.out_nchw && (a.Cout & 15) == 0)
    return dispatch_split<PL, 1, 1>(a, s);
  // stride-2 1x1 conv (ResNet downsample paths) = stride-1 1x1 conv on the even pixels
  if (a.k == 1 && a.pad == 0 && a.stride == 2 && a.dil == 1 && !a.out_nchw && (a.Cout & 15) == 0 && a.in_sub_log2 == 0) {
    ConvArgs b = a;
    b.stride = 1;
    b.in_sub_log2 = 1;
    return dispatch_split<PL, 1, 1>(b, s);
  }
  // 2x2 parity kernel of a transposed conv (weights packed with mode 3, one parity's 4 taps at a.w)
  if (a.k == 2 && a.pad == 1 && a.stride == 1 && a.dil == 1 && !a.out_nchw && a.up == 0 && (a.Cout & 3) == 0)
    return dispatch_split<PL, 2, 1>(a, s);
  if (a.k != 3 || a.pad != 1) return 1;
  // dil == 2: data gradient of a stride-2 conv (dz read as a zero-dilated input: 3 of 4 staged values
  // are zeros, still ~2x the exact-fp32 MFMA kernel)
  if (a.stride == 1 && (a.dil == 1 || (a.dil == 2 && PL == 3))) return dispatch_split<PL, 3, 1>(a, s);
  // stride 2: the patch is ~4x larger per output pixel, so 32-pixel tiles (44 KB of LDS planes);
  // 64-pixel tiles (87 KB, 196 VGPRs) measured slower than the exact-fp32 kernel
  if (a.stride == 2 && a.dil == 1) {
    int th, tw, tn;
    split_pick_tile(a.Hout, a.Wout, 32, &th, &tw, &tn);
    if (a.NS_total <= 2) return launch_split<PL, 3, 2, 2, 2, 1, 1>(a, th, tw, tn, s);
    split_odd_tile(a.Hout, a.Wout, a.N, 32, &th, &tw, &tn);
    const bool w3 = a.NS_total % 3 == 0 && a.NS_total % 4 != 0;
    if (w3) return launch_split<PL, 3, 2, 3, 1, 1, 2>(a, th, tw, tn, s);
    return launch_split<PL, 3, 2, 4, 1, 1, 2>(a, th, tw, tn, s);
  }
  return 1;
}

// a.planes: 3 = bf16x3 (six products), 2 = fp16x2 (three products; needs a.in_amax and a.w_unscale)
int mval_launch_conv_split(const ConvArgs& a, hipStream_t s) {
  if (a.planes == 2) return launch_conv_split<2>(a, s);
  if (a.planes == 3) return launch_conv_split<3>(a, s);
  return 1;
}

int mval_conv_split_supported(const ConvArgs& a) {
  g_bf3_dry = 1;
  int rc = mval_launch_conv_split(a, nullptr);
  g_bf3_dry = 0;
  return rc == 0;
}

// ---- weight packing: [tap][cin/32][cout/16][plane][lane][8 x 16 bit] --------------------------
// Source element of packed position (tap t, cin ci, cout co) for the `mode`s of mval_pack_conv_weights.
__device__ __forceinline__ float split_w_at(const float* __restrict__ w, int mode, int cout, int cin, int k, int t, int ci, int co) {
  const int T = k * k;
  if (mode == 4) {
    // data gradient of Conv2d(k3, s2, p1) as four 2x2 stride-1 convs over dz, one per parity (py, px) of dx (packed with
    // k = 4: t = parity * 4 + (dy * 2 + dx); the SOURCE is the conv's own [cin' = cout][cout' = cin][3][3] weight).
    // dx[2a] = dz[a] W[1]; dx[2a + 1] = dz[a] W[2] + dz[a + 1] W[0].  The parity kernel's window is rows (a - 1, a) for
    // py = 0 and (a, a + 1) for py = 1 (as the transposed conv's, mode 3): py = 0 uses its second row only.
    const int pp = t >> 2, dy = (t >> 1) & 1, dx = t & 1;
    const int ky = (pp >> 1) ? (dy ? 0 : 2) : (dy ? 1 : -1), kx = (pp & 1) ? (dx ? 0 : 2) : (dx ? 1 : -1);
    if (ky < 0 || kx < 0) return 0.f;
    return w[((int64_t)ci * cout + co) * 9 + ky * 3 + kx];
  }
  if (mode == 3) {
    // ConvTranspose2d(k4, s2, p1) as four 2x2 stride-1 convs, one per output parity (py, px):
    // t = parity * 4 + (dy * 2 + dx); window row dy of parity py reads kernel row 3 - 2 dy (py = 0:
    // input rows a - 1, a) or 2 - 2 dy (py = 1: rows a, a + 1); columns alike
    const int pp = t >> 2, dy = (t >> 1) & 1, dx = t & 1;
    const int ky = (pp >> 1) ? 2 - 2 * dy : 3 - 2 * dy, kx = (pp & 1) ? 2 - 2 * dx : 3 - 2 * dx;
    return w[((int64_t)ci * cout + co) * 16 + ky * 4 + kx];
  }
  if (mode == 2) return w[((int64_t)ci * cout + co) * T + (T - 1 - t)];  // data-gradient form
  if (mode == 1) return w[((int64_t)ci * cout + co) * T + t];            // ConvTranspose2d layout
  return w[((int64_t)co * cin + ci) * T + t];
}

// wmul: power-of-two weight scale of the fp16 split (unused for PL = 3)
template <int PL>
__device__ __forceinline__ void pack_split_element(const float* __restrict__ w, unsigned short* __restrict__ p, int mode,
                                                   int cout, int cin, int k, int64_t i, float wmul) {
  const int G = (cin + 31) / 32, NS = (cout + 15) / 16;
  const int64_t total = (int64_t)k * k * G * NS * 512;  // (lane, j) pairs per block
  if (i >= total) return;
  const int j = (int)(i & 7);
  const int lane = (int)((i >> 3) & 63);
  int64_t r = i >> 9;
  const int ns = (int)(r % NS);
  const int g = (int)((r / NS) % G);
  const int t = (int)(r / ((int64_t)NS * G));
  const int co = ns * 16 + (lane & 15);
  const int ci = g * 32 + (lane >> 4) * 8 + j;
  const float v = (co < cout && ci < cin) ? split_w_at(w, mode, cout, cin, k, t, ci, co) : 0.f;
  const int64_t base = r * (PL * 512) + lane * 8 + j;  // PL planes x 512 values per block
  if constexpr (PL == 3) {
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    p[base] = __builtin_bit_cast(unsigned short, h);
    p[base + 512] = __builtin_bit_cast(unsigned short, m);
    p[base + 1024] = __builtin_bit_cast(unsigned short, l);
  } else {
    const float vs = v * wmul;
    const _Float16 h = (_Float16)vs;
    const _Float16 l = (_Float16)(vs - (float)h);
    p[base] = __builtin_bit_cast(unsigned short, h);
    p[base + 512] = __builtin_bit_cast(unsigned short, l);
  }
}

__global__ void pack_bf3_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int mode, int cout, int cin,
                                int k) {
  pack_split_element<3>(w, p, mode, cout, cin, k, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, 1.f);
}

// fp16 split: max |w| first (into trailer[1], zeroed by the launcher), then the pack scales every weight by the
// power of two that puts that maximum in [2^13, 2^14) and leaves the exact inverse in trailer[0] for the conv epilogue.
__global__ void weight_amax_kernel(const float* __restrict__ w, int64_t n, unsigned* __restrict__ slot) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(w[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}

__device__ __forceinline__ void split_weight_scale(unsigned amax_bits, float& mul, float& inv) {
  const int e = (int)((amax_bits >> 23) & 0xff);
  int s = (e == 0 || e == 255) ? 0 : 13 - (e - 127);
  s = max(-110, min(110, s));
  mul = __uint_as_float((unsigned)(127 + s) << 23);
  inv = __uint_as_float((unsigned)(127 - s) << 23);
}

__global__ void pack_h2_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int mode, int cout, int cin,
                               int k, float* __restrict__ trailer) {
  float mul, inv;
  split_weight_scale(reinterpret_cast<const unsigned*>(trailer)[1], mul, inv);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) trailer[0] = inv;
  pack_split_element<2>(w, p, mode, cout, cin, k, i, mul);
}

// All weight tensors of a training plan in ONE launch (they are re-packed after every optimizer step: ~580 tiny
// launches otherwise).  jobs / first_block live in device memory; block b belongs to the last job whose first
// block is <= b.
struct PackBf3Job {
  const float* w;
  unsigned short* p;
  int mode, cout, cin, k;
};
static_assert(sizeof(PackBf3Job) == 32, "mval_pack_job layout");

__global__ void pack_bf3_batch_kernel(const PackBf3Job* __restrict__ jobs, const int* __restrict__ first_block, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackBf3Job jb = jobs[lo];
  pack_split_element<3>(jb.w, jb.p, jb.mode, jb.cout, jb.cin, jb.k, (int64_t)(blockIdx.x - first_block[lo]) * blockDim.x + threadIdx.x, 1.f);
}

// Mixed batch: jobs with bit 8 of `mode` are fp16-split packings (trailer after the fragments; max |w| by the two kernels below first).
#define PACK_AMAX_PARTS 32
// max |w| of every job in two launches: PACK_AMAX_PARTS workgroups per job leave their partial maxima in the first floats of the job's
// OWN packed buffer (the packing kernel, next on the stream, overwrites them with fragments: no scratch memory), one wave per job
// folds them into the trailer.  (One workgroup per job walked the 590 k weights of a 256 -> 256 layer alone: 430 us per step.)
__global__ __launch_bounds__(256) void pack_amax_part_kernel(const PackBf3Job* __restrict__ jobs) {
  const PackBf3Job jb = jobs[blockIdx.x];
  if (!(jb.mode & 0x100)) return;
  __shared__ float red[4];
  const int64_t n = (int64_t)jb.cout * jb.cin * ((jb.mode & 0xff) == 4 ? 9 : jb.k * jb.k);  // (mode 4: a 3x3 source packed as k = 4)
  const int64_t per = (((n + PACK_AMAX_PARTS - 1) / PACK_AMAX_PARTS) + 3) & ~(int64_t)3;
  const int64_t lo = min(n, per * blockIdx.y), hi = min(n, lo + per);
  float m = 0.f;
  if ((reinterpret_cast<uintptr_t>(jb.w) & 15) == 0) {
    const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
    for (int64_t i = lo + 4 * threadIdx.x; i < hi4; i += 1024) {
      const float4 q = *reinterpret_cast<const float4*>(jb.w + i);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
    }
    for (int64_t i = hi4 + threadIdx.x; i < hi; i += 256) m = fmaxf(m, fabsf(jb.w[i]));
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, fabsf(jb.w[i]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) reinterpret_cast<float*>(jb.p)[blockIdx.y] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(64) void pack_amax_fold_kernel(const PackBf3Job* __restrict__ jobs) {
  const PackBf3Job jb = jobs[blockIdx.x];
  if (!(jb.mode & 0x100)) return;
  static_assert(PACK_AMAX_PARTS <= 64, "one wave folds a job's partial maxima");
  float m = threadIdx.x < PACK_AMAX_PARTS ? reinterpret_cast<const float*>(jb.p)[threadIdx.x] : 0.f;
  m = wave_max(m);
  if (threadIdx.x == 0) {
    const int G = (jb.cin + 31) / 32, NS = (jb.cout + 15) / 16;
    float* trailer = reinterpret_cast<float*>(jb.p) + (int64_t)jb.k * jb.k * G * NS * 512;
    float mul, inv;
    split_weight_scale(__float_as_uint(m), mul, inv);
    trailer[0] = inv;
    reinterpret_cast<unsigned*>(trailer)[1] = __float_as_uint(m);
    trailer[2] = trailer[3] = 0.f;
  }
}

__global__ void pack_split_batch_kernel(const PackBf3Job* __restrict__ jobs, const int* __restrict__ first_block, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackBf3Job jb = jobs[lo];
  const int64_t i = (int64_t)(blockIdx.x - first_block[lo]) * blockDim.x + threadIdx.x;
  if (jb.mode & 0x100) {
    const int G = (jb.cin + 31) / 32, NS = (jb.cout + 15) / 16;
    const float* trailer = reinterpret_cast<const float*>(jb.p) + (int64_t)jb.k * jb.k * G * NS * 512;
    float mul, inv;
    split_weight_scale(reinterpret_cast<const unsigned*>(trailer)[1], mul, inv);
    pack_split_element<2>(jb.w, jb.p, jb.mode & 0xff, jb.cout, jb.cin, jb.k, i, mul);
  } else {
    pack_split_element<3>(jb.w, jb.p, jb.mode, jb.cout, jb.cin, jb.k, i, 1.f);
  }
}

int mval_pack_split_batch(const void* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_amax_part_kernel, dim3((unsigned)n_jobs, PACK_AMAX_PARTS), dim3(256), 0, s, reinterpret_cast<const PackBf3Job*>(jobs_dev));
  hipLaunchKernelGGL(pack_amax_fold_kernel, dim3((unsigned)n_jobs), dim3(64), 0, s, reinterpret_cast<const PackBf3Job*>(jobs_dev));
  hipLaunchKernelGGL(pack_split_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, s,
                     reinterpret_cast<const PackBf3Job*>(jobs_dev), first_block_dev, n_jobs);
  return 0;
}

int mval_pack_bf3_batch(const void* jobs_dev, const int* first_block_dev, int n_jobs, int total_blocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_bf3_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, s,
                     reinterpret_cast<const PackBf3Job*>(jobs_dev), first_block_dev, n_jobs);
  return 0;
}

int mval_pack_bf3(int mode, const float* w, float* packed, int cout, int cin, int k, hipStream_t s) {
  const int G = (cin + 31) / 32, NS = (cout + 15) / 16;
  const int64_t total = (int64_t)k * k * G * NS * 512;
  hipLaunchKernelGGL(pack_bf3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w,
                     reinterpret_cast<unsigned short*>(packed), mode, cout, cin, k);
  return 0;
}

// fp16-split packing: `packed` holds k*k*G*NS*512 floats of fragments followed by a 4-float trailer
// [2^-s (the epilogue's factor), bits of max |w|, 0, 0].
int mval_pack_h2(int mode, const float* w, float* packed, int cout, int cin, int k, hipStream_t s) {
  const int G = (cin + 31) / 32, NS = (cout + 15) / 16;
  const int64_t total = (int64_t)k * k * G * NS * 512;
  float* trailer = packed + total;  // 2 planes x 512 halves = 512 floats per block
  (void)hipMemsetAsync(trailer, 0, 4 * sizeof(float), s);
  const int64_t nw = (int64_t)cout * cin * (mode == 4 ? 9 : k * k);
  hipLaunchKernelGGL(weight_amax_kernel, dim3((unsigned)((nw + 1023) / 1024 > 256 ? 256 : (nw + 1023) / 1024)), dim3(256), 0, s, w, nw,
                     reinterpret_cast<unsigned*>(trailer) + 1);
  hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w,
                     reinterpret_cast<unsigned short*>(packed), mode, cout, cin, k, trailer);
  return 0;
}
