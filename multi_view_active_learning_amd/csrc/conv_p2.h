// "P2" activations: every fp32 activation value kept as the PAIR of fp16 planes the fp16x2-split MFMA convs consume,
// written once by the producer's epilogue (csrc/conv_p2.hip, conv_block_p2.hip).
//
//   tensor of N images, C channels (C % 8 == 0), H x W:
//       halves  [n][plane p = 0 (h), 1 (l)][C / 8][H][W][8]          4 bytes per element, like fp32
//       x[n][c][y][x] = (h + l) * 2^-s[n]         h = RNE_fp16(x * 2^s), l = RNE_fp16(x * 2^s - h)
//   row of the tensor's image n (P2_ROW = 512 dwords; the fp32-activation kernels' rows are MVAL_AMAX_ROW = 4096):
//       row[0 .. P2_SLOTS - 1] = partial max |x| (float bits; one slot per producing workgroup, unused slots stay 0:
//                                the rows are zeroed once when the plan is built and a slot only ever belongs to one
//                                workgroup, so a consumer reads a FIXED number of slots -- no count, no dependent load),
//       row[P2_INV_SLOT] = float bits of 2^-s[n]
//
// Why: with fp32 NHWC activations every CONSUMER re-splits each element it stages (12 vector instructions per
// float4, two 8-byte LDS stores) -- 2.4x per element for a 3x3 conv's halo and once more per cout group.  With
// P2 the consumer's staging is a 16-byte copy (global -> register -> LDS, no arithmetic), and the channel-blocked
// layout makes the LDS image [8-channel block][pixel][16 B] conflict-free for ds_read_b128 at ANY pixel offset
// (a 16-lane fragment read covers 256 consecutive bytes per block), so the 3x3 taps are plain immediate offsets.
//
// The scale 2^s[n] must be known BEFORE the producer's first store, i.e. before the image's true maximum exists.
// It comes from a rigorous per-image bound of the fused operator out = act((bn(conv(x)) + r1) + r2):
//       |out| <= A * max|x| + B + max|r1| + max|r2|,   A = max_c |scale_c| * sum_k |w_ck|,  B = max_c |shift_c|
// (A, B per layer at parameter-refresh time; the maxima are the exact per-image maxima the producers keep).  The
// bound is typically 2^4 .. 2^7 above the true maximum; fp16's 5 exponent bits absorb that: 2^s puts the BOUND in
// [2^13, 2^14), every value above 2^-3 (scaled) keeps all 22 significand bits of the pair and smaller ones an
// absolute error <= 2^-25, i.e. <= 2^-38 of the bound.  Per IMAGE, so results do not depend on the batch.
#pragma once
#include "conv_common.h"

#define P2_ROW 512    // dwords per (tensor, image) row
#define P2_SLOTS 256  // partial maxima per image: 4 per lane of the reading wave
#define P2_INV_SLOT (P2_ROW - 1)

typedef _Float16 p2_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 p2_f16x8 __attribute__((ext_vector_type(8)));
typedef float p2_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int p2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int p2_u32x2 __attribute__((ext_vector_type(2)));

// 2^s that puts `bound` in [2^13, 2^14) and its inverse (zero / inf / NaN bound: unscaled)
__device__ __forceinline__ void p2_scale_of(float bound, float& mul, float& inv) {
  const int e = (int)((__float_as_uint(bound) >> 23) & 0xff);
  int s = (e == 0 || e == 255) ? 0 : 13 - (e - 127);
  s = max(-110, min(110, s));
  mul = __uint_as_float((unsigned)(127 + s) << 23);
  inv = __uint_as_float((unsigned)(127 - s) << 23);
}

// max(v, floor) that PROPAGATES NaN (IEEE 754-2019 maximum: one v_maximum3_f32 on gfx950), for the ReLU / "no activation"
// clamps of the epilogues: fmaxf returns the other operand for a NaN, which turned a diverged model's NaNs into 0 (ReLU) or
// -inf where torch -- and the h2 / bf3 / fp32 plans -- hand them on.
__device__ __forceinline__ float p2_max_nan(float v, float floor_) { return __builtin_elementwise_maximum(v, floor_); }

__device__ __forceinline__ void p2_split(const p2_f32x4 v, p2_f16x4& h, p2_f16x4& l) {
  h = __builtin_convertvector(v, p2_f16x4);
#ifdef P2_NO_FMA_MIX
  l = __builtin_convertvector(v - __builtin_convertvector(h, p2_f32x4), p2_f16x4);
#else
  // v - float(h) as ONE v_fma_mix_f32 per value (fma(float(h), -1, v): the same singly-rounded difference as convert +
  // subtract; the compiler does not form it by itself): 8 instead of 12 instructions per split
  const p2_u32x2 hu = __builtin_bit_cast(p2_u32x2, h);
  p2_f32x4 r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(hu.x), "v"(v.x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(hu.x), "v"(v.y));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r.z) : "v"(hu.y), "v"(v.z));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.w) : "v"(hu.y), "v"(v.w));
  l = __builtin_convertvector(r, p2_f16x4);
#endif
}
__device__ __forceinline__ p2_f32x4 p2_join(const p2_f16x4 h, const p2_f16x4 l) {
#ifdef P2_NO_FMA_MIX
  return __builtin_convertvector(h, p2_f32x4) + __builtin_convertvector(l, p2_f32x4);  // exact (22 bits)
#else
  // float(h) * 1 + float(l) in one v_fma_mix_f32 per value (both operands read as fp16 halves): 4 instead of 12
  const p2_u32x2 hu = __builtin_bit_cast(p2_u32x2, h), lu = __builtin_bit_cast(p2_u32x2, l);
  p2_f32x4 r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r.x) : "v"(hu.x), "v"(lu.x));
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r.y) : "v"(hu.x), "v"(lu.x));
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r.z) : "v"(hu.y), "v"(lu.y));
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r.w) : "v"(hu.y), "v"(lu.y));
  return r;
#endif
}

// A workgroup's max |x| of what it stored of image n -> its slot of the row.  More producing workgroups per image than
// slots (inputs above ~512 x 512): they fold into slot % P2_SLOTS with atomicMax (the launcher zeroes the rows first).
__device__ __forceinline__ void p2_slot_put(unsigned* row, int slot, int total, unsigned bits) {
  if (total <= P2_SLOTS) row[slot] = bits;
  else atomicMax(row + slot % P2_SLOTS, bits);
}
// Consumer side: the lanes' share of the partial slots of image n (4 independent loads) ...
struct P2RowRegs {
  unsigned v[P2_SLOTS / 64];
  unsigned inv;
};
__device__ __forceinline__ void p2_row_request(const unsigned* rows, int n, P2RowRegs& r) {
  const unsigned* row = rows + (int64_t)n * P2_ROW;
#pragma unroll
  for (int i = 0; i < P2_SLOTS / 64; i++) r.v[i] = row[i * 64 + (threadIdx.x & 63)];
  r.inv = row[P2_INV_SLOT];
}
// Maximum over the wave of non-negative float bits (they order like unsigned integers), uniform result: four DPP
// steps inside the 16-lane rows, then the four rows through scalar registers -- a dozen instructions; the
// __shfl_xor butterfly is six dependent ds_bpermute round trips (~0.3 us each time it is used in an epilogue).
__device__ __forceinline__ unsigned p2_wave_umax(unsigned m) {
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, false));  // row_half_mirror
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, false));  // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)m, 0), b = (unsigned)__builtin_amdgcn_readlane((int)m, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)m, 32), d = (unsigned)__builtin_amdgcn_readlane((int)m, 48);
  return max(max(a, b), max(c, d));
}
// Sum over each 16-lane row of the wave, in every lane of the row (four DPP steps, fixed order: deterministic)
__device__ __forceinline__ float p2_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}
// Maximum over each 16-lane row of the wave, in every lane of the row
__device__ __forceinline__ float p2_row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
  return v;
}
// ... and their maximum (uniform over the wave)
__device__ __forceinline__ float p2_row_amax(const P2RowRegs& r) {
  unsigned m = r.v[0];
#pragma unroll
  for (int i = 1; i < P2_SLOTS / 64; i++) m = max(m, r.v[i]);
  return __uint_as_float(p2_wave_umax(m));
}

// Launcher side of the persistent kernels: resident workgroups per CU of one kernel instantiation, from its LDS size and its
// register count (allocation granule 8, 512 per SIMD lane; hipOccupancyMaxActiveBlocksPerMultiprocessor's answer was not
// usable: with it the 32-channel block kernel ran 20x slower).  `slot` is the instantiation's own static std::atomic<int>
// (0 = not computed yet): threads racing on the first launch compute the same value, so no lock is needed.
#include <atomic>
int mval_cu_count();  // common.hip: compute units of the CURRENT device (hipDeviceProp_t::multiProcessorCount), cached per device
template <typename K>
static inline int p2_resident_wgs(K kernel, std::atomic<int>& slot, size_t smem, int waves_per_wg) {
  int nb = slot.load(std::memory_order_relaxed);
  if (nb) return nb;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncAttributes fa;
  nb = (int)((160 * 1024) / smem);
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) == hipSuccess && fa.numRegs > 0) {
    const int waves_simd = 512 / ((fa.numRegs + 7) / 8 * 8);
    nb = min(nb, max(1, waves_simd * 4 / waves_per_wg));
  } else {
    nb = min(nb, 2);
  }
  if (nb < 1) nb = 1;
  slot.store(nb, std::memory_order_relaxed);
  return nb;
}

struct P2Args {
  const _Float16* in;  // P2 planes of the input tensor
  const float* w;      // MVAL_PACK_MFMA16_H2 fragments (+ trailer)
  const float* w_unscale;
  const float* scale;
  const float* shift;
  const float* bound;  // [A, B]
  const _Float16* res1;
  const _Float16* res2;
  const unsigned* in_row;
  const unsigned* res1_row;
  const unsigned* res2_row;
  unsigned* out_row;
  _Float16* out;    // P2 planes ...
  float* out_f32;   // ... or fp32 NCHW (the heat-map layer)
  // ... or (training forward, round 4: EPI 3) the RAW conv output z = acc * 2^-s as fp32 NHWC, for train-mode BatchNorm; bn_part != nullptr:
  // every (workgroup, pixel-wave) also leaves the per-channel (sum, sum of squares) of what it stored over its whole tile walk,
  // float64 [cout][bn_slots][2] (bn_slots = wgs_x * WM, filled by the launcher and reported in *bn_slots_host)
  float* out_nhwc;
  int acc_nhwc;  // (EPI 3) != 0: out_nhwc += the result instead of = (a data gradient accumulating into the producer's gradient slot)
  double* bn_part;
  int64_t bn_part_cap;
  int* bn_slots_host;
  int bn_slots;
  unsigned long long* argmax_keys;  // heat-map layer: != nullptr = also keep the arg-max keys of every map, [N][MVAL_ARGMAX_SLOTS][Cout] (mval_common.h)
  int N, Hin, Win, Cin, Hout, Wout, Cout;  // Hout / Wout before the fused upsample
  int k, stride;
  int up, relu;
  // (round 5) k == 2: one PARITY of a transposed conv / of a stride-2 conv's data gradient (four 2 x 2 stride-1 convs over the input grid,
  // conv_mfma_split.hip's pack modes 3 / 4): the window of output pixel (y, x) covers input rows y - pad_y .. y - pad_y + 1 (pad_y = 1 - py;
  // columns alike), Hout x Wout = the INPUT grid, and with os = 1 the result goes to pixel (2 y + oy, 2 x + ox) of a (2 Hout) x (2 Wout)
  // tensor (P2 planes or the fp32 NHWC slot).  The four launches of a tensor share its rows: slot_base / slot_total give each its share
  // of the partial-maximum slots, keep_rows != 0 stops the launcher from zeroing rows an earlier parity has already written.
  int pad_y, pad_x, os, oy, ox, slot_base, slot_total, keep_rows;
  // in_sub = 1 (k == 1): the conv reads every second pixel of its input (a stride-2 1 x 1 conv as a stride-1 one over the sub-sampled view:
  // pose_resnet.py's downsample branches); Hin / Win stay the full input size, Hout / Wout = ceil(Hin / 2), ceil(Win / 2)
  int in_sub;
  int th, tw, tiles_x, tiles_y;
  int NS_total;
  int amax_tiles;
  unsigned tiles_img_magic, tiles_x_magic;  // 2^32 / d + 1: tile index -> (image, tile row) without a divide
  int tiles_total, wgs_x;  // persistent tile walk: tiles_x * tiles_y * N tiles over wgs_x workgroups per cout group
  unsigned long long* dbg;  // diagnostic builds (-DP2_STAMP): per-wave phase time stamps; nullptr otherwise
  // (round 6, training forward, EPI 3, 3x3 stride 1) in_z != nullptr: the input activation does not exist as planes -- it is
  // relu(BatchNorm(in_z)) of the producer's RAW conv output in_z (fp32 NHWC, N x Hin x Win x Cin) with the producer's batch statistics and
  // affine parameters, and the staging applies it while it copies: the same alpha = invstd * gamma, beta' = fma(-mean, alpha, beta),
  // r = fma(z, alpha, beta'), ReLU, scale 2^s from Samuelson's bound max_c(|gamma| sqrt(M - 1) + |beta|), split into (h, l) as the separate
  // apply pass (train_ops.hip bn_apply_fwd_p2_kernel) writes -- the LDS image is bit-identical to the one staged from its planes.
  const float* in_z;
  const float* zin_mean; const float* zin_invstd; const float* zin_gamma; const float* zin_beta;
  float zin_sqrt_m1;
  // (round 6, a data gradient, EPI 3) bs_z != nullptr: the tensor this launch writes is the gradient g of relu(BatchNorm(bs_z)) and this
  // launch is its ONLY writer -- the epilogue also leaves what the BatchNorm backward's reduction pass would compute from a second read of
  // it: per (workgroup, pixel wave) slot and channel the sums of m g and m g xhat (m = BatchNorm(z) > 0, xhat = (z - mean) invstd; float64
  // bs_part[slot][Cout][2]) and max |m g| (float bs_gmax[slot][Cout], behind the sums); *bs_slots_host = the number of slots (0: no room
  // in bs_cap doubles -- nothing kept).  bs_bound_slot: the dword the reduction pass zeroes for its finalize kernel.
  const float* bs_z;
  const unsigned char* bs_mask;  // != nullptr: m comes from the (out > 0) bits the producer's forward apply kept (one byte per float4: a ReLU behind
                                 // residual adds, train_ops.hip mask mode 3) and this launch may ACCUMULATE (acc_nhwc): it is the slot's LAST writer
  const float* bs_mean; const float* bs_invstd; const float* bs_gamma; const float* bs_beta;
  double* bs_part;
  float* bs_gmax;
  int64_t bs_cap;
  int* bs_slots_host;
  int bs_slots;
  unsigned* bs_bound_slot;
};

int mval_launch_conv_p2(const P2Args& a, hipStream_t s);  // conv_p2.hip; 1 = unsupported (dry != 0: no launch)
int mval_launch_nhwc_to_p2(const float* x, const unsigned* rows_in, _Float16* planes, unsigned* rows, int n_images, int HW, int C,
                            hipStream_t s);
int mval_conv_p2_supported(int k, int stride, int cin, int cout, int hin, int win, int up, int out_nchw, int n);
int mval_conv_p2_inz_supported(int cin, int cout, int h, int w, int n);  // the in_z form above (3x3 stride 1, raw fp32 NHWC out)
int mval_conv_p2_bsum_supported(int cin, int cout, int h, int w, int n);  // the bs_z form (3x3 stride-1 data gradient: cin = the conv's cout)
int mval_conv_p2_parity_supported(int cin, int cout, int h, int w, int n, int nhwc_out);  // one parity conv (k 2) over an h x w grid

// conv_block_p2.hip: a whole BasicBlock over P2 activations in one launch; 1 = unsupported
int mval_conv_block_p2_supported(int C, int N, int H, int W);
int mval_launch_conv_block_p2(int C, const void* in, void* out, const float* w1, const float* w1_unscale, const float* scale1,
                              const float* shift1, const float* bound1, const float* w2, const float* w2_unscale, const float* scale2,
                              const float* shift2, const float* bound2, const unsigned* in_row, unsigned* out_row, int N, int H, int W,
                              hipStream_t s);
// conv_bneck_p2.hip: a whole Bottleneck (cin -> 64 -> 64 -> 256, cin = 64 or 256) over P2 activations in one launch; the
// arrays hold the FLOAT OFFSETS into `params` of conv1 / conv2 / conv3's packed weights, their trailers, BN vectors and
// bounds; 1 = unsupported
int mval_conv_bneck_p2_supported(int cin, int planes, int N, int H, int W);
int mval_launch_conv_bneck_p2(int cin, const void* in, const void* res, void* out, const float* params, const int64_t* w, const int64_t* w_unscale,
                              const int64_t* scale, const int64_t* shift, const int64_t* bound, const unsigned* in_row,
                              const unsigned* res_row, unsigned* out_row, int N, int H, int W, hipStream_t s);
// conv_stem_p2.hip: HRNet's stem (3 -> 64 -> 64 channels, two 3x3 stride-2 convs) from the fp32 NCHW image to P2 planes in
// one launch (plus a small max |x| pass over the image into in_row); 1 = unsupported
int mval_conv_stem_p2_supported(int N, int H, int W);
int mval_launch_conv_stem_p2(const float* in, void* out, const float* w1, const float* scale1, const float* shift1, const float* bound1,
                             const float* w2, const float* w2_unscale, const float* scale2, const float* shift2, const float* bound2,
                             unsigned* in_row, unsigned* out_row, int N, int H, int W, hipStream_t s);
// conv_fuse_up_p2.hip: the two or three up-sampling 1x1 terms of a fuse-layer output (32 / 64 channels) added to the partial
// sum `res` in one launch; arrays per term (float offsets into `params`); 1 = unsupported
int mval_conv_fuse_up_p2_supported(int cout, int n_terms, const int* cin, const int* up, int N, int H, int W);
int mval_launch_conv_fuse_up_p2(int cout, int n_terms, int relu, const void* res, const unsigned* res_row, void* out, unsigned* out_row, const float* params,
                                const void* const* in, const unsigned* const* in_row, const int* cin, const int* up, const int64_t* w,
                                const int64_t* w_unscale, const int64_t* scale, const int64_t* shift, const int64_t* bound, int N, int H, int W,
                                hipStream_t s);
