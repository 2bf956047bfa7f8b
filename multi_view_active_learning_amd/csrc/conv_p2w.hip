// 3x3 stride-1 conv over P2 activations (conv_p2.h) on v_mfma_f32_32x32x16_f16: the same arithmetic as conv_p2.hip (three fp16
// MFMA products per fp32 product, fp32 accumulate) with twice the FLOPs per MFMA issue and per operand fragment.
//
// conv_p2.hip's kernels are bound by the SIMD's issue port, not by a pipe: a 16x16x32 MFMA holds the port for 8 of
// its 16 cycles and every sixteen-pixel fragment read from LDS feeds three of them.  With 32x32x16 a wave multiplies 32 output
// channels by 32 pixels per instruction: half the MFMA issues and two thirds of the LDS fragment reads per FLOP (row sharing
// kept), the weight stream per FLOP unchanged.  A fragment is a ROW of 32 pixels, so this form is for maps at least 32 wide
// (HRNet-W32: the 64-channel branch at 32x32); everything else stays on conv_p2.hip.
//
// Workgroup = 4 waves on a 4 x 32 output tile x 64 output channels: wave = 32 channels (wn) x 2 rows (wm); persistent over an
// XCD-contiguous tile range; K in 32-channel chunks, double-buffered LDS, register-staged 16-byte copies (as conv_p2.hip).
// Weights: the MVAL_PACK_MFMA16_H2 fragments, re-addressed -- the A operand of 32x32x16 wants lane (m = lane & 31, k octet
// lane >> 5): element (16-channel sub-tile m >> 4, octet 2 s + (lane >> 5), row m & 15) of the packed (tap, chunk) block.
// Output layout of 32x32: lane (n = lane & 31) holds channels 8 i + 4 (lane >> 5) + j of pixel n: lanes l / l + 32 own the halves
// of the 8-channel granule i -- the v_permlane32_swap store / residual path of conv_p2.hip works without permuting weight rows.
#include <stdlib.h>

#include "conv_p2.h"

#ifndef P2_VALU_PRIO
#define P2_VALU_PRIO 2
#endif

typedef p2_f32x4 f32x4;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef p2_f16x8 f16x8;
typedef p2_f16x4 f16x4;
typedef p2_u32x4 u32x4;
typedef p2_u32x2 u32x2;

__device__ __forceinline__ f32x16 pw_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int pw_fresh(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void conv_p2w_kernel(P2Args a) {
  constexpr int TH = 4, TW = 32, MS = 2;                 // tile; output rows per wave
  constexpr int PH = TH + 2, PW = TW + 2, PPX = 208;     // patch 6 x 34 = 204 slots per 8-channel block
  constexpr int plane_b = 4 * PPX * 16, buf_bytes = 8 * PPX * 16;
  constexpr int NE = (PH * 8 * PW + 255) / 256;          // staged granules per thread and chunk: 7
  constexpr int SB = 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* wgred = reinterpret_cast<unsigned*>(smem + 2 * buf_bytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int grp = (int)blockIdx.y * 2 + wn;  // 32-channel group of the output
  const int ngrp = a.Cout >> 5;
  const bool wave_active = grp < ngrp;

  // ---- tile walk (as conv_p2.hip) ------------------------------------------------------------------------------------
  const int X = a.wgs_x >= 8 ? 8 : 1;
  const int per = (a.tiles_total + X - 1) / X, wgx = a.wgs_x / X;
  const int xg = (int)blockIdx.x % X;
  int tile = xg * per + (int)blockIdx.x / X;
  const int tile_end = min(a.tiles_total, (xg + 1) * per);
  if (tile >= tile_end) return;
  const int tiles_img = a.tiles_x * a.tiles_y;
  int tn, toy, tox;
  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    n = a.tiles_img_magic ? (int)__umulhi((unsigned)t, a.tiles_img_magic) : t;
    const int r = t - n * tiles_img;
    const int tyi = a.tiles_x_magic ? (int)__umulhi((unsigned)r, a.tiles_x_magic) : r;
    oy0 = tyi * TH;
    ox0 = (r - tyi * a.tiles_x) * TW;
  };

  // ---- staging plan: granule e = tid + 256 i -> (patch row py, block sp = plane*4 + c8, column px) ----------------------
  const int C8 = a.Cin >> 3, nchunks = a.Cin >> 5;
  const unsigned hw16 = (unsigned)(a.Hin * a.Win) * 16u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, (unsigned)a.N * 2u * (unsigned)C8 * hw16, 0x00020000);
  unsigned lp[NE], goff[NE];  // LDS byte offset << 16 | sp << 12 | py << 7 | px (px = 127: none); global offset of chunk 0 (2^31: zero)
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int e = tid + 256 * i;
    const int r = e / PW, px = e - r * PW;
    const int sp = r & 7, py = r >> 3;
    lp[i] = py < PH ? ((unsigned)((sp * PPX + py * PW + px) * 16) << 16) | (sp << 12) | (py << 7) | px : 127u;
  }
  auto plan = [&](int n, int oy0, int ox0) {
    const unsigned nbase = (unsigned)n * 2u * (unsigned)C8 * hw16;
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const unsigned l = (unsigned)pw_fresh((int)lp[i]);
      const int px = l & 127, py = (l >> 7) & 31, sp = (l >> 12) & 7;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      const bool inb = px != 127 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      goff[i] = inb ? nbase + (unsigned)((sp >> 2) * C8 + (sp & 3)) * hw16 + (unsigned)(iy * a.Win + ix) * 16u : 0x80000000u;
    }
  };
  u32x4 stage[NE];
  auto load_stage = [&](int st) {
#pragma unroll
    for (int i = 0; i < NE; i++) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, __builtin_elementwise_add_sat(goff[i], (unsigned)st * 4u * hw16), 0, 0);
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NE; i++)
      if ((lp[i] & 127u) != 127u) *reinterpret_cast<u32x4*>(smem + buf * buf_bytes + (lp[i] >> 16)) = stage[i];
  };

  // ---- fragments ----------------------------------------------------------------------------------------------------------
  // x (B operand): lane -> pixel lane & 31 of the row fragment, k octet lane >> 5 of the 16-channel step
  const int xb = ((lane >> 5) * PPX + wm * MS * PW + (lane & 31)) * 16;
  // weights (A operand) out of the MVAL_PACK_MFMA16_H2 blocks [tap][chunk][16-channel sub-tile][plane][octet g][row r][16 B]
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7fffffff, 0x00020000);
  const int blk_bytes = (a.Cout >> 4) * 2048;
  const int m = lane & 31;
  const int wlane = ((min(grp, ngrp - 1) * 2 + (m >> 4)) * 128 + (lane >> 5) * 16 + (m & 15)) * 16;  // (+ s * 512: octet 2 s + (lane >> 5))
  auto wfrag = [&](int tap, int chunk, int s, int p) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(wr, wlane + s * 512 + p * 1024, (tap * nchunks + chunk) * blk_bytes, 0);
  };
  const float w_unscale = *a.w_unscale;
  const unsigned obytes = (unsigned)((int64_t)a.N * (a.Cout >> 3) * a.Hout * a.Wout * 32);
  const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res1), 0, a.res1 ? obytes : 0u, 0x00020000);
  const float bound_a = a.bound[0], bound_b = a.bound[1];
  const int C8o = a.Cout >> 3;
  const unsigned plane_bytes = (unsigned)C8o * (unsigned)(a.Hout * a.Wout) * 16u;

  f32x16 acc[MS];
  // a "column" = (16-channel step s, column tap kx): its three row taps x two planes; one column ahead
  u32x4 B[2][3][2];
  auto wload = [&](int par, int chunk, int col) {
    const int s = col / 3, kx = col % 3;
#pragma unroll
    for (int ky = 0; ky < 3; ky++) { B[par][ky][0] = wfrag(ky * 3 + kx, chunk, s, 0); B[par][ky][1] = wfrag(ky * 3 + kx, chunk, s, 1); }
  };
  // one chunk's MFMAs from LDS buffer `buf`; pre: request the first column of chunk `chunk_next` at the end
  auto mfma_stage = [&](int buf, int chunk, bool pre, int chunk_next) {
    const char* xs = smem + buf * buf_bytes + xb;
    constexpr int COLS = 6, QR = MS + 2, Q = COLS * QR;
    auto xoff = [&](int q) {
      const int col = q / QR, pr = q % QR, s = col / 3, kx = col % 3;
      return s * 2 * PPX * 16 + (pr * PW + kx) * 16;
    };
    u32x4 Xf[2][2];
    Xf[0][0] = *reinterpret_cast<const u32x4*>(xs + xoff(0));
    Xf[0][1] = *reinterpret_cast<const u32x4*>(xs + xoff(0) + plane_b);
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int col = q / QR, pr = q % QR;
      if (pr == 0) {
        if (col + 1 < COLS) wload((col + 1) & 1, chunk, col + 1);
        else if (pre) wload(0, chunk_next, 0);
      }
      if (q + 1 < Q) {
        Xf[(q + 1) & 1][0] = *reinterpret_cast<const u32x4*>(xs + xoff(q + 1));
        Xf[(q + 1) & 1][1] = *reinterpret_cast<const u32x4*>(xs + xoff(q + 1) + plane_b);
      }
      __builtin_amdgcn_sched_barrier(SB);
      const u32x4 xh = Xf[q & 1][0], xl = Xf[q & 1][1];
#pragma unroll
      for (int t3 = 0; t3 < 3; t3++) {
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const int ms = pr - ky;
          if (ms < 0 || ms >= MS) continue;
          const u32x4* wv = B[col & 1][ky];
          acc[ms] = t3 == 0 ? pw_mfma(wv[1], xh, acc[ms]) : t3 == 1 ? pw_mfma(wv[0], xl, acc[ms]) : pw_mfma(wv[0], xh, acc[ms]);
        }
      }
      __builtin_amdgcn_sched_barrier(SB);
    }
  };

  // ---- prologue -------------------------------------------------------------------------------------------------------------
  decode(tile, tn, toy, tox);
  plan(tn, toy, tox);
  load_stage(0);
  if (tid == 0) wgred[0] = wgred[1] = 0u;
  if (wave_active) wload(0, 0, 0);
  store_stage(0);
  __syncthreads();
  int buf = 0;

  for (;;) {
#pragma unroll
    for (int ms = 0; ms < MS; ms++)
#pragma unroll
      for (int k = 0; k < 16; k++) acc[ms][k] = 0.f;
    for (int st = 0; st + 1 < nchunks; st++) {
      load_stage(st + 1);
      if (wave_active) mfma_stage(buf, st, true, st + 1);
      store_stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    // ---- the tile's last chunk: the next tile's first chunk and the epilogue's operands are requested before its MFMAs ----
    const int n = tn, oy0 = toy, ox0 = tox;
    const int next_tile = tile + wgx;
    const bool have_next = next_tile < tile_end;
    if (have_next) {
      decode(next_tile, tn, toy, tox);
      plan(tn, toy, tox);
      load_stage(0);
    }
    P2RowRegs row_in, row_r1;
    u32x4 R1[MS][4];
    f32x4 scv[4], shv[4];  // BN vectors of the lane's channels, per 8-channel block
    const int L = pw_fresh(lane);
    const int xo = ox0 + (L & 31), yo = oy0 + wm * MS;
    // byte offset of the lane's granule of 8-channel block grp * 4 (row yo, its plane: h for lanes < 32)
    const unsigned vb = (wave_active && xo < a.Wout) ? (unsigned)n * 2u * plane_bytes + ((L >> 5) ? plane_bytes : 0u) + (unsigned)(((grp * 4) * a.Hout + yo) * a.Wout + xo) * 16u
                                                     : 0x80000000u;
    auto goffs = [&](int ms, int i) -> unsigned {  // (row, block): uniform offsets, saturating: rows below the image drop out
      return __builtin_elementwise_add_sat(vb, yo + ms < a.Hout ? (unsigned)((i * a.Hout + ms) * a.Wout) * 16u : 0x80000000u);
    };
    if (wave_active) {
      p2_row_request(a.in_row, n, row_in);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int c0 = grp * 32 + i * 8 + (L >> 5) * 4;
        scv[i] = *reinterpret_cast<const f32x4*>(a.scale + c0);
        shv[i] = *reinterpret_cast<const f32x4*>(a.shift + c0);
      }
      if (a.res1) {
        p2_row_request(a.res1_row, n, row_r1);
#pragma unroll
        for (int ms = 0; ms < MS; ms++)
#pragma unroll
          for (int i = 0; i < 4; i++) R1[ms][i] = __builtin_amdgcn_raw_buffer_load_b128(r1r, goffs(ms, i), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(SB);
      mfma_stage(buf, nchunks - 1, have_next, 0);
    }

    // ---- epilogue: lane = (pixel lane & 31, channels 8 i + 4 (lane >> 5) + j of the wave's 32) -----------------------------------
    __builtin_amdgcn_s_setprio(P2_VALU_PRIO);
    float amax = 0.f;
    if (wave_active) {
      const float in_inv = __uint_as_float(row_in.inv);
      float r1_inv = 0.f, out_mul, out_inv;
      float bound = bound_a * p2_row_amax(row_in) + bound_b;
      if (a.res1) {
        bound += p2_row_amax(row_r1);
        r1_inv = __uint_as_float(row_r1.inv);
      }
      p2_scale_of(bound, out_mul, out_inv);
      if (oy0 == 0 && ox0 == 0 && blockIdx.y == 0 && tid == 0) a.out_row[(int64_t)n * P2_ROW + P2_INV_SLOT] = __float_as_uint(out_inv);
      const float unscale = in_inv * w_unscale;
      const float floor_ = a.relu ? 0.f : -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const f32x4 sc = scv[i] * unscale, sh = shv[i];
#pragma unroll
        for (int ms = 0; ms < MS; ms++) {
          f32x4 r = (f32x4){acc[ms][4 * i], acc[ms][4 * i + 1], acc[ms][4 * i + 2], acc[ms][4 * i + 3]} * sc + sh;
          if (a.res1) {
            const u32x4 g = R1[ms][i];
            const auto s0 = __builtin_amdgcn_permlane32_swap(g.x, g.z, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(g.y, g.w, false, false);
            const u32x2 h = {s0[0], s1[0]}, l = {s0[1], s1[1]};
            r += p2_join(__builtin_bit_cast(f16x4, h), __builtin_bit_cast(f16x4, l)) * r1_inv;
          }
          r.x = p2_max_nan(r.x, floor_); r.y = p2_max_nan(r.y, floor_); r.z = p2_max_nan(r.z, floor_); r.w = p2_max_nan(r.w, floor_);
          if (yo + ms < a.Hout && xo < a.Wout) amax = conv_amax4(amax, r.x, r.y, r.z, r.w);
          f16x4 h, l;
          p2_split(r * out_mul, h, l);
          const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
          const auto s0 = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
          // (offsets in the vector offset: conv_p2.hip on the x4-store / SGPR-soffset hazard)
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){s0[0], s1[0], s0[1], s1[1]}, orr, goffs(ms, i), 0, 0);
          asm volatile("s_nop 1");
          __builtin_amdgcn_sched_barrier(SB);
        }
      }
    }
    {
      const unsigned amax_bits = p2_wave_umax(__float_as_uint(amax));
      if (lane == 0) {
        atomicMax(&wgred[0], amax_bits);
        if (atomicAdd(&wgred[1], 1u) == 3u) {
          const unsigned mm = atomicExch(&wgred[0], 0u);
          wgred[1] = 0u;
          const int timg = (oy0 / TH) * a.tiles_x + ox0 / TW;
          p2_slot_put(a.out_row + (int64_t)n * P2_ROW, timg * (int)gridDim.y + (int)blockIdx.y, tiles_img * (int)gridDim.y, mm);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (!have_next) break;
    store_stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    tile = next_tile;
  }
}

int mval_conv_p2w_supported(const P2Args& a) {
  if (a.k != 3 || a.stride != 1 || a.up || a.out_f32 || a.res2) return 0;
  if ((a.Cin & 31) || (a.Cout & 31) || a.Cout < 64 || a.Wout < 32 || a.Hout < 4) return 0;
  return 1;
}

int mval_launch_conv_p2w(const P2Args& a0, hipStream_t s) {
  if (!mval_conv_p2w_supported(a0)) return 1;
  P2Args a = a0;
  constexpr size_t smem = 2 * 8 * 208 * 16 + 16;
  a.tiles_x = (a.Wout + 31) / 32;
  a.tiles_y = (a.Hout + 3) / 4;
  a.amax_tiles = a.tiles_x * a.tiles_y;
  a.tiles_total = a.amax_tiles * a.N;
  a.tiles_img_magic = a.amax_tiles > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)a.amax_tiles + 1) : 0u;
  a.tiles_x_magic = a.tiles_x > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)a.tiles_x + 1) : 0u;
  const unsigned groups = (unsigned)((a.Cout + 63) / 64);
  static std::atomic<int> occ{0};
  int per_cu = p2_resident_wgs(&conv_p2w_kernel, occ, smem, 4);
  const char* pe = getenv("MVAL_P2_WGS");
  if (pe && atoi(pe) > 0) per_cu = atoi(pe);
  int wgs = (mval_cu_count() * per_cu / (int)groups) & ~7;
  if (wgs < 8) wgs = 8;
  if (wgs >= a.tiles_total) wgs = a.tiles_total;
  else {
    const int per = (a.tiles_total + 7) / 8, rounds = (per + wgs / 8 - 1) / (wgs / 8);
    wgs = 8 * ((per + rounds - 1) / rounds);
  }
  a.wgs_x = wgs;
  if ((int64_t)a.amax_tiles * groups > P2_SLOTS) mval_launch_zero_rows(a.out_row, (int64_t)a.N * P2_ROW, s);
  hipLaunchKernelGGL(conv_p2w_kernel, dim3((unsigned)wgs, groups), dim3(256), smem, s, a);
  return 0;
}
