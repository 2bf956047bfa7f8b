// Heat-map uncertainty statistics (K13): HP / MPE / BSB (reference strategy.py:1149-1215).
//
// HBM-bound by design: one workgroup per (frame, view, joint) map stages the map once into
// LDS (coalesced float loads, row stride padded by one word so that "one thread per row"
// reads are bank-conflict free) and derives everything from that copy; the only HBM traffic
// is hh*wh*4 bytes in and 8 bytes out per map.
//
//   HP   1 - max(softmax(map, dim=1))  == 1 - max_r 1 / sum_c exp(x_rc - max_c x_rc)
//   MPE  peaks = peak_local_max(map, min_distance=2)   (scikit-image 0.18/0.19 semantics)
//        p = exp(peaks) / sum(exp(peaks)) ; H = sum -p * log(p)   -- float32, python left-to-right
//   BSB  q = softmax(map, dim=1) ; |q[peak0] - q[peak1]| of its two highest local peaks
#include "mval_common.h"

#define SC_THREADS 256
#define SC_MAX_PEAKS 512

struct ScoreSmem {
  float red[SC_THREADS];
  int n_cand;
  int overflow;
  int adjacent;
  float vmin;
  float ssum;
};

__device__ __forceinline__ float block_reduce_min(float v, float* red) {
  v = wave_min(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return r;
}

// (value desc, flat index asc) ordering for the candidate sort
__device__ __forceinline__ bool cand_before(float va, int ia, float vb, int ib) {
  return (va > vb) || (va == vb && ia < ib);
}

// order of torch.argmax: NaN is the maximum; ties -> lowest flat index (as csrc/decode.hip)
__device__ __forceinline__ bool sc_better(float v, int i, float bv, int bi) {
  const bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (vn) return i < bi;
  return (v > bv) || (v == bv && i < bi);
}

// python's sum() over a list of float32: strictly left to right.  Eight values per pair of 16-byte LDS reads (the one-at-a-time loop
// paid the LDS latency per element: ~160 candidates x 2 sums per map on noise maps)
__device__ __forceinline__ float serial_sum_lds(const float* a, int n) {
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const float4 u = *reinterpret_cast<const float4*>(a + i), w = *reinterpret_cast<const float4*>(a + i + 4);
    s += u.x; s += u.y; s += u.z; s += u.w;
    s += w.x; s += w.y; s += w.z; s += w.w;
  }
  for (; i < n; i++) s += a[i];
  return s;
}

// DECODE: the same staged copy also yields the hard arg-max key-point of the map (utils/evaluation.py:13-30, what
// mval_argmax_decode computes from a second read of the heat-maps): one pass over each map per scoring pass.
struct ScoreDecodeArgs {
  const uint8_t* valid;  // [B,J] or null
  int64_t* kp2d;         // [n_maps,2] (x, y)
  int V, J, stride, split_width;
};

template <int KIND, bool DECODE>
__global__ __launch_bounds__(SC_THREADS) void score_maps_kernel(const float* __restrict__ hm, float* __restrict__ stat,
                                                                int32_t* __restrict__ n_peaks, int hh, int wh,
                                                                ScoreDecodeArgs d, int cap, int rescue) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ld = wh + 1;
  float* tile = reinterpret_cast<float*>(smem_raw);                      // hh * ld
  // (the HP kernel has no candidate list: its workgroups need only the tile, so twice as many fit a CU)
  float* cval = tile + (((hh * ld) + 3) & ~3);                           // cap
  int* cidx = reinterpret_cast<int*>(cval + (KIND == MVAL_SCORE_HP ? 0 : cap));  // cap
  ScoreSmem* sm = reinterpret_cast<ScoreSmem*>(cidx + (KIND == MVAL_SCORE_HP ? 0 : cap));
  const int tid = threadIdx.x;
  const int64_t map = blockIdx.x;
  // second pass (candidate list as large as the map's interior, one workgroup per CU): only the maps whose list
  // overflowed SC_MAX_PEAKS in the first pass -- wide plateaus: flat backgrounds -- are redone
  if (rescue && n_peaks[map] != -1) return;
  const float* p = hm + map * (int64_t)hh * wh;
  const int npix = hh * wh;

  float bv = -INFINITY;
  int bi = 0x7fffffff;
  float vmin = INFINITY;  // min of the map the peaks are searched in (MPE: the heat-map, here; BSB: its row softmax, below)
  if ((npix & 3) == 0 && (wh & 3) == 0) {  // float4 loads (maps are 16-byte aligned then); a quad never straddles rows
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = tid; i < (npix >> 2); i += SC_THREADS) {
      const float4 q = p4[i];
      const int base = i << 2;
      const int y = base / wh, x = base - y * wh;
      float* t = tile + y * ld + x;
      t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
      if (KIND == MVAL_SCORE_MPE) vmin = fminf(fminf(vmin, fminf(q.x, q.y)), fminf(q.z, q.w));
      if (DECODE) {
        // a thread meets its pixels in increasing index order, so "strictly greater, or the first NaN" is the whole
        // torch.argmax order here (the cross-lane reduction below uses the full comparison)
        if (q.x > bv || (q.x != q.x && bv == bv)) { bv = q.x; bi = base; }
        if (q.y > bv || (q.y != q.y && bv == bv)) { bv = q.y; bi = base + 1; }
        if (q.z > bv || (q.z != q.z && bv == bv)) { bv = q.z; bi = base + 2; }
        if (q.w > bv || (q.w != q.w && bv == bv)) { bv = q.w; bi = base + 3; }
      }
    }
  } else {
    for (int i = tid; i < npix; i += SC_THREADS) {
      const int y = i / wh, x = i - y * wh;
      const float q = p[i];
      tile[y * ld + x] = q;
      if (KIND == MVAL_SCORE_MPE) vmin = fminf(vmin, q);
      if (DECODE && (q > bv || (q != q && bv == bv))) { bv = q; bi = i; }
    }
  }
  if (tid == 0) { sm->n_cand = 0; sm->overflow = 0; sm->adjacent = 0; }
  if (DECODE) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (sc_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    // (slots 16.. of red[]: the reductions further down use 0..3, so no barrier is needed between the two)
    if ((tid & 63) == 0) { sm->red[16 + (tid >> 6)] = bv; sm->red[20 + (tid >> 6)] = __int_as_float(bi); }
  }
  __syncthreads();
  if (DECODE && tid == 0) {
    for (int w = 1; w < SC_THREADS / 64; w++) {
      const float ov = sm->red[16 + w];
      const int oi = __float_as_int(sm->red[20 + w]);
      if (sc_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 0;  // all -inf: first element
    const int j = (int)(map % d.J);
    const int64_t b = map / ((int64_t)d.V * d.J);
    const bool ok = !d.valid || d.valid[b * d.J + j];
    d.kp2d[map * 2] = ok ? (int64_t)(bi % d.split_width) * d.stride : 0;
    d.kp2d[map * 2 + 1] = ok ? (int64_t)(bi / d.split_width) * d.stride : 0;
  }

  // ---- row-wise softmax statistics (HP, BSB): T lanes per row ---------------------------------
  if (KIND == MVAL_SCORE_HP || KIND == MVAL_SCORE_BSB) {
    int T = 1;
    while (T < 16 && T * 2 * hh <= SC_THREADS) T <<= 1;  // 4 lanes per row on 64-row maps, 2 on 96-row maps
    const int sub = tid & (T - 1);
    float best = 0.f;
    const int rows_per_pass = SC_THREADS / T;
    for (int r0 = 0; r0 < hh; r0 += rows_per_pass) {  // every thread makes every pass: the shuffles need whole lane groups
      const int r = r0 + tid / T;
      const bool live = r < hh;
      float* row = tile + (live ? r : 0) * ld;
      // lane `sub` owns the contiguous column segment [c0, c1): with the odd row stride (wh + 1) the 64 lanes of a
      // wave (16 rows x 4 segments on 64-wide maps) then hit 64 different banks (interleaved columns: 2-4-way conflicts)
      const int seg = (wh + T - 1) / T, c0 = sub * seg, c1 = min(wh, c0 + seg);
      float m = -INFINITY;
      for (int c = c0; c < c1; c++) m = fmaxf(m, row[c]);
      for (int o = T >> 1; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      // the reference sums exp(x - m) over a row left to right in float32 (torch softmax on CPU vectorises, so the
      // last bits differ anyway: tests allow 3e-6 relative); partial sums over the lanes' segments here
      float s = 0.f;
      for (int c = c0; c < c1; c++) s += expf(row[c] - m);
      for (int o = T >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (KIND == MVAL_SCORE_BSB && live) {
        for (int c = c0; c < c1; c++) {
          const float q = expf(row[c] - m) / s;
          row[c] = q;
          vmin = fminf(vmin, q);
        }
      }
      if (live) best = fmaxf(best, 1.0f / s);
    }
    if (KIND == MVAL_SCORE_HP) {
      best = block_reduce_max(best, sm->red);
      if (tid == 0) { stat[map] = 1.0f - best; n_peaks[map] = 0; }
      return;
    }
    __syncthreads();
  }

  // ---- peak_local_max(min_distance=2) -------------------------------------------------
  vmin = block_reduce_min(vmin, sm->red);
  const int ih = hh - 4, iw = wh - 4;  // interior after the 2-px border exclusion
  auto put = [&](float v, int idx) {
    int slot = atomicAdd(&sm->n_cand, 1);
    if (slot < cap) {
      cval[slot] = v;
      cidx[slot] = idx;
    } else {
      sm->overflow = 1;
    }
  };
  if (ih > 0 && iw > 0 && iw <= SC_THREADS) {
    // a pixel is a candidate iff no value of its 5x5 window exceeds it (and none is NaN): the window maximum as the maximum of five
    // ROW maxima -- a thread walks one interior column of its band of rows with the last five row maxima in registers: 5 + 1 LDS reads
    // per pixel instead of 25 (consecutive lanes = consecutive columns: conflict-free with the odd row stride).  The maximum
    // propagates NaN, so "m <= v" is exactly "every neighbour <= v".
    const int G = SC_THREADS / iw, g = tid / iw, x = tid - g * iw + 2;
    const int rows = (ih + G - 1) / G, ya = 2 + g * rows, yb = min(2 + ih, ya + rows);
    if (g < G && ya < yb) {
      auto rowmax = [&](int y) {
        const float* t = tile + y * ld + x;
        return __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_elementwise_maximum(t[-2], t[-1]), __builtin_elementwise_maximum(t[0], t[1])), t[2]);
      };
      float r0 = rowmax(ya - 2), r1 = rowmax(ya - 1), r2 = rowmax(ya), r3 = rowmax(ya + 1);
      for (int y = ya; y < yb; y++) {
        const float r4 = rowmax(y + 2);
        const float m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_elementwise_maximum(r0, r1), __builtin_elementwise_maximum(r2, r3)), r4);
        const float v = tile[y * ld + x];
        if (v > vmin && m <= v) put(v, y * wh + x);
        r0 = r1; r1 = r2; r2 = r3; r3 = r4;
      }
    }
  } else if (ih > 0 && iw > 0) {
    for (int i = tid; i < ih * iw; i += SC_THREADS) {
      int y = i / iw + 2, x = i % iw + 2;
      float v = tile[y * ld + x];
      if (!(v > vmin)) continue;
      bool is_max = true;
#pragma unroll
      for (int dy = -2; dy <= 2; dy++)
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) is_max = is_max && (tile[(y + dy) * ld + x + dx] <= v);
      if (is_max) put(v, y * wh + x);
    }
  }
  __syncthreads();
  if (sm->overflow) {
    if (tid == 0) { stat[map] = NAN; n_peaks[map] = -1; }
    return;
  }
  int n = sm->n_cand;
  if (n <= SC_THREADS) {
    // short lists (what network outputs give: a handful to ~200 candidates): RANK sort -- thread i counts the candidates that come
    // before its own (broadcast reads of four at a time, no barrier inside) and stores it at that position; the same scan sees
    // whether any two candidates share a value (the only case the spacing pass below has work to do)
    const int n4 = (n + 3) & ~3;
    if (tid >= n && tid < n4) { cval[tid] = -INFINITY; cidx[tid] = 0x7fffffff; }
    __syncthreads();
    const float v = tid < n ? cval[tid] : 0.f;
    const int ix = tid < n ? cidx[tid] : 0;
    int rank = 0, same = 0;  // candidates with a larger value; with the same value (the thread's own among them)
    for (int j = 0; j < n4; j += 4) {
      const float4 vv = *reinterpret_cast<const float4*>(cval + j);
      rank += (int)(vv.x > v) + (int)(vv.y > v) + (int)(vv.z > v) + (int)(vv.w > v);
      same += (int)(vv.x == v) + (int)(vv.y == v) + (int)(vv.z == v) + (int)(vv.w == v);
    }
    const bool tie = tid < n && same > 1;
    if (tie)  // equal values: row-major order among them (rare: the scan is repeated with the indices)
      for (int j = 0; j < n; j++) rank += (int)(cval[j] == v && cidx[j] < ix);
    __syncthreads();
    if (tid < n) {
      cval[rank] = v;
      cidx[rank] = ix;
      if (tie) sm->overflow = 1;  // (0 here: an overflowed list has returned above)
    }
    __syncthreads();
  } else {
  // bitonic sort of the candidate list (padded with -inf / INT_MAX)
  int npow = 1;
  while (npow < n) npow <<= 1;
  for (int i = n + tid; i < npow; i += SC_THREADS) { cval[i] = -INFINITY; cidx[i] = 0x7fffffff; }
  __syncthreads();
  for (int k = 2; k <= npow; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow; i += SC_THREADS) {
        int l = i ^ j;
        if (l > i) {
          float va = cval[i], vb = cval[l];
          int ia = cidx[i], ib = cidx[l];
          bool up = (i & k) == 0;  // ascending position == "before" order
          bool swap = up ? cand_before(vb, ib, va, ia) : cand_before(va, ia, vb, ib);
          if (swap) { cval[i] = vb; cval[l] = va; cidx[i] = ib; cidx[l] = ia; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i + 1 < n; i += SC_THREADS)
    if (cval[i] == cval[i + 1]) sm->overflow = 1;  // (0 here: an overflowed list has returned above)
  __syncthreads();
  }
  // ensure_spacing: in sorted order a kept peak rejects later peaks at Chebyshev distance < 2.  Two 5x5 maxima can only be that
  // close when their values are equal (plateaus), and among equal values the order is row-major: a candidate is rejected iff one
  // of its four row-major-earlier neighbours (up-left, up, up-right, left) was kept.
  //   * No two equal values among the candidates (what network outputs look like away from plateaus): nothing can be rejected.
  //   * Equal values far apart (two of ~150 noise maxima sharing a float: a few maps per thousand) reject nothing either: only
  //     candidates with an ADJACENT pixel of their own value -- looked up in the staged map, in parallel -- take part ("flagged":
  //     the sign bit of their index).
  //   * Flagged candidates exist: the staged map is not needed any more (the values live in cval), its memory becomes the "kept"
  //     map, and ONE WAVE walks the sorted list 64 candidates at a time -- the flagged ones of a group decide one after the other
  //     (a ballot names them; everybody else is kept without a look), the group is compacted with a prefix count.  O(flagged)
  //     serial steps whatever the plateaus -- a final layer whose inputs are all zero over the background writes its bias there,
  //     thousands of equal candidates per map.  (Round 4: the serial pass over ALL candidates by one thread cost 37 us for a
  //     single map of a noise batch.)
  bool ties = sm->overflow != 0;
  if (ties) {
    for (int i = tid; i < n; i += SC_THREADS) {
      const float v = cval[i];
      const int idx = cidx[i];
      const int yi = idx / wh, xi = idx - yi * wh;
      const float* t = tile + yi * ld + xi;  // (candidates are interior pixels: the neighbours exist)
      if (t[-ld - 1] == v || t[-ld] == v || t[-ld + 1] == v || t[-1] == v || t[1] == v || t[ld - 1] == v || t[ld] == v || t[ld + 1] == v) {
        cidx[i] = idx | (int)0x80000000;
        sm->adjacent = 1;
      }
    }
    __syncthreads();
    ties = sm->adjacent != 0;
  }
  if (ties)
    for (int i = tid; i < hh * ld; i += SC_THREADS) tile[i] = 0.f;
  __syncthreads();
  if (ties && tid < 64) {
    const int lane = tid;
    int kept = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const bool in = i < n;
      const int raw = in ? cidx[i] : 0;
      const float v = in ? cval[i] : 0.f;
      const int idx = raw & 0x7fffffff;
      const int yi = idx / wh, xi = idx - yi * wh;
      volatile float* t = tile + yi * ld + xi;  // (volatile: another LANE's store decides what this lane reads)
      bool rejected = false;
      unsigned long long fm = __ballot(in && raw < 0);
      while (fm) {
        const int l = __builtin_ctzll(fm);
        fm &= fm - 1;
        if (lane == l) {
          rejected = !(t[-ld - 1] == 0.f && t[-ld] == 0.f && t[-ld + 1] == 0.f && t[-1] == 0.f);
          if (!rejected) *t = 1.f;
        }
        __builtin_amdgcn_wave_barrier();  // (the next candidate's reads follow this one's write in program order)
      }
      const unsigned long long km = __ballot(in && !rejected);
      const int pos = kept + __popcll(km & ((1ull << lane) - 1ull));
      if (in && !rejected) {  // (pos <= i, and the group's values are in registers: in place)
        cval[pos] = v;
        cidx[pos] = idx;
      }
      kept += __popcll(km);
    }
    if (lane == 0) sm->n_cand = kept;
  }
  __syncthreads();
  n = sm->n_cand;  // (unchanged without ties)

  if (KIND == MVAL_SCORE_BSB) {
    if (tid == 0) {
      n_peaks[map] = n;
      stat[map] = (n >= 2) ? fabsf(cval[0] - cval[1]) : NAN;  // reference: IndexError when < 2 peaks
    }
    return;
  }
  // ---- MPE entropy (strategy.py:1171-1175) --------------------------------------------
  for (int i = tid; i < n; i += SC_THREADS) cval[i] = expf(cval[i]);
  __syncthreads();
  if (tid == 0) sm->ssum = serial_sum_lds(cval, n);  // python sum(): left to right in float32
  __syncthreads();
  const float s = sm->ssum;
  for (int i = tid; i < n; i += SC_THREADS) {
    float pr = cval[i] / s;
    // -prob * math.log(prob): log in double, weak python float -> float32, float32 product
    cval[i] = (-pr) * (float)log((double)pr);
  }
  __syncthreads();
  if (tid == 0) {
    stat[map] = serial_sum_lds(cval, n);  // no peaks -> python int 0
    n_peaks[map] = n;
  }
}

template <bool DECODE>
static int launch_score(int kind, const float* heatmaps, float* stat, int32_t* n_peaks, int64_t n_maps, int hh, int wh,
                        const ScoreDecodeArgs& d, hipStream_t s) {
  const size_t tile_b = (size_t)((hh * (wh + 1) + 3) & ~3) * 4, fixed_b = sizeof(ScoreSmem) + 16;
  dim3 grid((unsigned)n_maps), block(SC_THREADS);
  // the candidate list never needs more than the interior (hh-4)(wh-4), rounded up to the sort's power of two
  int full = 4;  // (at least one 16-byte group: the rank sort reads the list four candidates at a time)
  while (full < (hh - 4) * (wh - 4)) full <<= 1;
  for (int pass = 0; pass < 2; pass++) {
    const int cap = pass == 0 ? (full < SC_MAX_PEAKS ? full : SC_MAX_PEAKS) : full;
    if (pass == 1 && (kind == MVAL_SCORE_HP || full <= SC_MAX_PEAKS)) break;
    const size_t smem = tile_b + (kind == MVAL_SCORE_HP ? 0 : (size_t)cap * 8) + fixed_b;
    if (smem > 160 * 1024) return pass == 0 ? 1 : 0;  // (no room for the rescue: overflowed maps keep NaN / -1)
    if (smem > 64 * 1024) {
      const void* fn = kind == MVAL_SCORE_HP    ? (const void*)score_maps_kernel<MVAL_SCORE_HP, DECODE>
                       : kind == MVAL_SCORE_MPE ? (const void*)score_maps_kernel<MVAL_SCORE_MPE, DECODE>
                                                : (const void*)score_maps_kernel<MVAL_SCORE_BSB, DECODE>;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return pass == 0 ? 1 : 0;
    }
    if (kind == MVAL_SCORE_HP)
      hipLaunchKernelGGL((score_maps_kernel<MVAL_SCORE_HP, DECODE>), grid, block, smem, s, heatmaps, stat, n_peaks, hh, wh, d, cap, pass);
    else if (kind == MVAL_SCORE_MPE)
      hipLaunchKernelGGL((score_maps_kernel<MVAL_SCORE_MPE, DECODE>), grid, block, smem, s, heatmaps, stat, n_peaks, hh, wh, d, cap, pass);
    else
      hipLaunchKernelGGL((score_maps_kernel<MVAL_SCORE_BSB, DECODE>), grid, block, smem, s, heatmaps, stat, n_peaks, hh, wh, d, cap, pass);
  }
  return 0;
}

extern "C" int mval_score_maps(int kind, const float* heatmaps, float* stat, int32_t* n_peaks, int64_t n_maps, int hh,
                               int wh, void* stream) {
  MVAL_REQUIRE(n_maps >= 0 && hh > 0 && wh > 0, "mval_score_maps: bad dims");
  MVAL_REQUIRE(kind >= 0 && kind <= 2, "mval_score_maps: unknown kind %d", kind);
  if (n_maps == 0) return 0;
  ScoreDecodeArgs d = {};
  MVAL_REQUIRE(launch_score<false>(kind, heatmaps, stat, n_peaks, n_maps, hh, wh, d, mval_stream(stream)) == 0,
               "mval_score_maps: heat-map %dx%d does not fit LDS", hh, wh);
  MVAL_CHECK_LAUNCH("mval_score_maps");
  return 0;
}

extern "C" int mval_score_decode_maps(int kind, const float* heatmaps, const uint8_t* valid, float* stat, int32_t* n_peaks,
                                      int64_t* kp2d, int B, int V, int J, int hh, int wh, int stride, int split_width,
                                      void* stream) {
  MVAL_REQUIRE(B >= 0 && V > 0 && J > 0 && hh > 0 && wh > 0 && split_width > 0 && kp2d, "mval_score_decode_maps: bad arguments");
  MVAL_REQUIRE(kind >= 0 && kind <= 2, "mval_score_decode_maps: unknown kind %d", kind);
  const int64_t n_maps = (int64_t)B * V * J;
  if (n_maps == 0) return 0;
  ScoreDecodeArgs d = {valid, kp2d, V, J, stride, split_width};
  MVAL_REQUIRE(launch_score<true>(kind, heatmaps, stat, n_peaks, n_maps, hh, wh, d, mval_stream(stream)) == 0,
               "mval_score_decode_maps: heat-map %dx%d does not fit LDS", hh, wh);
  MVAL_CHECK_LAUNCH("mval_score_decode_maps");
  return 0;
}

// ---- AVG / STD over the valid maps of a frame, in python / numpy order -----------------
template <typename T>
__device__ T np_pairwise_sum_t(const T* a, int n) {
  if (n < 8) {
    T r = 0;
    for (int i = 0; i < n; i++) r += a[i];
    return r;
  }
  if (n <= 128) {
    T r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; k++) r[k] += a[i + k];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum_t<T>(a, n2) + np_pairwise_sum_t<T>(a + n2, n - n2);
}

#define SR_MAX 1024

__global__ void score_reduce_kernel(const float* __restrict__ per_map, const uint8_t* __restrict__ valid,
                                    double* __restrict__ out, int B, int V, int J, int mode) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* pm = per_map + (int64_t)b * V * J;
  int n = 0;
  for (int j = 0; j < J; j++) n += (!valid || valid[(int64_t)b * J + j]) ? 1 : 0;
  n *= V;
  if (n == 0) { out[b] = NAN; return; }  // reference: ZeroDivisionError / nan
  if (mode == MVAL_REDUCE_AVG_F64) {  // sum(python floats) / len
    double s = 0.0;
    for (int v = 0; v < V; v++)
      for (int j = 0; j < J; j++)
        if (!valid || valid[(int64_t)b * J + j]) s += (double)pm[v * J + j];
    out[b] = s / (double)n;
  } else if (mode == MVAL_REDUCE_AVG_F32) {  // sum(np.float32) / len  (float32 throughout)
    float s = 0.f;
    for (int v = 0; v < V; v++)
      for (int j = 0; j < J; j++)
        if (!valid || valid[(int64_t)b * J + j]) s += pm[v * J + j];
    out[b] = (double)(s / (float)n);
  } else if (mode == MVAL_REDUCE_STD_F64) {  // np.std(float64 array)
    double buf[SR_MAX];
    int k = 0;
    for (int v = 0; v < V; v++)
      for (int j = 0; j < J; j++)
        if (!valid || valid[(int64_t)b * J + j]) buf[k++] = (double)pm[v * J + j];
    double mean = np_pairwise_sum_t<double>(buf, n) / (double)n;
    for (int i = 0; i < n; i++) { double d = buf[i] - mean; buf[i] = d * d; }
    out[b] = sqrt(np_pairwise_sum_t<double>(buf, n) / (double)n);
  } else {  // np.std(float32 array): float32 intermediates
    float buf[SR_MAX];
    int k = 0;
    for (int v = 0; v < V; v++)
      for (int j = 0; j < J; j++)
        if (!valid || valid[(int64_t)b * J + j]) buf[k++] = pm[v * J + j];
    float mean = (float)((double)np_pairwise_sum_t<float>(buf, n) / (double)n);
    for (int i = 0; i < n; i++) { float d = buf[i] - mean; buf[i] = d * d; }
    out[b] = (double)sqrtf(np_pairwise_sum_t<float>(buf, n) / (float)n);
  }
}

extern "C" int mval_score_reduce(const float* per_map, const uint8_t* valid, double* out, int B, int V, int J, int mode,
                                 void* stream) {
  MVAL_REQUIRE(B >= 0 && V > 0 && J > 0 && V * J <= SR_MAX, "mval_score_reduce: bad dims (V*J must be <= %d)", SR_MAX);
  MVAL_REQUIRE(mode >= 0 && mode <= 3, "mval_score_reduce: unknown mode %d", mode);
  if (B == 0) return 0;
  hipLaunchKernelGGL(score_reduce_kernel, dim3((B + 63) / 64), dim3(64), 0, mval_stream(stream), per_map, valid, out, B,
                     V, J, mode);
  MVAL_CHECK_LAUNCH("mval_score_reduce");
  return 0;
}
