// Micro-benchmark (round 5, the review's item 4): the MFMA loop of the dominant inference kernel
//   conv_p2_kernel<3, 1, 1, 4, 1, 1, 4, 16, true, 0>   (3x3 stride-1 conv over P2 planes, 64-pixel row-sharing tiles, 4 cout waves)
// as a SKELETON -- the same LDS image of the input patch, the same fragment reads, MFMA order, weight stream and barrier structure on
// synthetic operands, no epilogue -- in the form the product runs (MODE 0: every wave pulls its own weight fragments from L2, two
// 4-wave workgroups per CU) and in the form the review asked for (MODE 1: ONE 8-wave workgroup per CU = 4 cout waves x 2 pixel waves,
// the stage's weight fragments staged ONCE per workgroup into LDS and read from there by both pixel waves), plus ablations that switch
// single streams off (x fragment reads, weight stream, staging) so that each one's cost in the loop can be read off.
// 256 -> 256 channels: 8 stages (32-channel chunks) of 108 MFMAs per wave and stage; weights 2.36 MB (fragment order, L2 / MALL resident).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/p2_loop.hip -o tools/micro/bin/p2_loop        run: tools/micro/bin/p2_loop [stages=512]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

constexpr int MS = 4, NCHUNK = 8, NS_TOTAL = 16, PW = 18;
// ablation bits
constexpr int NO_X = 1;      // x fragments read once per stage (no LDS fragment stream)
constexpr int NO_W = 2;      // weight fragments fetched once per kernel (no weight stream)
constexpr int NO_STAGE = 4;  // no global -> LDS staging of the next patch (the barrier stays)
constexpr int NO_MFMA = 8;   // everything but the MFMAs
// round 6: issue-order / prefetch-depth variants of the product form (MODE 0).  Loads return IN ORDER (one vmcnt counter): a wait for a weight
// fragment also waits for every load issued before it -- in the product's order that is the next stage's patch granules, straight from HBM / MALL.
constexpr int OPT_WFIRST = 1;  // request the next column's weights BEFORE the next stage's patch granules (the column-1 wait then leaves them in flight)
constexpr int OPT_X2 = 2;      // x fragments two steps ahead (three register pairs) instead of one
constexpr int OPT_W2 = 4;      // weight fragments two columns ahead (three register sets) instead of one
constexpr int OPT_PRIO = 8;    // s_setprio 1 around each step's MFMA group
constexpr int OPT_LATE = 16;   // patch granules requested at the start of column 1 (behind column 2's weights) instead of the stage's start

template <int MODE, int ABL, int WMX = 2, int OPT = 0>
__global__ __launch_bounds__(MODE ? 256 * WMX : 256) __attribute__((amdgpu_waves_per_eu(MODE ? WMX : 2, MODE ? WMX : 2))) void loop_kernel(const u32x4* __restrict__ W,
                                                                                                         const u32x4* __restrict__ X,
                                                                                                         float* __restrict__ out, int stages,
                                                                                                         unsigned xmask) {
  constexpr int WM = MODE ? WMX : 1, NTH = 256 * WM;
  constexpr int PH = 4 * WM + 2, slots = PH * PW, PPX = (slots + 15) & ~15;
  constexpr int buf_bytes = 8 * PPX * 16, plane_b = 4 * PPX * 16;
  constexpr int NE = (8 * PPX + NTH - 1) / NTH;
  constexpr int WL = 9 * 8192;  // MODE 1: the stage's weight fragments of the workgroup's 64 couts in LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem + 2 * buf_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave & 3, wm = wave >> 2;
  const int by = blockIdx.y;  // cout group (4 sub-tiles of 16)
  const int xb = ((lane >> 4) * PPX + wm * MS * PW + (lane & 15)) * 16;
  const int wq = ((by * 4 + wn) * 128 + lane);  // u32x4 index of the lane's fragment inside a block (plane 0)
  const int blkq = NS_TOTAL * 128;             // u32x4 per (tap, chunk) block
  f32x4 acc[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ms++) acc[ms] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr bool W2 = (OPT & OPT_W2) && !MODE, X2 = (OPT & OPT_X2) != 0;
  u32x4 B[W2 ? 3 : 2][3][2];
  u32x4 stage[NE], wst[MODE ? 9 : 1];
  const unsigned g0 = (unsigned)((blockIdx.y * gridDim.x + blockIdx.x) * 8 * PPX);
  auto load_stage = [&](int s) {
    if (ABL & NO_STAGE) return;
#pragma unroll
    for (int i = 0; i < NE; i++) stage[i] = X[(g0 + (unsigned)s * 7919u * 64u + (unsigned)(tid + NTH * i)) & xmask];
    if (MODE && !(ABL & NO_W)) {
#pragma unroll
      for (int t = 0; t < 9; t++) wst[t] = W[(t * NCHUNK + (s & 7)) * blkq + by * 512 + (tid & 511)];
    }
  };
  auto store_x = [&](int buf) {
    if (ABL & NO_STAGE) return;
#pragma unroll
    for (int i = 0; i < NE; i++)
      if (tid + NTH * i < 8 * PPX) *reinterpret_cast<u32x4*>(smem + buf * buf_bytes + (tid + NTH * i) * 16) = stage[i];
  };
  auto store_w = [&]() {
    if (MODE && !(ABL & NO_W)) {
#pragma unroll
      for (int t = 0; t < 9; t++)
        if (tid < 512) *reinterpret_cast<u32x4*>(wl + t * 8192 + tid * 16) = wst[t];
    }
  };
  // weight fragments of column kx of chunk `c` into B[par]
  auto wload = [&](int par, int c, int kx) {
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int p = 0; p < 2; p++) {
        if (MODE) B[par][ky][p] = *reinterpret_cast<const u32x4*>(wl + (ky * 3 + kx) * 8192 + (wn * 2 + p) * 1024 + lane * 16);
        else B[par][ky][p] = W[((ky * 3 + kx) * NCHUNK + c) * blkq + wq + p * 64];
      }
  };
  // fill LDS with something finite
  for (int i = tid; i < (2 * buf_bytes + (MODE ? WL : 0)) / 16; i += NTH) *reinterpret_cast<u32x4*>(smem + i * 16) = (u32x4){0x3c003c00u, 0x38003800u, 0x34003400u, 0x30003000u};
  __syncthreads();
  load_stage(0);
  if (MODE) store_w();
  store_x(0);
  __syncthreads();
  wload(0, 0, 0);
  if (W2) wload(1, 0, 1);
  int buf = 0;
  for (int s = 0; s < stages; s++) {
    if (!(OPT & (OPT_WFIRST | OPT_LATE))) load_stage(s + 1);
    const char* xs = smem + buf * buf_bytes + xb;
    constexpr int Q = 3 * (MS + 2), XD = X2 ? 3 : 2;
    u32x4 Xf[XD][2];
    Xf[0][0] = *reinterpret_cast<const u32x4*>(xs);
    Xf[0][1] = *reinterpret_cast<const u32x4*>(xs + plane_b);
    if (X2) {
      Xf[1][0] = *reinterpret_cast<const u32x4*>(xs + PW * 16);
      Xf[1][1] = *reinterpret_cast<const u32x4*>(xs + PW * 16 + plane_b);
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int kx = q / (MS + 2), pr = q % (MS + 2);
      if (pr == 0 && !(ABL & NO_W)) {
        if (W2) {  // column kx lives in set kx; request column kx + 2 (the next stage's kx - 1 for kx >= 1) into the set column kx - 1 left
          if (kx == 0) wload(2, s & 7, 2);
          else wload(kx - 1, (s + 1) & 7, kx - 1);
        } else {
          if (kx + 1 < 3) wload((kx + 1) & 1, s & 7, kx + 1);
          else if (!MODE) wload(1, (s + 1) & 7, 0);  // (MODE 1: the next stage's weights are not in LDS yet: fetched after the barrier)
        }
      }
      if ((OPT & OPT_WFIRST) && q == 0) load_stage(s + 1);
      if ((OPT & OPT_LATE) && q == MS + 2) load_stage(s + 1);
      if (q + (XD - 1) < Q && !(ABL & NO_X)) {
        const int kx1 = (q + XD - 1) / (MS + 2), pr1 = (q + XD - 1) % (MS + 2);
        const char* ap = xs + (pr1 * PW + kx1) * 16;
        Xf[(q + XD - 1) % XD][0] = *reinterpret_cast<const u32x4*>(ap);
        Xf[(q + XD - 1) % XD][1] = *reinterpret_cast<const u32x4*>(ap + plane_b);
      }
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 xh = Xf[(ABL & NO_X) ? 0 : (q % XD)][0], xl = Xf[(ABL & NO_X) ? 0 : (q % XD)][1];
      if (!(ABL & NO_MFMA)) {
        if (OPT & OPT_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++)
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            const int ms = pr - ky;
            if (ms < 0 || ms >= MS) continue;
            const u32x4* wv = B[(ABL & NO_W) ? 0 : W2 ? kx : (kx & 1)][ky];
            acc[ms] = t3 == 0 ? mfma(wv[1], xh, acc[ms]) : t3 == 1 ? mfma(wv[0], xl, acc[ms]) : mfma(wv[0], xh, acc[ms]);
          }
        if (OPT & OPT_PRIO) __builtin_amdgcn_s_setprio(0);
      } else {
        acc[0] += __builtin_bit_cast(f32x4, xh) + __builtin_bit_cast(f32x4, xl) + __builtin_bit_cast(f32x4, B[W2 ? kx : (kx & 1)][pr % 3][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!MODE && !(ABL & NO_W) && !W2) {  // the next stage starts on parity 0
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int p = 0; p < 2; p++) B[0][ky][p] = B[1][ky][p];
    }
    if (MODE) {
      __syncthreads();  // every wave is done with this stage's weights in LDS
      store_w();
    }
    store_x(buf ^ 1);
    __syncthreads();
    if (MODE && !(ABL & NO_W)) wload(0, 0, 0);
    buf ^= 1;
  }
  f32x4 r = acc[0] + acc[1] + acc[2] + acc[3];
  out[(blockIdx.y * gridDim.x + blockIdx.x) * NTH + tid] = r.x + r.y + r.z + r.w;
}

template <int MODE, int ABL, int WMX = 2, int OPT = 0>
static void run(const char* name, const u32x4* W, const u32x4* X, float* out, int stages, unsigned xmask) {
  constexpr int WM = MODE ? WMX : 1;
  constexpr int PH = 4 * WM + 2, PPX = (PH * PW + 15) & ~15;
  const size_t smem = 2 * 8 * PPX * 16 + (MODE ? 9 * 8192 : 0);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE, ABL, WMX, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  // 4 cout groups x (CUs * 2 / 4 / WM ...) workgroups: two 4-wave workgroups or one 8-wave workgroup per CU
  dim3 grid(MODE ? cus / 4 : cus * 2 / 4, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<MODE, ABL, WMX, OPT>), grid, dim3(MODE ? 256 * WMX : 256), smem, 0, W, X, out, stages, xmask);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  hipError_t e = hipGetLastError();
  const double waves = (double)grid.x * grid.y * (MODE ? 4 * WMX : 4);
  const double mfmas = waves * stages * 108.0;
  const double us_stage = best * 1e3 / stages;
  printf("%-62s %8.3f ms  %6.2f us per stage  MFMA pipe %5.1f %% of 16 cycles/MFMA at 2.4 GHz  (%6.1f alg. TFLOP/s of 833)%s\n", name, best, us_stage,
         100.0 * (mfmas / (cus * 4.0)) * 16.0 / 2.4e9 / (best * 1e-3), mfmas * 16384.0 / 3.0 / (best * 1e-3) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
}

// MODE 2 ("P3" sketch): ONE 8-wave workgroup per CU = 2 cout waves x 4 pixel waves on a 16 x 16-pixel tile; BOTH operands of a stage travel
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), double-buffered, requested at the START of the
// stage before theirs and waited for (vmcnt(0) + the stage's one barrier) at its end; weight fragments read from LDS by the four pixel
// waves of a cout wave (a quarter of the product form's L2 -> CU weight stream), x fragments by both cout waves.
template <int ABL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void loop3_kernel(const u32x4* __restrict__ W, const u32x4* __restrict__ X,
                                                                                           float* __restrict__ out, int stages, unsigned xmask) {
  constexpr int WM = 4, NTH = 512;
  constexpr int PH = 4 * WM + 2, slots = PH * PW, PPX = (slots + 15) & ~15;
  constexpr int xbytes = 8 * PPX * 16, plane_b = 4 * PPX * 16;
  constexpr int wbytes = 9 * 4096;  // 9 taps x (2 cout sub-tiles x 2 planes x 1 KiB)
  constexpr int XCH = (8 * PPX + 63) / 64;  // 1 KiB chunks of the patch image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave & 1, wm = wave >> 1;
  const int by = blockIdx.y;  // cout group (2 sub-tiles of 16)
  const int xb = ((lane >> 4) * PPX + wm * MS * PW + (lane & 15)) * 16;
  const int blkq = NS_TOTAL * 128;
  f32x4 acc[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ms++) acc[ms] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 B[2][3][2];
  const unsigned g0 = (unsigned)((blockIdx.y * gridDim.x + blockIdx.x) * 8 * PPX);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  auto dma_stage = [&](int s, int buf) {
    char* xd = smem + buf * (xbytes + wbytes);
    char* wd = xd + xbytes;
    if (!(ABL & NO_STAGE)) {
#pragma unroll
      for (int i = 0; i < (XCH + 7) / 8; i++) {
        const int ch = wave + 8 * i;
        if (ch < XCH)
          __builtin_amdgcn_global_load_lds((glb_ptr)(X + ((g0 + (unsigned)s * 7919u * 64u + (unsigned)(ch * 64 + lane)) & xmask)), (lds_ptr)(xd + ch * 1024), 16, 0, 0);
      }
    }
    if (!(ABL & NO_W)) {
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const int ch = wave + 8 * i;  // 36 chunks: tap = ch / 4, (sub-tile, plane) = ch % 4
        if (ch < 36)
          __builtin_amdgcn_global_load_lds((glb_ptr)(W + ((ch >> 2) * NCHUNK + (s & 7)) * blkq + by * 256 + (ch & 3) * 64 + lane), (lds_ptr)(wd + ch * 1024), 16, 0, 0);
      }
    }
  };
  for (int i = tid; i < 2 * (xbytes + wbytes) / 16; i += NTH) *reinterpret_cast<u32x4*>(smem + i * 16) = (u32x4){0x3c003c00u, 0x38003800u, 0x34003400u, 0x30003000u};
  __syncthreads();
  dma_stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int s = 0; s < stages; s++) {
    dma_stage(s + 1, buf ^ 1);  // (every wave is past the barrier that ended the stage which read that buffer)
    const char* xs = smem + buf * (xbytes + wbytes) + xb;
    const char* ws = smem + buf * (xbytes + wbytes) + xbytes + (wn * 2) * 1024 + lane * 16;
    auto wload = [&](int par, int kx) {
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int p = 0; p < 2; p++) B[par][ky][p] = *reinterpret_cast<const u32x4*>(ws + (ky * 3 + kx) * 4096 + p * 1024);
    };
    if (!(ABL & NO_W) || s == 0) wload(0, 0);
    constexpr int Q = 3 * (MS + 2);
    u32x4 Xf[2][2];
    Xf[0][0] = *reinterpret_cast<const u32x4*>(xs);
    Xf[0][1] = *reinterpret_cast<const u32x4*>(xs + plane_b);
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int kx = q / (MS + 2), pr = q % (MS + 2);
      if (pr == 0 && kx + 1 < 3 && !(ABL & NO_W)) wload((kx + 1) & 1, kx + 1);
      if (q + 1 < Q && !(ABL & NO_X)) {
        const int kx1 = (q + 1) / (MS + 2), pr1 = (q + 1) % (MS + 2);
        const char* ap = xs + (pr1 * PW + kx1) * 16;
        Xf[(q + 1) & 1][0] = *reinterpret_cast<const u32x4*>(ap);
        Xf[(q + 1) & 1][1] = *reinterpret_cast<const u32x4*>(ap + plane_b);
      }
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 xh = Xf[(ABL & NO_X) ? 0 : (q & 1)][0], xl = Xf[(ABL & NO_X) ? 0 : (q & 1)][1];
      if (!(ABL & NO_MFMA)) {
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++)
#pragma unroll
          for (int ky = 0; ky < 3; ky++) {
            const int ms = pr - ky;
            if (ms < 0 || ms >= MS) continue;
            const u32x4* wv = B[(ABL & NO_W) ? 0 : (kx & 1)][ky];
            acc[ms] = t3 == 0 ? mfma(wv[1], xh, acc[ms]) : t3 == 1 ? mfma(wv[0], xl, acc[ms]) : mfma(wv[0], xh, acc[ms]);
          }
      } else {
        acc[0] += __builtin_bit_cast(f32x4, xh) + __builtin_bit_cast(f32x4, xl) + __builtin_bit_cast(f32x4, B[kx & 1][pr % 3][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the next stage have landed
    __syncthreads();
    buf ^= 1;
  }
  f32x4 r = acc[0] + acc[1] + acc[2] + acc[3];
  out[(blockIdx.y * gridDim.x + blockIdx.x) * NTH + tid] = r.x + r.y + r.z + r.w;
}

template <int ABL>
static void run3(const char* name, const u32x4* W, const u32x4* X, float* out, int stages, unsigned xmask) {
  constexpr int PH = 4 * 4 + 2, PPX = (PH * PW + 15) & ~15;
  const size_t smem = 2 * (8 * PPX * 16 + 9 * 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loop3_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  dim3 grid(cus / 8, 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop3_kernel<ABL>), grid, dim3(512), smem, 0, W, X, out, stages, xmask);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  hipError_t e = hipGetLastError();
  const double waves = (double)grid.x * grid.y * 8;
  const double mfmas = waves * stages * 108.0;
  printf("%-62s %8.3f ms  %6.2f us per stage  MFMA pipe %5.1f %% of 16 cycles/MFMA at 2.4 GHz  (%6.1f alg. TFLOP/s of 833)%s\n", name, best, best * 1e3 / stages,
         100.0 * (mfmas / (cus * 4.0)) * 16.0 / 2.4e9 / (best * 1e-3), mfmas * 16384.0 / 3.0 / (best * 1e-3) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
}

int main(int argc, char** argv) {
  const int stages = argc > 1 ? atoi(argv[1]) : 512;
  const size_t wq = (size_t)9 * NCHUNK * NS_TOTAL * 128;  // u32x4
  const size_t xq = (size_t)1 << 24;                     // 256 MB of "planes"
  u32x4 *W, *X;
  float* out;
  (void)hipMalloc(&W, wq * 16);
  (void)hipMalloc(&X, xq * 16);
  (void)hipMalloc(&out, 1 << 22);
  std::vector<unsigned> h(wq * 4);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0x3c003c00u ^ (unsigned)((i * 2654435761u) & 0x03ff03ffu);  // fp16 values in [1, 2)
  (void)hipMemcpy(W, h.data(), wq * 16, hipMemcpyHostToDevice);
  (void)hipMemset(X, 0x3c, xq * 16);
  // argv[2]: log2 of the window of "planes" the staging loads walk, in 16-byte granules (24 = 256 MB: every stage's patch comes from
  // HBM; 19 = 8 MB: from L2 / the infinity cache, as a layer's input just written by its producer does)
  const int xlog = argc > 2 ? atoi(argv[2]) : 24;
  const unsigned xmask = (unsigned)(((size_t)1 << xlog) - 1);
  printf("p2_loop: %d stages of 108 MFMAs per wave; weights %.2f MB; staging window %.0f MB\n", stages, wq * 16 / 1e6, ((size_t)1 << xlog) * 16 / 1e6);
  run<0, 0>("product form (2 x 4 waves per CU, weights per wave from L2)", W, X, out, stages, xmask);
  if (argc > 3) {  // round 6: the issue-order / prefetch-depth variants only
    run<0, 0, 2, OPT_WFIRST>("  weights requested before the patch granules", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_LATE>("  patch granules requested at the start of column 1", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_X2>("  x fragments two steps ahead", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_W2>("  weights two columns ahead", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_PRIO>("  s_setprio 1 around the MFMA groups", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_WFIRST | OPT_X2>("  weights first + x two ahead", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_WFIRST | OPT_W2>("  weights first + weights two columns ahead", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_WFIRST | OPT_X2 | OPT_W2>("  weights first + x two ahead + weights two ahead", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_LATE | OPT_X2 | OPT_W2>("  granules at column 1 + x two ahead + weights two ahead", W, X, out, stages, xmask);
    run<0, 0, 2, OPT_WFIRST | OPT_X2 | OPT_W2 | OPT_PRIO>("  all four", W, X, out, stages, xmask);
    run<0, NO_STAGE, 2, OPT_X2 | OPT_W2>("  (x + weights two ahead, no patch staging)", W, X, out, stages, xmask);
    run<0, NO_W | NO_X | NO_STAGE>("  (MFMAs + barrier only)", W, X, out, stages, xmask);
    return 0;
  }
  run<0, NO_STAGE>("  - without the patch staging (loads + LDS stores)", W, X, out, stages, xmask);
  run<0, NO_W>("  - without the weight stream", W, X, out, stages, xmask);
  run<0, NO_X>("  - without the x fragment reads", W, X, out, stages, xmask);
  run<0, NO_W | NO_X | NO_STAGE>("  - MFMAs + barrier only", W, X, out, stages, xmask);
  run<0, NO_MFMA>("  - everything but the MFMAs", W, X, out, stages, xmask);
  run<1, 0>("review form (1 x 8 waves per CU, weights once per WG via LDS)", W, X, out, stages, xmask);
  run<1, NO_STAGE>("  - without the patch staging", W, X, out, stages, xmask);
  run<1, NO_W>("  - without the weight stream (staging + LDS reads)", W, X, out, stages, xmask);
  run<1, NO_X>("  - without the x fragment reads", W, X, out, stages, xmask);
  run<1, NO_MFMA>("  - everything but the MFMAs", W, X, out, stages, xmask);
  run3<0>("P3 sketch (1 x 8 waves: 2 cout x 4 pixel waves, x and W by LDS-DMA)", W, X, out, stages, xmask);
  run3<NO_STAGE>("  - without the patch DMA", W, X, out, stages, xmask);
  run3<NO_W>("  - without the weight stream (DMA + LDS reads)", W, X, out, stages, xmask);
  run3<NO_X>("  - without the x fragment reads", W, X, out, stages, xmask);
  run3<NO_MFMA>("  - everything but the MFMAs", W, X, out, stages, xmask);
  run<1, 0, 3>("review form with 3 pixel waves (1 x 12 waves per CU, 3 per SIMD)", W, X, out, stages, xmask);
  run<1, NO_W, 3>("  - without the weight stream", W, X, out, stages, xmask);
  run<1, NO_X, 3>("  - without the x fragment reads", W, X, out, stages, xmask);
  return 0;
}
