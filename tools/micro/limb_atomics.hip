// Micro-benchmark (round 5): can the ~580 BatchNorm finalize launches of a training step go away WITHOUT a cross-workgroup hand-shake?
// The producers (forward conv epilogue, bn_bwd_reduce2) hold per-workgroup (sum, sum of squares) partials; today they store them as
// float64 and a one-workgroup-per-channel kernel folds them in a fixed order.  Integer addition is associative, so partials converted to
// a wide fixed-point number (three 40-bit limbs, each in its own int64 slot with 24 bits of head-room: no carries between atomics) can be
// ADDED ATOMICALLY in any order with a bit-reproducible result, fire-and-forget; the consumer kernel rebuilds the sum in its prologue.
// Measured here:
//   producer side   G workgroups x 256 threads, each adding NA limb words -- (0) plain float64 partial stores (today), (1) agent-scope
//                   atomics on ONE set of NA slots, (2) workgroup-scope atomics (executed in the XCD's L2) on a per-XCD set picked by
//                   HW_REG_XCC_ID, (3) agent-scope atomics WITH return (what the round-5 last-arriving-workgroup fold paid)
//   consumer side   256 workgroups x 1024 threads whose prologue reads (a) 2 floats per channel (today: mean, invstd from the finalize
//                   kernel), (b) 6 limb words per channel and does the float64 mean / variance / rsqrt itself, (c) 8 x 6 (per-XCD sets)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/limb_atomics.hip -o tools/micro/bin/limb_atomics
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

constexpr int LIMB_BITS = 40, FRAC_BITS = 56;  // value = (l0 + l1 2^40 + l2 2^80) 2^-56

__device__ __forceinline__ void to_limbs(double v, long long l[3]) {
  // |v| < 2^63: exact to 2^-56 (the fp32 partials this replaces carry 24 bits)
  const double hi = floor(v);                      // integer part, |hi| < 2^63
  const double lo = (v - hi) * 72057594037927936.0;  // fraction * 2^56 in [0, 2^56)
  const long long ih = (long long)hi, il = (long long)lo;
  // q = ih * 2^56 + il  as limbs of 40 bits
  const unsigned long long m40 = (1ull << LIMB_BITS) - 1;
  l[0] = (long long)((unsigned long long)il & m40);
  const long long mid = (il >> LIMB_BITS) + ((ih & ((1ll << 24) - 1)) << 16);  // bits 40..79: 16 from il, 24 from ih
  l[1] = mid;
  l[2] = ih >> 24;  // arithmetic: carries the sign
}
__device__ __forceinline__ double from_limbs(long long l0, long long l1, long long l2) {
  // exact in __int128, then one rounding to float64
  const __int128 q = (__int128)l0 + ((__int128)l1 << LIMB_BITS) + ((__int128)l2 << (2 * LIMB_BITS));
  const long long top = (long long)(q >> 64);
  const unsigned long long bot = (unsigned long long)q;
  return ((double)top * 18446744073709551616.0 + (double)bot) * (1.0 / 72057594037927936.0);
}

__device__ __forceinline__ int xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return (int)(v & 0xf);
}

template <int MODE>
__global__ __launch_bounds__(256) void producer(double* __restrict__ part, long long* __restrict__ acc, int NA, int slots, float seed) {
  // one "partial" per thread < NA / 3 ... the cost model: NA words per workgroup
  for (int i = threadIdx.x; i < NA; i += 256) {
    const double v = (double)(seed * (float)(1 + (i % 97)) + (float)blockIdx.x * 0.001f);
    if constexpr (MODE == 0) {
      part[(int64_t)i * slots + blockIdx.x] = v;
    } else {
      long long l[3];
      to_limbs(v, l);
      const long long w = l[i % 3];
      if constexpr (MODE == 1) __hip_atomic_fetch_add(acc + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (MODE == 2) __hip_atomic_fetch_add(acc + (int64_t)xcc_id() * ((NA + 15) & ~15) + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if constexpr (MODE == 3) {
        const long long old = __hip_atomic_fetch_add(acc + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0x7fffffffffffffffll) part[i] = 1.0;
      }
    }
  }
}

// the fold today: one 256-thread workgroup per channel pair (sum, sumsq interleaved as in the product)
__global__ __launch_bounds__(256) void finalize_today(const double* __restrict__ part, int slots, float* __restrict__ mean, float* __restrict__ invstd) {
  __shared__ double red[2][4];
  const double* p = part + (int64_t)blockIdx.x * 2 * slots;
  double s = 0, ss = 0;
  for (int b = threadIdx.x; b < slots; b += 256) {
    s += p[b];
    ss += p[slots + b];
  }
  for (int o = 32; o; o >>= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = ss;
  }
  __syncthreads();
  if (threadIdx.x) return;
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double mu = s / 524288.0;
  double var = ss / 524288.0 - mu * mu;
  mean[blockIdx.x] = (float)mu;
  invstd[blockIdx.x] = (float)(1.0 / sqrt(fabs(var) + 1e-5));
}

// consumer: a BatchNorm-apply-like stream over `n4` float4 (NHWC, C channels) with the per-channel constants from (KIND 0) two float
// arrays, (KIND 1) one limb set, (KIND 2) eight per-XCD limb sets
template <int KIND>
__global__ __launch_bounds__(1024) void consumer(const float* __restrict__ z, float* __restrict__ out, int64_t n4, int C, const float* __restrict__ mean,
                                                 const float* __restrict__ invstd, const long long* __restrict__ acc) {
  __shared__ float sm[2][512];
  if constexpr (KIND == 0) {
    for (int c = threadIdx.x; c < C; c += 1024) {
      sm[0][c] = mean[c];
      sm[1][c] = invstd[c];
    }
  } else {
    constexpr int SETS = KIND == 2 ? 8 : 1;
    const int NA = 6 * C, NAp = (NA + 15) & ~15;
    for (int c = threadIdx.x; c < C; c += 1024) {
      long long l[6] = {0, 0, 0, 0, 0, 0};
      for (int x = 0; x < SETS; x++)
#pragma unroll
        for (int j = 0; j < 6; j++) l[j] += acc[(int64_t)x * NAp + c * 6 + j];
      const double s = from_limbs(l[0], l[1], l[2]), ss = from_limbs(l[3], l[4], l[5]);
      const double mu = s / 524288.0;
      double var = ss / 524288.0 - mu * mu;
      sm[0][c] = (float)mu;
      sm[1][c] = (float)(1.0 / sqrt(fabs(var) + 1e-5));
    }
  }
  __syncthreads();
  const int c4n = C >> 2;
  for (int64_t t = (int64_t)blockIdx.x * 1024 + threadIdx.x; t < n4; t += (int64_t)gridDim.x * 1024) {
    const int q = (int)(t % c4n) * 4;
    float4 v = reinterpret_cast<const float4*>(z)[t];
    v.x = (v.x - sm[0][q]) * sm[1][q];
    v.y = (v.y - sm[0][q + 1]) * sm[1][q + 1];
    v.z = (v.z - sm[0][q + 2]) * sm[1][q + 2];
    v.w = (v.w - sm[0][q + 3]) * sm[1][q + 3];
    reinterpret_cast<float4*>(out)[t] = v;
  }
}

static float time_loop(int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; i++) f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

int main() {
  double* part;
  long long* acc;
  float *mean, *invstd, *z, *out;
  const int64_t zmax = (int64_t)128 * 64 * 64 * 32;  // the largest branch-1 tensor (16.7 M floats)
  hipMalloc(&part, 1 << 26);
  hipMalloc(&acc, 1 << 22);
  hipMalloc(&mean, 4096);
  hipMalloc(&invstd, 4096);
  hipMalloc(&z, zmax * 4);
  hipMalloc(&out, zmax * 4);
  hipMemset(acc, 0, 1 << 22);
  hipMemset(z, 0, zmax * 4);
  const int reps = 200;

  // ---- correctness of the limb representation and of the two atomic placements -------------------------------------------------
  {
    const int G = 512, C = 64, NA = 6 * C;
    hipMemset(acc, 0, 1 << 22);
    hipLaunchKernelGGL(producer<1>, dim3(G), dim3(256), 0, 0, part, acc, NA, G, 0.37f);
    hipLaunchKernelGGL(producer<2>, dim3(G), dim3(256), 0, 0, part, acc + 65536, NA, G, 0.37f);
    hipDeviceSynchronize();
    std::vector<long long> a(NA), b(8 * 384);
    hipMemcpy(a.data(), acc, NA * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), acc + 65536, 8 * 384 * 8, hipMemcpyDeviceToHost);
    int bad = 0, sets_used = 0;
    for (int x = 0; x < 8; x++) {
      bool any = false;
      for (int i = 0; i < NA; i++) any |= b[x * 384 + i] != 0;
      sets_used += any;
    }
    for (int i = 0; i < NA; i++) {
      long long s = 0;
      for (int x = 0; x < 8; x++) s += b[x * 384 + i];
      bad += s != a[i];
    }
    // host check of one limb triple against the float64 sum
    double ref = 0;
    for (int g = 0; g < G; g++) ref += (double)(0.37f * (float)(1 + (0 % 97)) + (float)g * 0.001f);
    const __int128 q = (__int128)a[0];
    printf("correctness: agent-scope vs per-XCD sets: %d of %d words differ; XCD sets used %d of 8; limb 0 of word 0 = %lld (float64 sum of the partials %.6f)\n",
           bad, NA, sets_used, (long long)q, ref);
  }

  printf("\nproducer: G workgroups x NA limb words each (us per launch, back-to-back, %d launches)\n", reps);
  printf("%6s %6s | %10s %12s %14s %14s\n", "G", "NA", "f64 store", "agent atomic", "XCD-L2 atomic", "agent + return");
  for (int G : {512, 1024, 128, 32})
    for (int NA : {192, 384, 768, 1536}) {
      const float t0 = time_loop(reps, [&] { hipLaunchKernelGGL(producer<0>, dim3(G), dim3(256), 0, 0, part, acc, NA, G, 0.37f); });
      const float t1 = time_loop(reps, [&] { hipLaunchKernelGGL(producer<1>, dim3(G), dim3(256), 0, 0, part, acc, NA, G, 0.37f); });
      const float t2 = time_loop(reps, [&] { hipLaunchKernelGGL(producer<2>, dim3(G), dim3(256), 0, 0, part, acc, NA, G, 0.37f); });
      const float t3 = time_loop(reps, [&] { hipLaunchKernelGGL(producer<3>, dim3(G), dim3(256), 0, 0, part, acc, NA, G, 0.37f); });
      printf("%6d %6d | %10.2f %12.2f %14.2f %14.2f\n", G, NA, t0, t1, t2, t3);
    }

  printf("\nchain per layer (us): [producer + finalize + consumer] today  vs  [producer with atomics + consumer with limb prologue]\n");
  printf("%6s %10s | %10s %10s %10s | %12s %12s | %12s %12s\n", "C", "floats", "prod f64", "finalize", "cons(2f)", "prod agent", "cons(limbs)", "prod XCD", "cons(8 sets)");
  struct L { int C; int64_t n; int G; };
  for (L l : {L{32, zmax, 512}, L{64, zmax / 2, 512}, L{128, zmax / 4, 128}, L{256, zmax / 8, 32}}) {
    const int NA = 6 * l.C;
    const int64_t n4 = l.n / 4;
    const float p0 = time_loop(reps, [&] { hipLaunchKernelGGL(producer<0>, dim3(l.G), dim3(256), 0, 0, part, acc, 2 * l.C, l.G, 0.37f); });
    const float f0 = time_loop(reps, [&] { hipLaunchKernelGGL(finalize_today, dim3(l.C), dim3(256), 0, 0, part, l.G, mean, invstd); });
    const float c0 = time_loop(reps, [&] { hipLaunchKernelGGL(consumer<0>, dim3(256), dim3(1024), 0, 0, z, out, n4, l.C, mean, invstd, acc); });
    const float p1 = time_loop(reps, [&] { hipLaunchKernelGGL(producer<1>, dim3(l.G), dim3(256), 0, 0, part, acc, NA, l.G, 0.37f); });
    const float c1 = time_loop(reps, [&] { hipLaunchKernelGGL(consumer<1>, dim3(256), dim3(1024), 0, 0, z, out, n4, l.C, mean, invstd, acc); });
    const float p2 = time_loop(reps, [&] { hipLaunchKernelGGL(producer<2>, dim3(l.G), dim3(256), 0, 0, part, acc, NA, l.G, 0.37f); });
    const float c2 = time_loop(reps, [&] { hipLaunchKernelGGL(consumer<2>, dim3(256), dim3(1024), 0, 0, z, out, n4, l.C, mean, invstd, acc); });
    // the chains, interleaved as a step would run them
    const float ch0 = time_loop(reps, [&] {
      hipLaunchKernelGGL(producer<0>, dim3(l.G), dim3(256), 0, 0, part, acc, 2 * l.C, l.G, 0.37f);
      hipLaunchKernelGGL(finalize_today, dim3(l.C), dim3(256), 0, 0, part, l.G, mean, invstd);
      hipLaunchKernelGGL(consumer<0>, dim3(256), dim3(1024), 0, 0, z, out, n4, l.C, mean, invstd, acc);
    });
    const float ch1 = time_loop(reps, [&] {
      hipLaunchKernelGGL(producer<1>, dim3(l.G), dim3(256), 0, 0, part, acc, NA, l.G, 0.37f);
      hipLaunchKernelGGL(consumer<1>, dim3(256), dim3(1024), 0, 0, z, out, n4, l.C, mean, invstd, acc);
    });
    const float ch2 = time_loop(reps, [&] {
      hipLaunchKernelGGL(producer<2>, dim3(l.G), dim3(256), 0, 0, part, acc, NA, l.G, 0.37f);
      hipLaunchKernelGGL(consumer<2>, dim3(256), dim3(1024), 0, 0, z, out, n4, l.C, mean, invstd, acc);
    });
    printf("%6d %10lld | %10.2f %10.2f %10.2f | %12.2f %12.2f | %12.2f %12.2f   chains: today %.2f  agent %.2f  XCD %.2f\n", l.C, (long long)l.n, p0, f0, c0, p1, c1,
           p2, c2, ch0, ch1, ch2);
  }
  return 0;
}
