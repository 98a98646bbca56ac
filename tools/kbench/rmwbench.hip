// rmwbench.hip - developer micro-benchmark (not part of the product): what does one random access into a table
// cost on MI355X, by table size and access width?  The name index is made of exactly this: 100 M hashed
// accesses (CAS on an 8-byte key, optionally followed by stores to / loads from the rest of a 64-byte bucket).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rmwbench rmwbench.hip && ./rmwbench [Mops]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}

// MODE 0: CAS on the bucket's first word; 1: CAS + 48 bytes of stores behind it (bucket = 64 B);
// 2: load the whole 64-byte bucket; 3: load it and atomicMin on its second word; 4: plain 8-byte store
template <int MODE, int WORDS>
__global__ __launch_bounds__(256) void k_rand(u64* __restrict__ tab, u64 mask, u64 n, u64 salt, u64* __restrict__ sink) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    const u64 h = mix64(i ^ salt);
    u64* b = tab + (h & mask) * WORDS;
    if (MODE == 0) acc += atomicCAS(b, ~0ull, i);
    if (MODE == 1) {
      acc += atomicCAS(b, ~0ull, i);
      u64x2* d = reinterpret_cast<u64x2*>(b) + 1;
      u64x2 x; x.x = h; x.y = i;
      d[0] = x; d[1] = x; d[2] = x;
    }
    if (MODE == 2 || MODE == 3) {
      const u64x2* s = reinterpret_cast<const u64x2*>(b);
      const u64x2 a0 = s[0], a1 = s[1], a2 = s[2], a3 = s[3];
      acc += a0.x ^ a1.x ^ a2.y ^ a3.y;
      if (MODE == 3) acc += atomicMin(b + 1, i);
    }
    if (MODE == 4) b[0] = i;
  }
  if (acc == 0x1234567ull) sink[0] = acc;
}

static double time_ms(int reps, const std::function<void()>& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const u64 n = (argc > 1 ? strtoull(argv[1], 0, 10) : 100ull) * 1000000ull;
  u64* sink; CK(hipMalloc(&sink, 64));
  const unsigned grid = 256 * 16;
  printf("%llu M random accesses per launch; time per launch in ms\n", n / 1000000ull);
  printf("%-10s %10s %10s %10s %10s %10s %10s\n", "table", "cas8", "store8", "cas64", "cas64+48B", "load64", "load64+min");
  for (int lg = 30; lg <= 35; ++lg) {
    const u64 bytes = 1ull << lg;
    u64* tab;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("%4llu GiB: allocation failed\n", bytes >> 30); break; }
    CK(hipMemset(tab, 0xFF, bytes));
    u64 salt = 1;
    const u64 m8 = bytes / 8 - 1, m64 = bytes / 64 - 1;
    auto run = [&](auto kernel, u64 mask) {
      return time_ms(2, [&] { CK(hipMemsetAsync(tab, 0xFF, 64, 0)); hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, tab, mask, n, salt++, sink); });
    };
    const double a = run(k_rand<0, 1>, m8);
    const double e = run(k_rand<4, 1>, m8);
    const double b = run(k_rand<0, 8>, m64);
    const double c = run(k_rand<1, 8>, m64);
    const double d = run(k_rand<2, 8>, m64);
    const double f = run(k_rand<3, 8>, m64);
    printf("%4llu GiB   %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", bytes >> 30, a, e, b, c, d, f);
    CK(hipFree(tab));
  }
  return 0;
}
