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

// more variants: 5 = 8-byte atomicMin whose result is not used; 6 = atomicOr of one random bit of a 32 MiB bitmap (tab);
// 7 = random 8-byte load; 8 = CAS at WORKGROUP scope into the eighth of the table that belongs to the XCD the wave runs
// on (a workgroup-scope atomic is carried out in that XCD's L2); 9 = the same at agent scope (what the eighth alone buys);
// 10 = 4-byte CAS
template <int MODE>
__global__ __launch_bounds__(256) void k_more(u64* __restrict__ tab, u64 mask, u64 n, u64 salt, u64* __restrict__ sink) {
  u64 acc = 0;
  unsigned xcc = 0;
  if (MODE == 8 || MODE == 9) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
  }
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    const u64 h = mix64(i ^ salt);
    if (MODE == 5) (void)__hip_atomic_fetch_min(tab + (h & mask), i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 6) acc += atomicOr(reinterpret_cast<unsigned*>(tab) + ((h >> 5) & ((1u << 23) - 1u)), 1u << (h & 31));
    if (MODE == 7) acc += tab[h & mask];
    if (MODE == 8 || MODE == 9) {
      u64* b = tab + (((h & mask) >> 3) | ((u64)xcc * ((mask + 1) >> 3)));
      u64 expect = ~0ull;
      if (MODE == 8) __hip_atomic_compare_exchange_strong(b, &expect, i, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_compare_exchange_strong(b, &expect, i, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += expect;
    }
    if (MODE == 10) acc += atomicCAS(reinterpret_cast<unsigned*>(tab) + (h & (2 * mask + 1)), ~0u, (unsigned)i);
    if (MODE == 11) acc += atomicMin(tab + (i & mask), h);                       // neighbouring lanes, neighbouring words
    if (MODE == 12) acc += atomicMin(tab + ((i ^ (h & 0xFFF)) & mask), h);       // ... shuffled inside 32 KiB windows
    if (MODE == 13) acc += atomicOr(reinterpret_cast<unsigned*>(tab) + ((i >> 5) & (2 * mask + 1)), 1u << (i & 31));  // bitmap, in order
  }
  if (acc == 0x1234567ull) sink[0] = acc;
}

// sequential reads of 64-byte records, one record per lane (array of structs): 0 = four 16-byte loads together,
// 1 = the first, a look at it, then the other three; 2 = as 0 but only 12 of every 16 records; 3 = lanes read
// 16-byte pieces in order (fully coalesced) - the reference
template <int MODE>
__global__ __launch_bounds__(256) void k_seq64(const u64x2* __restrict__ recs, u64 n_recs, u64* __restrict__ sink) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_recs; i += (u64)gridDim.x * 256) {
    if (MODE == 3) {
      const u64 w = (i >> 6) * 256 + (i & 63);  // wave-contiguous 16-byte pieces, four instructions
      const u64x2 a = recs[w], b = recs[w + 64], c = recs[w + 128], d = recs[w + 192];
      acc += a.x ^ b.y ^ c.x ^ d.y;
      continue;
    }
    if (MODE == 2 && (i & 15) >= 12) continue;
    const u64x2* s = recs + i * 4;
    const u64x2 a = __builtin_nontemporal_load(s);
    if (MODE == 1 && (a.x & 0xFFFF) == 0x1234) continue;
    const u64x2 b = __builtin_nontemporal_load(s + 1), c = __builtin_nontemporal_load(s + 2), d = __builtin_nontemporal_load(s + 3);
    acc += a.x ^ b.y ^ c.x ^ d.y;
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
  {
    const u64 n_recs = 136000000ull;  // 8.7 GB
    u64x2* recs; CK(hipMalloc(&recs, n_recs * 64));
    CK(hipMemset(recs, 0x5A, n_recs * 64));
    printf("sequential 64-byte records (8.7 GB), one per lane: together %.2f ms, first-then-rest %.2f ms, 12 of 16 %.2f ms, coalesced pieces %.2f ms\n",
           time_ms(2, [&] { hipLaunchKernelGGL(k_seq64<0>, dim3(grid), dim3(256), 0, 0, recs, n_recs, sink); }),
           time_ms(2, [&] { hipLaunchKernelGGL(k_seq64<1>, dim3(grid), dim3(256), 0, 0, recs, n_recs, sink); }),
           time_ms(2, [&] { hipLaunchKernelGGL(k_seq64<2>, dim3(grid), dim3(256), 0, 0, recs, n_recs, sink); }),
           time_ms(2, [&] { hipLaunchKernelGGL(k_seq64<3>, dim3(grid), dim3(256), 0, 0, recs, n_recs, sink); }));
    CK(hipFree(recs));
  }
  printf("%llu M random accesses per launch; time per launch in ms\n", n / 1000000ull);
  printf("%-10s %10s %10s %10s %10s %10s %10s\n", "table", "cas8", "store8", "cas64", "cas64+48B", "load64", "load64+min");
  for (int lg = (argc > 2 ? atoi(argv[2]) : 30); lg <= (argc > 3 ? atoi(argv[3]) : 35); ++lg) {
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
    printf("           min8(no ret) %.2f  or(32MiB bitmap) %.2f  load8 %.2f  cas8 xcd-local wg-scope %.2f  xcd-local agent-scope %.2f  cas4 %.2f\n",
           run(k_more<5>, m8), run(k_more<6>, m8), run(k_more<7>, m8), run(k_more<8>, m8), run(k_more<9>, m8), run(k_more<10>, m8));
    printf("           min8 in order %.2f  min8 shuffled in 32 KiB windows %.2f  bitmap or, in order (one per lane) %.2f\n", run(k_more<11>, m8), run(k_more<12>, m8), run(k_more<13>, m8));
    CK(hipFree(tab));
  }
  return 0;
}
