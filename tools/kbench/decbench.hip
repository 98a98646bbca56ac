// decbench.hip - developer micro-benchmark (not part of the product): what does turning a capture record into a
// name key (name_from_record, fqg_index_kernels.hip) cost next to loading it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o decbench decbench.hip && ./decbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include "../../fastq_utils_amd/csrc/fqg_device.h"
#include "../../fastq_utils_amd/csrc/fqg_kernels.hip"
#include "../../fastq_utils_amd/csrc/fqg_stream_kernels.hip"
#include "../../fastq_utils_amd/csrc/fqg_index_kernels.hip"
using namespace fqg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_fill(unsigned long long* recs, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    uint8_t b[64];
    for (int o = 0; o < 60; ++o) b[4 + o] = o < 44 ? synth_header_byte(i, o + 1, 1) : (uint8_t)"ACGT"[(i + o) & 3];
    const uint32_t meta = 44u | ((uint32_t)((i & 15) * 4) << 10) | (3u << 19);
    __builtin_memcpy(b, &meta, 4);
    for (int k = 0; k < 8; ++k) __builtin_memcpy(&recs[i * 8 + k], b + 8 * k, 8);
  }
}

// MODE 0: load only; 1: full name_from_record; 2: the 12-of-16 live pattern + full
template <int MODE>
__global__ __launch_bounds__(256) void k_dec(const unsigned long long* __restrict__ recs, uint64_t n, int fmt, unsigned long long* __restrict__ sink) {
  unsigned long long acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    if (MODE == 2 && (i & 15) >= 12) continue;
    unsigned long long w[kNameRecWords];
    const u64x2_t* src = reinterpret_cast<const u64x2_t*>(recs + i * kNameRecWords);
#pragma unroll
    for (uint32_t k = 0; k < kNameRecWords / 2; ++k) {
      const u64x2_t x = __builtin_nontemporal_load(src + k);
      w[2 * k] = x.x;
      w[2 * k + 1] = x.y;
    }
    if (MODE == 0) {
      acc += w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[4] ^ w[5] ^ w[6] ^ w[7];
    } else {
      NameKey k;
      bool at;
      uint32_t v;
      const bool ok = name_from_record(w, fmt, 0, k, &at, &v);
      acc += ok ? (k.h ^ k.nm[0] ^ k.nm[3] ^ k.n ^ v) : 1;
    }
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

int main() {
  const uint64_t n = 136000000ull;
  unsigned long long *recs, *sink;
  CK(hipMalloc(&recs, n * 64)); CK(hipMalloc(&sink, 64));
  hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, recs, n);
  CK(hipDeviceSynchronize());
  const unsigned grid = 256 * 16;
  printf("136 M capture records: load only %.2f ms, + name_from_record (Casava) %.2f ms, (whole line) %.2f ms, 12 of 16 live + Casava %.2f ms\n",
         time_ms(2, [&] { hipLaunchKernelGGL(k_dec<0>, dim3(grid), dim3(256), 0, 0, recs, n, FQG_NAME_CASAVA18, sink); }),
         time_ms(2, [&] { hipLaunchKernelGGL(k_dec<1>, dim3(grid), dim3(256), 0, 0, recs, n, FQG_NAME_CASAVA18, sink); }),
         time_ms(2, [&] { hipLaunchKernelGGL(k_dec<1>, dim3(grid), dim3(256), 0, 0, recs, n, FQG_NAME_DEFAULT, sink); }),
         time_ms(2, [&] { hipLaunchKernelGGL(k_dec<2>, dim3(grid), dim3(256), 0, 0, recs, n, FQG_NAME_CASAVA18, sink); }));
  return 0;
}
