// What the FIRST copy out of pinned memory costs a process, by size (the runtime prepares its copy path lazily, and
// fqg_open pays that on purpose: see first_large_copy in csrc/fqg_abi.hip).  `firstcopy <bytes>`: times, in a fresh
// process, the allocations, the first copy of that size on a new stream, a second one, and a first copy back.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1u << 20);
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || !nd) return 1;
  hipSetDevice(0);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  void* warm = nullptr;
  hipMalloc(&warm, 256);
  double t0 = now();
  void *h = nullptr, *d = nullptr;
  hipHostMalloc(&h, n, hipHostMallocPortable);
  double t1 = now();
  hipMalloc(&d, n);
  double t2 = now();
  hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
  hipStreamSynchronize(st);
  double t3 = now();
  hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
  hipStreamSynchronize(st);
  double t4 = now();
  hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
  hipStreamSynchronize(st);
  double t5 = now();
  hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
  hipStreamSynchronize(st);
  double t6 = now();
  hipHostFree(h);
  hipFree(d);
  double t7 = now();
  printf("%zu bytes: hipHostMalloc %.2f ms, hipMalloc %.2f, first H2D %.2f, second H2D %.2f, first D2H %.2f, second D2H %.2f, frees %.2f\n", n,
         t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6);
  return 0;
}
