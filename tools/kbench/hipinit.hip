// What a process start on the GPU box costs before any of this repository's code runs: hipInit + a device + one
// tiny allocation, and (argument "k") one empty kernel launch, which makes the runtime load this file's code object.
// Used with /usr/bin/time to read the CPU seconds of a start (tests/: the golden sweeps are bounded by them).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_nothing() {}
int main(int argc, char**) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || !n) return 1;
  hipSetDevice(0);
  void* p = nullptr;
  hipMalloc(&p, 256);
  if (argc > 1) {
    hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, 0);
    hipDeviceSynchronize();
  }
  hipFree(p);
  return 0;
}
