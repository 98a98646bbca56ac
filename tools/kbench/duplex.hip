// duplex.hip - does this box carry host-to-device and device-to-host copies at once?  Pinned host memory, two streams.
//   hipcc --offload-arch=gfx950 -O3 -o duplex duplex.hip && ./duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t piece = 512ull << 20;
  const int n = 16;  // 8 GiB each way
  char *h_in, *h_out, *d_in, *d_out;
  CK(hipHostMalloc(&h_in, piece)); CK(hipHostMalloc(&h_out, piece));
  CK(hipMalloc(&d_in, piece)); CK(hipMalloc(&d_out, piece));
  for (size_t i = 0; i < piece; i += 4096) h_in[i] = 1, h_out[i] = 1;
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  auto run = [&](bool up, bool down, const char* label) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int i = 0; i < n; ++i) {
      if (up) CK(hipMemcpyAsync(d_in, h_in, piece, hipMemcpyHostToDevice, a));
      if (down) CK(hipMemcpyAsync(h_out, d_out, piece, hipMemcpyDeviceToHost, b));
    }
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
    const double s = now() - t0, gb = (double)piece * n / 1e9;
    printf("%-34s %.3f s:%s%s\n", label, s, up ? (std::string(" up ") + std::to_string(gb / s).substr(0, 5) + " GB/s").c_str() : "",
           down ? (std::string(" down ") + std::to_string(gb / s).substr(0, 5) + " GB/s").c_str() : "");
  };
  run(true, false, "host -> device alone");
  run(false, true, "device -> host alone");
  run(true, true, "both at once (two streams)");
  run(true, true, "both at once (again)");
  return 0;
}
