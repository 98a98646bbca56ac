// regbench - can a file in the page cache be handed to the GPU WITHOUT a copy into pinned staging slots?
// (mmap + hipHostRegister of page-cache pages instead of pread into pinned memory.)
//   ./regbench [GB=8] [piece MiB=256]
// Makes a tmpfs file, maps it (MAP_SHARED, read-only) and times, per piece and with 1 / 2 / 4 / 8 threads registering
// different pieces at once: hipHostRegister (read-only flag, then default flags), the host-to-device copy from the
// registered range, hipHostUnregister - against pread into a pinned slot + the same copy (what the programs do).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                            \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 8, piece = (argc > 2 ? (size_t)atoi(argv[2]) : 256) << 20;
  const size_t total = gb << 30, n_pieces = total / piece;
  char path[] = "/dev/shm/regbench_XXXXXX";
  const int fd = mkstemp(path);
  if (fd < 0 || ftruncate(fd, (off_t)total) != 0) return 2;
  {  // touch every page (a real file's pages are in the page cache because somebody wrote or read them)
    std::vector<char> buf(64 << 20, 'A');
    for (size_t o = 0; o < total; o += buf.size()) (void)!pwrite(fd, buf.data(), buf.size(), (off_t)o);
  }
  char* map = (char*)mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
  if (map == MAP_FAILED) return 3;
  CK(hipSetDevice(0));
  char* dev = nullptr;
  CK(hipMalloc(&dev, piece * 2));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // baseline: pread into a pinned slot (12 threads), then the copy
  {
    char* pin = nullptr;
    CK(hipHostMalloc(&pin, piece, hipHostMallocDefault));
    double t_read = 0, t_copy = 0;
    for (size_t k = 0; k < n_pieces; ++k) {
      const double a = now();
      std::vector<std::thread> th;
      for (int t = 0; t < 12; ++t)
        th.emplace_back([&, t] {
          const size_t part = piece / 12, from = (size_t)t * part, len = t == 11 ? piece - from : part;
          (void)!pread(fd, pin + from, len, (off_t)(k * piece + from));
        });
      for (auto& x : th) x.join();
      const double b = now();
      CK(hipMemcpyAsync(dev, pin, piece, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      t_read += b - a;
      t_copy += now() - b;
    }
    printf("pread(12 threads) into a pinned slot: %.2f GB/s; copy of the slot: %.2f GB/s; one after the other: %.2f GB/s\n",
           total / t_read / 1e9, total / t_copy / 1e9, total / (t_read + t_copy) / 1e9);
    CK(hipHostFree(pin));
  }
  for (unsigned flags : {(unsigned)hipHostRegisterReadOnly, (unsigned)hipHostRegisterDefault, 100u + (unsigned)hipHostRegisterReadOnly}) {
    const bool populate = flags >= 100u;  // MADV_POPULATE_READ by the threads first, then ONE registration per piece
    if (populate) flags -= 100u;
    for (int T : {1, 2, 4, 8, 16}) {
      // a fresh mapping for every configuration: what a program pays is the COLD cost (page tables to fill)
      munmap(map, total);
      map = (char*)mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
      if (map == MAP_FAILED) return 3;
      if (populate) {
        double t_pop = 0, t_reg = 0, t_copy = 0;
        for (size_t k = 0; k < n_pieces; ++k) {
          const double a = now();
          std::vector<std::thread> th;
          for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
              const size_t part = (piece / (size_t)T) & ~(size_t)4095, from = (size_t)t * part, len = t == T - 1 ? piece - from : part;
              if (madvise(map + k * piece + from, len, 22 /* MADV_POPULATE_READ */) != 0) {
                volatile char sink = 0;
                for (size_t o = 0; o < len; o += 4096) sink += map[k * piece + from + o];
              }
            });
          for (auto& x : th) x.join();
          const double b = now();
          CK(hipHostRegister(map + k * piece, piece, flags));
          const double c = now();
          CK(hipMemcpyAsync(dev, map + k * piece, piece, hipMemcpyHostToDevice, st));
          CK(hipStreamSynchronize(st));
          t_pop += b - a;
          t_reg += c - b;
          t_copy += now() - c;
          (void)hipHostUnregister(map + k * piece);
        }
        printf("populate with %d threads %.2f GB/s, then register %.2f GB/s, copy %.2f GB/s\n", T, total / t_pop / 1e9,
               total / t_reg / 1e9, total / t_copy / 1e9);
        continue;
      }
      double t_reg = 0, t_copy = 0, t_unreg = 0;
      bool ok = true;
      for (size_t k0 = 0; k0 < n_pieces && ok; k0 += (size_t)T) {
        const size_t m = std::min<size_t>((size_t)T, n_pieces - k0);
        std::vector<hipError_t> err(m, hipSuccess);
        const double a = now();
        std::vector<std::thread> th;
        for (size_t j = 0; j < m; ++j) th.emplace_back([&, j] { err[j] = hipHostRegister(map + (k0 + j) * piece, piece, flags); });
        for (auto& x : th) x.join();
        const double b = now();
        for (size_t j = 0; j < m; ++j)
          if (err[j] != hipSuccess) {
            printf("hipHostRegister(flags %u) -> %s\n", flags, hipGetErrorString(err[j]));
            ok = false;
          }
        if (!ok) break;
        for (size_t j = 0; j < m; ++j) CK(hipMemcpyAsync(dev + (j & 1) * piece, map + (k0 + j) * piece, piece, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double c = now();
        th.clear();
        for (size_t j = 0; j < m; ++j) th.emplace_back([&, j] { (void)hipHostUnregister(map + (k0 + j) * piece); });
        for (auto& x : th) x.join();
        t_reg += b - a;
        t_copy += c - b;
        t_unreg += now() - c;
      }
      if (ok)
        printf("flags %u, %d threads: register %.2f GB/s, copy from the mapping %.2f GB/s, unregister %.2f GB/s\n", flags, T,
               total / t_reg / 1e9, total / t_copy / 1e9, total / t_unreg / 1e9);
    }
  }
  // the copy straight from the mapping, not registered at all (the runtime stages it)
  {
    const double a = now();
    for (size_t k = 0; k < n_pieces; ++k) CK(hipMemcpyAsync(dev, map + k * piece, piece, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    printf("copy from the unregistered mapping: %.2f GB/s\n", total / (now() - a) / 1e9);
  }
  munmap(map, total);
  close(fd);
  unlink(path);
  return 0;
}
