// TEST DRIVER: the many-core gzip reader (fastq_utils_amd/host/fq_pgzip.h) against zlib's gzread on the same file.
// argv: file threads chunk_bytes [read_size [timing]]   (timing: no gzread, no comparison - only how long the reader takes)
// Reads the file through ParallelGunzip in calls of read_size bytes (default: odd sizes that change from call to call)
// and through gzread; exit status 0 = the same bytes (or both refuse the file, with the same message), 1 = they differ.
// One line of statistics goes to stdout: "ok|differ bytes=.. batches=.. joined=.. not_found=.. discarded=..
// serial_bits=.. members=.. fell_back=0|1 why=.. error=.. | zlib_error=.. zlib_s=.. pgz_s=.."
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../fastq_utils_amd/host/fq_pgzip.h"

int main(int argc, char** argv) {
  if (argc < 4) return 9;
  const char* path = argv[1];
  const unsigned threads = (unsigned)atoi(argv[2]);
  const size_t chunk = strtoull(argv[3], nullptr, 10);
  const size_t fixed = argc > 4 ? strtoull(argv[4], nullptr, 10) : 0;
  const bool timing_only = argc > 5;
  // zlib's view
  std::vector<char> want;
  std::string zerr;
  const auto t0 = std::chrono::steady_clock::now();
  if (!timing_only) {
    gzFile g = gzopen(path, "r");
    if (!g) return 8;
    gzbuffer(g, 1 << 20);
    std::vector<char> buf(1 << 22);
    for (;;) {
      const int got = gzread(g, buf.data(), (unsigned)buf.size());
      if (got < 0) {
        int en = 0;
        zerr = gzerror(g, &en);
        break;
      }
      if (got == 0) break;
      want.insert(want.end(), buf.data(), buf.data() + got);
    }
    gzclose(g);
  }
  const auto t1 = std::chrono::steady_clock::now();
  const int fd = open(path, O_RDONLY);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0) return 8;
  fqhost::ParallelGunzip pg(fd, (uint64_t)sb.st_size, path, threads, chunk);
  std::vector<char> have;
  have.reserve(want.size() + 1);
  bool at_end = false;
  size_t step = 1;
  std::vector<char> buf;
  double in_read = 0;
  size_t total_timing = 0;
  while (!at_end && !pg.failed()) {
    const size_t ask = fixed ? fixed : (step = step * 7 % 1000003 + 1, (step % 5 == 0 ? 1 : step * 11));
    buf.resize(ask);
    const auto r0 = std::chrono::steady_clock::now();
    const size_t got = pg.read(buf.data(), ask, &at_end);
    in_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
    if (!timing_only) have.insert(have.end(), buf.data(), buf.data() + got);
    else total_timing += got;
    if (got < ask && !at_end && !pg.failed()) {
      printf("differ short read without end\n");
      return 1;
    }
  }
  close(fd);
  const auto t2 = std::chrono::steady_clock::now();
  const fqhost::ParallelGunzip::Stats& s = pg.stats();
  if (timing_only) {
    printf("timing bytes=%zu threads=%u chunk=%zu in_read=%.3f GBps=%.2f (load %.2f decode %.2f join %.2f windows %.2f narrow %.2f) joined=%llu fell_back=%d | thread seconds: decode sum %.2f slowest %.2f, narrow sum %.2f slowest %.2f\n",
           total_timing, threads, chunk, in_read, total_timing / in_read / 1e9, s.s_load, s.s_decode, s.s_join, s.s_windows, s.s_narrow,
           (unsigned long long)s.chunks_joined, s.fell_back ? 1 : 0, s.t_decode_sum, s.t_decode_max, s.t_narrow_sum, s.t_narrow_max);
    return 0;
  }
  // (the reader sums CRC-32 with a routine of its own: it must be zlib's function, at every alignment and length)
  for (size_t off = 0; off < 9 && off < have.size(); ++off) {
    const size_t len = std::min<size_t>(have.size() - off, 100000 + 7 * off);
    const uint32_t a = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(have.data() + off), (uInt)len);
    const uint32_t b = fqhost::pgz::crc32_16((uint32_t)crc32(0L, Z_NULL, 0), reinterpret_cast<const uint8_t*>(have.data() + off), len);
    if (a != b) {
      printf("differ crc32_16\n");
      return 1;
    }
  }
  bool same;
  if (!zerr.empty() || pg.failed()) {
    // both must refuse, with the same text; what was handed out before may differ in length (zlib's gzread drops the
    // output of the call that meets the error), but never in content
    same = !zerr.empty() && pg.failed() && zerr == pg.error();
    const size_t n = std::min(want.size(), have.size());
    same = same && std::equal(want.begin(), want.begin() + (long)n, have.begin());
  } else {
    same = want == have;
  }
  printf("%s bytes=%zu batches=%llu joined=%llu not_found=%llu discarded=%llu serial_bits=%llu members=%llu fell_back=%d why=%s error=%s | zlib_error=%s zlib_s=%.3f pgz_s=%.3f in_read=%.3f (load %.2f decode %.2f join %.2f windows %.2f narrow %.2f)\n",
         same ? "ok" : "differ", have.size(), (unsigned long long)s.batches, (unsigned long long)s.chunks_joined,
         (unsigned long long)s.chunks_not_found, (unsigned long long)s.chunks_discarded, (unsigned long long)s.serial_bits,
         (unsigned long long)s.members, s.fell_back ? 1 : 0, s.why.empty() ? "-" : s.why.c_str(),
         pg.failed() ? pg.error().c_str() : "-", zerr.empty() ? "-" : zerr.c_str(), std::chrono::duration<double>(t1 - t0).count(),
         std::chrono::duration<double>(t2 - t1).count(), in_read, s.s_load, s.s_decode, s.s_join, s.s_windows, s.s_narrow);
  return same ? 0 : 1;
}
