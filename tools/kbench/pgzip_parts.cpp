// What the parts of the many-core gzip reader (fastq_utils_amd/host/fq_pgzip.h) cost on ONE core of this host:
// finding a block, inflating to 16-bit symbols with an unknown window, markers -> bytes, CRC-32 (ours and zlib's),
// and zlib's own inflate of the same file.  argv: a single-member .gz.  CPU only (tools/pgzip_scan.sh builds and runs it).
#include "../../fastq_utils_amd/host/fq_pgzip.h"
#include <chrono>
using namespace fqhost::pgz;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  if (argc < 2) return 9;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 8;
  fseek(f, 0, SEEK_END);
  size_t n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n > (512u << 20)) n = 512u << 20;  // the first 512 MiB are enough
  std::vector<uint8_t> b(n + kSlack, 0);
  if (fread(b.data(), 1, n, f) != n) return 7;
  fclose(f);
  Inflate16 dec;
  double t0 = now();
  const uint64_t at = find_block(b.data(), n, (uint64_t)(n / 4) * 8, (uint64_t)n * 8, dec);
  printf("find_block: %llu bits searched in %.3f ms\n", (unsigned long long)(at - (uint64_t)(n / 4) * 8), (now() - t0) * 1e3);
  ChunkState c;
  if (!c.out.init(n * 8)) return 6;
  c.out.window_unknown();
  c.bit = at;
  t0 = now();
  const Status st = dec.run(c, b.data(), n, false, ~0ull, 1ull << 40);
  double s = now() - t0;
  const size_t m = c.safe_bytes;
  printf("inflate to symbols (unknown window) + narrowing block by block: status %d, %zu bytes in %.3f s = %.0f MB/s\n", (int)st, m, s, m / s / 1e6);
  printf("markers left: %.2f %% of the bytes\n", 100.0 * c.safe_marks / m);
  std::vector<uint8_t> out(m);
  for (int rep = 0; rep < 2; ++rep) {
    t0 = now();
    const uint32_t c1 = crc32_16(0, c.out.bytes, m);
    const double s2 = now() - t0;
    t0 = now();
    memcpy(out.data(), c.out.bytes, m);
    const double s1 = now() - t0;
    t0 = now();
    uint32_t c2 = 0;
    for (size_t o = 0; o < m; o += 1u << 30) c2 = (uint32_t)crc32(c2, out.data() + o, (uInt)std::min<size_t>(m - o, 1u << 30));
    const double s3 = now() - t0;
    printf("copy %.0f MB/s, crc32_16 %.0f MB/s, zlib %s crc32 %.0f MB/s (%s)\n", m / s1 / 1e6, m / s2 / 1e6, zlibVersion(), m / s3 / 1e6,
           c1 == c2 ? "same" : "DIFFERENT");
  }
  // zlib's inflate of the same bytes, from the member's start
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  inflateInit2(&zs, 15 + 16);
  std::vector<uint8_t> o2(64u << 20);
  zs.next_in = b.data();
  zs.avail_in = (uInt)n;
  size_t total = 0;
  t0 = now();
  for (;;) {
    zs.next_out = o2.data();
    zs.avail_out = (uInt)o2.size();
    const int rc = inflate(&zs, Z_NO_FLUSH);
    total += o2.size() - zs.avail_out;
    if (rc != Z_OK || zs.avail_in == 0) break;
  }
  s = now() - t0;
  printf("zlib inflate: %zu bytes in %.3f s = %.0f MB/s\n", total, s, total / s / 1e6);
  return 0;
}
