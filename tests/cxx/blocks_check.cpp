// TEST DRIVER: the record-block cutter of the multi-GPU fastq_pre_barcodes (fastq_utils_amd/host/fq_blocks.h) without a
// GPU (pinned allocation = malloc), also under the sanitizers (tests/test_sanitizers.py).
// argv: file records_per_block n_consumers.  Checks, against the file read whole: the blocks in seq order concatenate
// to the file; block k starts at record k*B; every block but the last holds exactly 4*B lines and ends with '\n'; the
// last one is the only one flagged final.  Prints "<blocks> <bytes> ok" or the first violation.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <map>
#include <thread>

#include "../../fastq_utils_amd/host/fq_blocks.h"

extern "C" void* fqg_host_alloc(fqg_ctx*, size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void fqg_host_free(fqg_ctx*, void* p) { free(p); }

int main(int argc, char** argv) {
  if (argc < 4) return 9;
  const uint64_t B = strtoull(argv[2], nullptr, 10);
  const int n_cons = atoi(argv[3]);
  std::string whole;
  {
    gzFile g = gzopen(argv[1], "r");
    if (!g) return 8;
    char buf[1 << 16];
    int got;
    while ((got = gzread(g, buf, sizeof buf)) > 0) whole.append(buf, (size_t)got);
    gzclose(g);
  }
  struct Seen {
    std::string bytes;
    uint64_t first_record, lines;
    bool final;
  };
  std::map<uint64_t, Seen> seen;
  std::mutex mu;
  {
    // consumers that stop with blocks held (the program's error path): abort() lets the producer and every waiter go
    fqhost::RecordBlocks src(nullptr, argv[1], 3);
    if (whole.compare(0, src.peek_size(), std::string(src.peek(), src.peek_size())) != 0) return printf("peek differs\n"), 1;
    src.start(B);
    std::mutex fetch;
    std::atomic<int> held{0};
    auto hoard = [&] {
      fqhost::Block b;
      for (;;) {
        std::lock_guard<std::mutex> lk(fetch);
        if (!src.next(&b)) return;
        ++held;  // never released
        if (b.final) return;
      }
    };
    std::thread a(hoard), b(hoard);
    for (int spin = 0; spin < 150 && held.load() < 3; ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    src.abort();
    a.join();
    b.join();
  }
  {
    fqhost::RecordBlocks src(nullptr, argv[1], n_cons + 2);
    src.start(B);
    auto work = [&] {
      fqhost::Block b;
      for (;;) {
        {
          std::lock_guard<std::mutex> lk(mu);
          if (!src.next(&b)) return;
        }
        Seen s{std::string(b.data, b.size), b.first_record, b.lines, b.final};
        {
          std::lock_guard<std::mutex> lk(mu);
          seen.emplace(b.seq, std::move(s));
        }
        src.release(b);
        if (b.final) return;
      }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < n_cons; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
  size_t off = 0;
  uint64_t k = 0;
  for (auto& kv : seen) {
    const Seen& s = kv.second;
    if (kv.first != k) return printf("block %llu missing\n", (unsigned long long)k), 1;
    if (off + s.bytes.size() > whole.size() || whole.compare(off, s.bytes.size(), s.bytes) != 0)
      return printf("block %llu differs from the file at %zu\n", (unsigned long long)k, off), 1;
    if (s.first_record != k * B) return printf("block %llu: first_record %llu\n", (unsigned long long)k, (unsigned long long)s.first_record), 1;
    uint64_t nl = 0;
    for (char c : s.bytes) nl += c == '\n';
    if (nl != s.lines) return printf("block %llu: %llu lines counted, %llu reported\n", (unsigned long long)k, (unsigned long long)nl, (unsigned long long)s.lines), 1;
    const bool last = k + 1 == seen.size();
    if (s.final != last) return printf("block %llu: final flag\n", (unsigned long long)k), 1;
    if (!last && (nl != 4 * B || s.bytes.back() != '\n'))
      return printf("block %llu: %llu lines, not %llu whole records\n", (unsigned long long)k, (unsigned long long)nl, (unsigned long long)B), 1;
    if (last && nl > 4 * B) return printf("last block: %llu lines\n", (unsigned long long)nl), 1;
    off += s.bytes.size();
    ++k;
  }
  if (off != whole.size()) return printf("blocks cover %zu of %zu bytes\n", off, whole.size()), 1;
  printf("%llu %zu ok\n", (unsigned long long)k, off);
  return 0;
}
