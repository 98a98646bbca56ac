// TEST DRIVER: the record-aligned piece cutter of the multi-GPU -r pass (fastq_utils_amd/host/fq_multi.h) without a
// GPU (pinned allocation = malloc), also under the sanitizers (tests/test_sanitizers.py).
// argv: file piece_bytes n_consumers.  Checks, against the file read whole: the pieces in seq order concatenate to the
// file; every piece starts at line 4*first_record; every piece but the last holds a multiple of four lines and ends
// with '\n'.  Prints "<pieces> <bytes> ok" or the first violation.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <map>
#include <thread>

#include "../../fastq_utils_amd/host/fq_multi.h"

extern "C" void* fqg_host_alloc(fqg_ctx*, size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void fqg_host_free(fqg_ctx*, void* p) { free(p); }

int main(int argc, char** argv) {
  if (argc < 4) return 9;
  const size_t piece = strtoull(argv[2], nullptr, 10);
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
    uint64_t first_record;
    bool final;
  };
  std::map<uint64_t, Seen> seen;
  std::mutex mu;
  {
    // the error path of the program (fastq_info.cpp: join_all after a finding): consumers stop with pieces still held -
    // one of them blocked in next() under the fetch lock, the producer waiting for a slot nobody will release.
    // abort() must let all of them go (this block would hang without it).
    fqhost::AlignedPieces src(nullptr, argv[1], piece, 3);
    std::mutex fetch;
    std::atomic<int> held{0}, gone{0};
    auto hoard = [&] {
      struct Gone {
        std::atomic<int>& g;
        ~Gone() { ++g; }
      } on_return{gone};
      fqhost::Piece p;
      for (;;) {
        std::lock_guard<std::mutex> lk(fetch);
        if (!src.next(&p)) return;
        ++held;  // never released
        if (p.final) return;
      }
    };
    std::thread a(hoard), b(hoard);
    // (until three pieces are held - or the file has no more to give: a hoarder that got the last piece has returned,
    // and waiting out the two seconds for every small file was most of the sanitizer tests' time)
    for (int spin = 0; spin < 2000 && held.load() < 3 && gone.load() == 0; ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    src.abort();
    a.join();
    b.join();
  }
  {
    fqhost::AlignedPieces src(nullptr, argv[1], piece, 2 * n_cons + 2);
    auto work = [&] {
      fqhost::Piece p;
      for (;;) {
        {
          std::lock_guard<std::mutex> lk(mu);  // (the program fetches under a lock too)
          if (!src.next(&p)) return;
        }
        Seen s{std::string(p.data, p.size), p.first_record, p.final};
        {
          std::lock_guard<std::mutex> lk(mu);
          seen.emplace(p.seq, std::move(s));
        }
        src.release(p);
        if (p.final) return;
      }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < n_cons; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
  size_t off = 0;
  uint64_t lines = 0, k = 0;
  for (auto& kv : seen) {
    const Seen& s = kv.second;
    if (kv.first != k) return printf("piece %llu missing\n", (unsigned long long)k), 1;
    if (whole.compare(off, s.bytes.size(), s.bytes) != 0 || off + s.bytes.size() > whole.size())
      return printf("piece %llu differs from the file at %zu\n", (unsigned long long)k, off), 1;
    if (lines % 4 || s.first_record != lines / 4)
      return printf("piece %llu: first_record %llu, lines before %llu\n", (unsigned long long)k,
                    (unsigned long long)s.first_record, (unsigned long long)lines), 1;
    uint64_t nl = 0;
    for (char c : s.bytes) nl += c == '\n';
    const bool last = k + 1 == seen.size();
    if (s.final != last) return printf("piece %llu: final flag\n", (unsigned long long)k), 1;
    if (!last && (nl % 4 || s.bytes.empty() || s.bytes.back() != '\n'))
      return printf("piece %llu: %llu lines, not record-aligned\n", (unsigned long long)k, (unsigned long long)nl), 1;
    lines += nl;
    off += s.bytes.size();
    ++k;
  }
  if (off != whole.size()) return printf("pieces cover %zu of %zu bytes\n", off, whole.size()), 1;
  printf("%llu %zu ok\n", (unsigned long long)k, off);
  return 0;
}
