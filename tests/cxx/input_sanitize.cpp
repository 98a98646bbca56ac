// TEST DRIVER: the host stager of the drop-in programs (fastq_utils_amd/host/fq_input.h: reader thread, ring of
// slots, carried tails, parallel pread) under -fsanitize=address,undefined and -fsanitize=thread
// (tests/test_sanitizers.py).  The two library calls it makes (pinned allocation) are plain malloc/free here: no GPU.
// argv: file piece_bytes carry_mode; prints the bytes it saw as a checksum + length, which must equal the file's.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../fastq_utils_amd/host/fq_input.h"

extern "C" void* fqg_host_alloc(fqg_ctx*, size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void fqg_host_free(fqg_ctx*, void* p) { free(p); }

int main(int argc, char** argv) {
  if (argc < 4) return 9;
  const size_t piece = strtoull(argv[2], nullptr, 10);
  const int mode = atoi(argv[3]);  // 0: consume everything; 1: leave a tail after the last '\n'; 2: whole file at once
  fqhost::Input in(nullptr, argv[1], piece);
  uint64_t sum = 1469598103934665603ull, total = 0;
  auto eat = [&](const char* p, size_t n) {
    for (size_t i = 0; i < n; ++i) sum = (sum ^ (unsigned char)p[i]) * 1099511628211ull;
    total += n;
  };
  while (mode == 2 ? in.next(true) : in.next()) {
    size_t use = in.size();
    if (mode == 1 && !in.final()) {
      while (use > 0 && in.data()[use - 1] != '\n') --use;  // up to the last complete line
      if (use == 0) use = in.size();
    }
    eat(in.data(), use);
    if (!in.final()) in.carry_from(use);
  }
  printf("%llu %llu\n", (unsigned long long)total, (unsigned long long)sum);
  return 0;
}
