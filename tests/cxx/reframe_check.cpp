// TEST DRIVER: the host stager in re-framing mode (fastq_utils_amd/host/fq_input.h + fq_reframe.h: input cut at the
// reference's gzgets limits while it is read).  No GPU: the two library calls it makes are malloc / free here.
// argv: file piece_bytes mode (0: pieces, a tail carried from the last newline; 2: the whole file at once); writes the
// bytes the consumer sees to stdout.  tests/test_reframe.py holds them against a restatement of the gzgets calls.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../fastq_utils_amd/host/fq_input.h"

extern "C" void* fqg_host_alloc(fqg_ctx*, size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void fqg_host_free(fqg_ctx*, void* p) { free(p); }

int main(int argc, char** argv) {
  if (argc < 4) return 9;
  const size_t piece = strtoull(argv[2], nullptr, 10);
  const int mode = atoi(argv[3]);
  fqhost::reframe_supported() = true;
  fqhost::Input in(nullptr, argv[1], piece);
  if (in.vflags() != FQG_VALIDATE_REFRAMED) return 8;
  while (mode == 2 ? in.next(true) : in.next()) {
    size_t use = in.size();
    if (!in.final()) {
      while (use > 0 && in.data()[use - 1] != '\n') --use;
      if (use == 0) use = in.size();
    }
    fwrite(in.data(), 1, use, stdout);
    if (!in.final()) in.carry_from(use);
  }
  return 0;
}
