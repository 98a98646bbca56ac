// fastq_filter_n - drop-in for the reference program (reference src/fastq_filter_n.c:33-95): print to
// stdout the records whose sequence holds at most read_len * n / 100 N characters (-n 0: none).
// Options, messages, exit codes as the reference; the record loop runs on the GPU
// (fqg_records_filter, FQG_FILTER_N).
#include <unistd.h>

#include "fq_filter_run.h"

using namespace fqhost;

int main(int argc, char** argv) {
  fqhost::install_counted_output(argv);  // (fq_respawn.h: a run that starts over on input cut at the gzgets limits prints nothing twice)
  int nopt = 0, c;
  opterr = 0;
  fprintf(stderr, "fastq_utils %s\n", "0.25.3");
  unsigned max_n = 0;
  while ((c = getopt(argc, argv, "n:")) != -1) switch (c) {
      case 'n':
        max_n = (unsigned)atoi(optarg);
        if (max_n > 100) max_n = 100;
        nopt += 2;
        break;
      default:
        ++nopt;
        FQ_PRINT_ERROR("Option -%c invalid", optopt);
        fqhost::leave(kExitParams);
    }
  if (argc - nopt < 2 || argc - nopt > 3) {
    FQ_PRINT_ERROR("Usage: fastq_filter_n [ -n 0 ] fastq1");
    fqhost::leave(kExitParams);
  }
  if (max_n > 0) fprintf(stderr, "Discard reads with more than %d%% of Ns\n", max_n);
  else fprintf(stderr, "Discard reads with at least one N\n");
  const char* path = argv[nopt + 1];

  fqg_ctx* ctx = nullptr;
  const char* dev = getenv("FQGPU_DEVICE");
  const int rc = fqg_open(dev ? atoi(dev) : 0, &ctx);
  if (rc != 0) {
    FQ_PRINT_ERROR("no usable MI355X GPU (fqg_open: %d); this build has no CPU path", rc);
    fqhost::leave(kExitSys);
  }
  fqg_filter_params fp;
  memset(&fp, 0, sizeof(fp));
  fp.mode = FQG_FILTER_N;
  fp.max_n_percent = max_n;
  run_filter(
      ctx, path, fp, [](const char* text, size_t n) { fwrite(text, 1, n, stdout); },
      [](unsigned long before, unsigned long after) {
        // PRINT_READS_PROCESSED(fd1->cline, 100000) after every record: cline = 4 * records
        for (unsigned long r = before / 25000 + 1; r * 25000 <= after; ++r) {
          fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", r * 100000);
          fflush(stderr);
        }
      });
  fflush(stdout);
  fqg_close(ctx);
  fqhost::leave(0);
}
