// fastq_trim_poly_at - drop-in for the reference program (reference src/fastq_trim_poly_at.c:123-233):
// cut a poly-A run off the 3' end or else a poly-T run off the 5' end of every read (N counts as
// either), drop reads that end up shorter than --min_len, write the rest gzipped.  Options,
// messages, exit codes as the reference; trim_poly_at (:77-119) and the length test run on the GPU
// (fqg_records_filter, FQG_FILTER_POLY_AT).
#include <getopt.h>

#include "fq_filter_run.h"
#include "fq_parallel.h"

using namespace fqhost;

#define FQ_PRINT_INFO(...)        \
  do {                            \
    fprintf(stderr, "INFO:");     \
    fprintf(stderr, __VA_ARGS__); \
    fprintf(stderr, "\n");        \
  } while (0)

static void print_usage() {  // src/fastq_trim_poly_at.c:121-133
  const char msg[] =
      "\n"
      "  --help       :print the usage\n"
      "  --file <filename> :fastq (optional gzipped) file name \n"
      "  --ofile <filename> : fastq file name where the processed reads will be written \n"
      "  --min_poly_at_len integer     : minimum length of poly-A|T sequence to remove.\n"
      "  --min_len integer     : minimum read length.\n";
  fprintf(stdout, "usage: fastq_trim_poly_at --file fastq_file --outfile out_file [optional parameters]");
  fprintf(stdout, "%s", msg);
}

int main(int argc, char** argv) {
  fqhost::install_counted_output(argv);  // (fq_respawn.h: a run that starts over on input cut at the gzgets limits prints nothing twice)
  const char* file = nullptr;
  const char* outfile = nullptr;
  long min_poly_at_len = 10, min_len = 10;
  static int help = 0;
  opterr = 0;
  fprintf(stderr, "fastq_utils %s\n", "0.25.3");
  static struct option long_options[] = {{"help", no_argument, &help, 1},
                                         {"min_poly_at_len", required_argument, 0, 'a'},
                                         {"file", required_argument, 0, 'b'},
                                         {"outfile", required_argument, 0, 'c'},
                                         {"min_len", required_argument, 0, 'd'},
                                         {0, 0, 0, 0}};
  while (1) {
    int option_index = 0;
    const int c = getopt_long(argc, argv, "a:b:c:d:", long_options, &option_index);
    if (c == -1) break;
    switch (c) {
      case 'a': min_poly_at_len = (int)atol(optarg); break;  // the field is an int (src/fastq_trim_poly_at.c:39)
      case 'b': file = optarg; break;
      case 'c': outfile = optarg; break;
      case 'd': min_len = atol(optarg); break;
      default: break;
    }
  }
  if (help) {
    print_usage();
    fqhost::leave(0);
  }
  FQ_PRINT_INFO("Validating options...");
  if (!file) {
    FQ_PRINT_ERROR("missing input file (--file)");
    fqhost::leave(kExitParams);
  }
  if (!outfile) {
    FQ_PRINT_ERROR("missing output file name (--outfile)");
    fqhost::leave(kExitParams);
  }
  FQ_PRINT_INFO("Options OK.");

  // the reference opens the input, then the output, before anything is read
  {
    gzFile probe = (file[0] == '-' && file[1] == 0) ? nullptr : gzopen(file, "r");
    if (!(file[0] == '-' && file[1] == 0)) {
      if (!probe) {
        FQ_PRINT_ERROR("Unable to open %s", file);
        fqhost::leave(kExitParams);
      }
      gzclose(probe);
    }
  }
  // gzip level 4 like the reference's "w4" ("-": its gzdopen(stdout, "wb"), default level); one member per
  // 4 MiB block, compressed on all cores (fq_parallel.h)
  GzipMembers out;
  if (!out.open(outfile, (outfile[0] == '-' && outfile[1] == 0) ? Z_DEFAULT_COMPRESSION : 4)) {
    FQ_PRINT_ERROR("Unable to open %s", outfile);
    fqhost::leave(kExitParams);
  }

  fqg_ctx* ctx = nullptr;
  const char* dev = getenv("FQGPU_DEVICE");
  const int rc = fqg_open(dev ? atoi(dev) : 0, &ctx);
  if (rc != 0) {
    FQ_PRINT_ERROR("no usable MI355X GPU (fqg_open: %d); this build has no CPU path", rc);
    fqhost::leave(kExitSys);
  }
  fqg_filter_params fp;
  memset(&fp, 0, sizeof(fp));
  fp.mode = FQG_FILTER_POLY_AT;
  fp.min_poly_at_len = min_poly_at_len;
  fp.min_len = min_len;
  const FilterTotals t = run_filter(
      ctx, file, fp,
      [&](const char* text, size_t n) {
        if (!out.write(text, n)) {
          FQ_PRINT_ERROR("%s.\n", out.error().c_str());  // GZ_WRITE's gzerror() text, src/fastq.c:211-235
          fqhost::leave(kExitSys);
        }
      },
      [](unsigned long before, unsigned long after) {
        // PRINT_READS_PROCESSED(fdi->cline / 4, 100000) after every record
        for (unsigned long c = (before / 100000 + 1) * 100000; c <= after; c += 100000) {
          fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", c);
          fflush(stderr);
        }
      });
  FQ_PRINT_INFO("Reads processed: %ld", (long)t.processed);
  FQ_PRINT_INFO("Reads trimmed: %ld", (long)t.trimmed);
  FQ_PRINT_INFO("Reads discarded: %ld", (long)t.discarded);
  if (!out.close()) {
    FQ_PRINT_ERROR("unable to close file descriptor");
    fqhost::leave(kExitSys);
  }
  fqg_close(ctx);
  fqhost::leave(0);
}
