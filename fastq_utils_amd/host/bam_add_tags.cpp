// bam_add_tags - drop-in for the reference program of the same name (reference src/bam_add_tags.c): the barcodes that
// fastq_pre_barcodes wrote into the read names become aux tags of the alignments (RX|UB, CR, BC, and with --tx the
// reference name tx and its gene GX).  The alignment loop (:250-294) is one bulk call into libfqgpu.so
// (fqg_bam_add_tags, include/fqg.h).
//
// Same command line, same stderr text, same exit status; the output BAM inflates to the same bytes (BGZF block
// boundaries and compressed bytes are zlib's business, not the format's).
//   host   option parsing (getopt_long with the reference's table), BGZF inflate / deflate on all cores, the header
//          (written as read), the transcript -> gene map (:203-232) resolved once per reference of the header
//   GPU    everything per alignment: get_barcodes on the name, the new tags, the rewritten record stream
// There is no CPU path for the record work: without a GPU the program fails before it reads the input.
#include "fq_parallel.h"
#include <errno.h>
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fqg.h"

namespace {

const char kVersion[] = "0.25.3";
const char kUsage[] =
    "Usage: bam_add_tags --inbam <in.bam> --outbam <out.bam or - for stdout> [--tx] [--tx2gx map_file_gene_2_trans.tsv]";
constexpr size_t kMaxFeatLen = 50;  // MAX_FEAT_LEN, src/bam_add_tags.c:36
fqg_ctx* g_ctx = nullptr;

// How the program leaves: with everything it wrote flushed, and WITHOUT exit()'s hooks - the HIP runtime tears itself
// down in one of them, and now and then that ended a run that had printed all it had to print with a segmentation
// fault (status 139 instead of 0: seen once in 300 runs of the GPU suite).  The other drop-in programs leave the same way.
[[noreturn]] static void leave(int code) {
  fflush(nullptr);
  if (getenv("FQGPU_PLAIN_EXIT")) exit(code);  // (tools/exit_stress.py: does the process survive exit()'s hooks?)
  _exit(code);
}

#define PRINT_ERROR(...)             \
  do {                               \
    fprintf(stderr, "\nERROR: ");    \
    fprintf(stderr, __VA_ARGS__);    \
    fprintf(stderr, "\n");           \
  } while (0)

void print_usage(int error) {  // :101-108
  if (error > 0) {
    PRINT_ERROR("%s", kUsage);
    leave(error);
  }
  fprintf(stderr, "%s\n", kUsage);
}

bool read_all(FILE* f, std::vector<uint8_t>& raw) {
  uint8_t buf[1 << 16];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) raw.insert(raw.end(), buf, buf + k);
  return !ferror(f);
}

}  // namespace

int main(int argc, char* argv[]) {
  char *inbam_file = nullptr, *outbam_file = nullptr, *map_file = nullptr;
  static int verbose = 0, help = 0, tx_tag = 0, tenx = 0;
  static struct option long_options[] = {  // :146-155
      {"verbose", no_argument, &verbose, 1},
      {"tx", no_argument, &tx_tag, 1},
      {"help", no_argument, &help, 1},
      {"inbam", required_argument, 0, 'i'},
      {"outbam", required_argument, 0, 'o'},
      {"tx_2_gx", required_argument, 0, 'm'},
      {"10x", no_argument, &tenx, 1},
      {0, 0, 0, 0}};
  for (;;) {
    int option_index = 0;
    const int c = getopt_long(argc, argv, "i:o:m:hX", long_options, &option_index);
    if (c == -1) break;
    switch (c) {
      case 'i': inbam_file = optarg; break;
      case 'o': outbam_file = optarg; break;
      case 'm': map_file = optarg; break;
      case 'h': help = 1; break;
      case 'X': tenx = 1; break;
      default: break;
    }
  }
  if (help) {
    print_usage(0);
    leave(0);
  }
  if (inbam_file == nullptr) print_usage(1);
  if (outbam_file == nullptr) print_usage(1);
  if (!tx_tag && map_file != nullptr) {
    PRINT_ERROR("missing  --tx when --tx_2_gx is provided\n");
    print_usage(1);  // PARAMS_ERROR_EXIT_STATUS
  }
  // both files are opened before either is looked at (:189-199): the output exists even when the input does not
  FILE* in = strcmp(inbam_file, "-") ? fopen(inbam_file, "rb") : stdin;
  if (!in) fprintf(stderr, "open: %s\n", strerror(errno));  // bgzf.c reports through perror("open")
  const bool out2stdout = strcmp(outbam_file, "-") == 0;
  FILE* out = out2stdout ? stdout : fopen(outbam_file, "wb");
  if (!in) {
    PRINT_ERROR("Failed to open BAM file %s", inbam_file);
    leave(1);
  }
  if (!out) {
    PRINT_ERROR("Failed to open BAM file %s", outbam_file);
    leave(1);
  }
  // gene <tab> transcript; the first line of a transcript wins (hash.c:161-184 appends, get_gene returns the first match)
  std::unordered_map<std::string, std::string> t2g;
  if (map_file != nullptr) {
    FILE* map_fd = fopen(map_file, "r");
    if (!map_fd) {
      PRINT_ERROR("Failed to open file %s", map_file);
      leave(1);
    }
    unsigned long long n_entries = 0;
    char buf[1000];
    while (!feof(map_fd)) {
      char* s = fgets(buf, 1000, map_fd);
      if (s == nullptr || s[0] == '\0') continue;
      char* gx = strtok(s, "\t\n");
      char* tx = strtok(nullptr, "\t\n");
      if (gx == nullptr || tx == nullptr) {
        PRINT_ERROR("Failed to find the gene and transcript ids in %s\n", s);
        leave(1);
      }
      if (strlen(gx) >= kMaxFeatLen || strlen(tx) >= kMaxFeatLen) {
        PRINT_ERROR("%s: an id of %zu characters or more (the reference copies ids into %zu-byte fields)", map_file,
                    kMaxFeatLen, kMaxFeatLen);
        leave(2);
      }
      t2g.emplace(tx, gx);
      ++n_entries;
    }
    fclose(map_fd);
    fprintf(stderr, "unique gene/transcript pairs %llu\n", n_entries);
  }

  int rc = fqg_open(0, &g_ctx);
  if (rc != 0) {
    PRINT_ERROR("no usable MI355X device (fqg_open: %d); this program has no CPU path", rc);
    leave(2);
  }
  std::vector<uint8_t> raw, stream;
  if (!read_all(in, raw) || !fqhost::bgzf_inflate_parallel(raw, stream)) {
    PRINT_ERROR("%s is not a readable BGZF / BAM file", inbam_file);
    leave(2);
  }
  raw.clear();
  raw.shrink_to_fit();
  uint64_t n_rec = 0, used = 0;
  if (fqg_bam_index_records(stream.data(), stream.size(), nullptr, 0, &n_rec, &used) != 0) {
    PRINT_ERROR("%s is not a BAM file", inbam_file);
    leave(2);
  }
  std::vector<uint64_t> offsets(n_rec ? n_rec : 1);
  fqg_bam_index_records(stream.data(), stream.size(), offsets.data(), n_rec, &n_rec, &used);
  const uint64_t header_end = n_rec ? offsets[0] : used;
  if (header_end > stream.size()) {
    PRINT_ERROR("%s is not a BAM file", inbam_file);
    leave(2);
  }

  // the references of the header (bam_header_read): names as C strings
  std::string names;
  std::vector<uint32_t> tx_off, tx_len, gx_off, gx_len;
  {
    auto rd32 = [&](uint64_t p) {
      int32_t v;
      memcpy(&v, &stream[p], 4);
      return v;
    };
    uint64_t p = 8 + (uint64_t)(uint32_t)rd32(4);
    const int32_t n_ref = rd32(p);
    p += 4;
    for (int32_t i = 0; i < n_ref; ++i) {
      // (fqg_bam_index_records vouches for the walk as a whole; each name is checked again where it is used)
      if (p + 8 > stream.size() || p + 8 + (uint64_t)(uint32_t)rd32(p) > stream.size()) {
        PRINT_ERROR("%s: truncated BAM header (reference %d)", inbam_file, i);
        leave(2);
      }
      const uint32_t l_name = (uint32_t)rd32(p);
      const char* nm = (const char*)&stream[p + 4];
      const size_t len = strnlen(nm, l_name);
      if (len + 1 != l_name) {
        PRINT_ERROR("%s: reference %d has a name that is not one C string (bam_header_write would not write it back as read)",
                    inbam_file, i);
        leave(2);
      }
      tx_off.push_back((uint32_t)names.size());
      tx_len.push_back((uint32_t)len);
      names.append(nm, len);
      names.push_back('\0');
      auto g = map_file ? t2g.find(std::string(nm, len)) : t2g.end();
      if (g == t2g.end()) {
        gx_off.push_back(0);
        gx_len.push_back(FQG_NO_GENE);
      } else {
        gx_off.push_back((uint32_t)names.size());
        gx_len.push_back((uint32_t)g->second.size());
        names += g->second;
        names.push_back('\0');
      }
      p += 4 + (uint64_t)l_name + 4;
    }
  }
  if (!out2stdout) {
    fprintf(stderr, "bam_add_tags version %s\n", kVersion);
    fprintf(stderr, "Processing %s\n", inbam_file);
  }
  fqg_bam_tags_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.tenx = tenx;
  prm.tx_tag = tx_tag;
  prm.n_targets = (uint32_t)tx_off.size();
  prm.tx_off = tx_off.data();
  prm.tx_len = tx_len.data();
  prm.gx_off = gx_off.data();
  prm.gx_len = gx_len.data();
  prm.names = names.data();
  prm.names_bytes = names.size();
  fqg_bam_tags_result res;
  rc = fqg_bam_add_tags(g_ctx, stream.data(), used, FQG_MEM_HOST, offsets.data(), n_rec, &prm, &res);
  if (rc != 0) {
    PRINT_ERROR("GPU library failure in fqg_bam_add_tags (%d): %s", rc, fqg_last_error(g_ctx));
    leave(2);
  }
  if (res.code == FQG_E_TAGS_NAME) {
    PRINT_ERROR("%s: alignment %llu: a barcode in the read name runs to the end of the record or has %d characters or more; the "
                "reference reads and writes memory it does not own there, this program refuses the file",
                inbam_file, (unsigned long long)res.record + 1, 50);
    leave(2);
  }
  if (res.code == FQG_E_TAGS_TID) {
    PRINT_ERROR("%s: alignment %llu: reference id beyond the header's references", inbam_file, (unsigned long long)res.record + 1);
    leave(2);
  }
  std::vector<uint8_t> recs(res.out_bytes ? res.out_bytes : 1);
  rc = fqg_bam_add_tags_output(g_ctx, recs.data(), res.out_bytes);
  if (rc != 0) {
    PRINT_ERROR("GPU library failure in fqg_bam_add_tags_output (%d): %s", rc, fqg_last_error(g_ctx));
    leave(2);
  }
  std::vector<uint8_t> bgzf;
  if (!fqhost::bgzf_deflate_parallel({{stream.data(), (size_t)header_end}, {recs.data(), (size_t)res.out_bytes}}, Z_DEFAULT_COMPRESSION,
                                     bgzf) ||
      fwrite(bgzf.data(), 1, bgzf.size(), out) != bgzf.size() || fflush(out) != 0) {
    PRINT_ERROR("Failed to write %s", outbam_file);
    leave(2);
  }
  if (!out2stdout) {
    fclose(out);
    fprintf(stderr, "Processing %s complete\n", inbam_file);
  }
  fqg_close(g_ctx);
  leave(0);
}
