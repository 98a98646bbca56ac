// bam_umi_count - drop-in for the reference program of the same name (reference
// src/bam_umi_count.c), with the alignment loop and the output decisions replaced by one bulk call
// into libfqgpu.so (fqg_umi_count, include/fqg.h).
//
// Same command line, same stderr text, same output files (.mtx, _rows, _cols), same exit status.
//   host   option parsing (getopt_long with the reference's table), whitelist files, BGZF inflate
//          (zlib; libbam does it inside bam_read1), the block_size walk that finds the records,
//          message text, file writing
//   GPU    everything per alignment: filters, aux tags, barcode packing, the label maps, the
//          (cell, feature, UMI) set, float32 counters, which lines each cell prints
// There is no CPU path for the record work: without a GPU the program fails at start-up.
#include "fq_parallel.h"
#include <errno.h>
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <string>
#include <vector>

#include "../../include/fqg.h"
#include "umi_multi.h"

namespace {

const char kVersion[] = "0.25.3";
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

[[noreturn]] void die_lib(const char* what, int rc) {
  PRINT_ERROR("GPU library failure in %s (%d): %s", what, rc, g_ctx ? fqg_last_error(g_ctx) : "no context");
  leave(2);
}
#define LIB(call)                         \
  do {                                    \
    int rc__ = (call);                    \
    if (rc__ != 0) die_lib(#call, rc__);  \
  } while (0)

void print_usage(int exit_status) {  // src/bam_umi_count.c:724-727
  PRINT_ERROR(
      "Usage: bam_umi_count --bam in.bam --ucounts output_filename [--min_reads 0] [--min_umis 0] "
      "[--uniq_mapped|--multi_mapped]  [--dump filename] [--tag gx|tx] [--known_umi file_one_umi_per_line] "
      "[--ucounts_MM |--ucounts_tsv] [--ucounts_MM|--ucounts_tsv] [--ignore_sample] [--cell_suffix suffix] "
      "[--max_cells number] [--max_feat number] [--feat_cell number] [--cell_tag tag] [--sorted_by_cell] [--10x]");
  if (exit_status >= 0) leave(exit_status);
}

// load_whitelist (:543-579): packed value of every line, in file order
std::vector<uint64_t> load_whitelist(const char* file) {
  FILE* fd = fopen(file, "r");
  if (!fd) {
    PRINT_ERROR("Failed to open file %s", file);
    leave(1);
  }
  fprintf(stderr, "Loading whitelist from %s\n", file);
  std::vector<uint64_t> v;
  char buf[200];
  while (!feof(fd)) {
    char* l = fgets(buf, 200, fd);
    if (l == nullptr || l[0] == '\0') continue;
    v.push_back(fqg_pack_barcode(l));
  }
  fclose(fd);
  fprintf(stderr, "Loading whitelist from %s...done.\n", file);
  return v;
}

// the whole file, BGZF members inflated back to back (SAM/BAM specification, section 4.1)
bool read_all(FILE* f, std::vector<uint8_t>& raw) {
  uint8_t buf[1 << 16];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) raw.insert(raw.end(), buf, buf + k);
  return !ferror(f);
}

void write_rows(const std::string& file, const std::vector<char>& names, uint64_t n) {  // write_map2fileL :271-290
  const std::string path = file + "_rows";
  FILE* fd = fopen(path.c_str(), "w+");
  if (!fd) {
    PRINT_ERROR("Failed to open file %s for writing", path.c_str());
    leave(1);
  }
  for (uint64_t i = 0; i < n; ++i) fprintf(fd, "%u\t%s\n", (unsigned)(i + 1), &names[i * 25]);
  fclose(fd);
}

void write_cols(const std::string& file, const std::vector<uint64_t>& cells, const char* suffix) {  // write_map2fileB :292-317
  const std::string path = file + "_cols";
  FILE* fd = fopen(path.c_str(), "w+");
  if (!fd) {
    PRINT_ERROR("Failed to open file %s for writing", path.c_str());
    leave(1);
  }
  char buf[24];
  for (size_t i = 0; i < cells.size(); ++i) {
    fqg_unpack_barcode(cells[i], buf);
    fprintf(fd, "%u\t%s%s\n", (unsigned)(i + 1), buf, suffix ? suffix : "");
  }
  fclose(fd);
}

}  // namespace

int main(int argc, char* argv[]) {
  unsigned min_num_reads = 0, min_num_umis = 0;
  char feat_tag[4] = "GX", cell_tag[4] = "CR";
  unsigned long max_features = 100000, max_cells = 1000000;
  char *bam_file = nullptr, *ucounts_file = nullptr, *rcounts_file = nullptr;
  char *known_umi_file = nullptr, *known_cells_file = nullptr, *cell_suffix = nullptr;
  static int bam_sorted_by_cell = 0, uniq_mapped_only = 0, verbose = 0, help = 0, ignore_sample = 0, tenx = 0;
  static struct option long_options[] = {  // src/bam_umi_count.c:767-790
      {"verbose", no_argument, &verbose, 1},
      {"multi_mapped", no_argument, &uniq_mapped_only, 0},
      {"uniq_mapped", no_argument, &uniq_mapped_only, 1},
      {"sorted_by_cell", no_argument, &bam_sorted_by_cell, 1},
      {"not_sorted_by_cell", no_argument, &bam_sorted_by_cell, 0},
      {"ignore_sample", no_argument, &ignore_sample, 1},
      {"help", no_argument, &help, 1},
      {"bam", required_argument, 0, 'b'},
      {"cell_suffix", required_argument, 0, 's'},
      {"known_umi", required_argument, 0, 'k'},
      {"known_cells", required_argument, 0, 'c'},
      {"ucounts", required_argument, 0, 'u'},
      {"rcounts", required_argument, 0, 'r'},
      {"tag", required_argument, 0, 'x'},
      {"cell_tag", required_argument, 0, 'X'},
      {"min_reads", required_argument, 0, 't'},
      {"min_umis", required_argument, 0, 'U'},
      {"max_cells", required_argument, 0, 'C'},
      {"max_feat", required_argument, 0, 'F'},
      {"feat_cell", required_argument, 0, 'T'},
      {"10x", no_argument, &tenx, 1},
      {0, 0, 0, 0}};
  ignore_sample = 1;        // :792-793
  bam_sorted_by_cell = 1;
  fprintf(stderr, "bam_umi_count version %sb\n", kVersion);
  while (true) {
    int option_index = 0;
    const int c = getopt_long(argc, argv, "F:T:C:b:U:u:r:t:x:c:s:hX:", long_options, &option_index);
    if (c == -1) break;
    switch (c) {
      case 'h': help = 1; break;
      case 'b': bam_file = optarg; break;
      case 'u': ucounts_file = optarg; break;
      case 'r': rcounts_file = optarg; break;
      case 's': cell_suffix = optarg; break;
      case 'k': known_umi_file = optarg; break;
      case 'c': known_cells_file = optarg; break;
      case 'x': strncpy(feat_tag, optarg, 3); feat_tag[3] = 0; break;
      case 'X': strncpy(cell_tag, optarg, 3); cell_tag[3] = 0; break;
      case 't': min_num_reads = (unsigned)atol(optarg); break;
      case 'U': min_num_umis = (unsigned)atol(optarg); break;
      case 'C': max_cells = (unsigned long)atol(optarg); break;
      case 'F': max_features = (unsigned long)atol(optarg); break;
      case 'T': break;
      default: break;
    }
  }
  if (help) print_usage(0);
  if (bam_file == nullptr) print_usage(1);
  if (ucounts_file == nullptr) print_usage(1);
  if (bam_sorted_by_cell) max_cells = 1;

  std::vector<uint64_t> kumi, kcells;
  if (known_umi_file) {
    kumi = load_whitelist(known_umi_file);
    fprintf(stderr, "UMIs whitelist %llu\n", (unsigned long long)kumi.size());
  }
  if (known_cells_file) {
    kcells = load_whitelist(known_cells_file);
    fprintf(stderr, "Cells whitelist %llu\n", (unsigned long long)kcells.size());
  }

  FILE* in = strcmp(bam_file, "-") ? fopen(bam_file, "rb") : stdin;
  if (!in) {
    fprintf(stderr, "open: %s\n", strerror(errno));  // bgzf.c reports through perror("open")
    PRINT_ERROR("Failed to open BAM file %s", bam_file);
    leave(1);
  }
  const char* umi_tag = tenx ? "UB" : "RX";
  fprintf(stderr, "@min_num_reads=%u\n", min_num_reads);
  fprintf(stderr, "@min_num_umis=%u\n", min_num_umis);
  fprintf(stderr, "@uniq mapped reads=%u\n", uniq_mapped_only);
  fprintf(stderr, "@sorted bam=%u\n", bam_sorted_by_cell);
  fprintf(stderr, "@tag=%s\n", feat_tag);
  fprintf(stderr, "@umi tag=%s\n", umi_tag);
  fprintf(stderr, "@unique counts file=%s\n", ucounts_file);
  if (cell_suffix) fprintf(stderr, "@cell_suffix=%s\n", cell_suffix);

  int rc = fqg_open(0, &g_ctx);
  if (rc != 0) {
    PRINT_ERROR("no usable MI355X device (fqg_open: %d); this program has no CPU path", rc);
    leave(2);
  }
  std::vector<uint8_t> raw, stream;
  if (!read_all(in, raw) || !fqhost::bgzf_inflate_parallel(raw, stream)) {
    PRINT_ERROR("%s is not a readable BGZF / BAM file", bam_file);
    leave(2);
  }
  raw.clear();
  raw.shrink_to_fit();
  fprintf(stderr, "Processing %s\n", bam_file);

  const char kHdr[] = "%%MatrixMarket matrix coordinate real general\n";  // the reference prints both percent signs
  FILE *counts_fd = nullptr, *rcounts_fd = nullptr;
  long header_loc = 0, rheader_loc = 0;
  auto mm_header = [&](const char* file, long* loc) {  // MM_header :708-722
    FILE* fd = fopen(file, "w+");
    if (!fd) {
      PRINT_ERROR("Failed to open file %s", file);
      leave(1);
    }
    fprintf(stderr, "Creating MM file %s...\n", file);
    fputs(kHdr, fd);
    *loc = ftell(fd);
    fprintf(fd, "%-10lu %-10lu %-15llu\n", 0L, 0L, 0LL);
    return fd;
  };
  if (bam_sorted_by_cell) {
    counts_fd = mm_header(ucounts_file, &header_loc);
    if (rcounts_file) rcounts_fd = mm_header(rcounts_file, &rheader_loc);
    fprintf(stderr, "Cells processed\n");
  }

  uint64_t n_rec = 0, used = 0;
  std::vector<uint64_t> offsets;
  if (fqg_bam_index_records(stream.data(), stream.size(), nullptr, 0, &n_rec, &used) != 0) {
    PRINT_ERROR("%s is not a BAM file", bam_file);
    leave(2);
  }
  offsets.resize(n_rec ? n_rec : 1);
  fqg_bam_index_records(stream.data(), stream.size(), offsets.data(), n_rec, &n_rec, &used);

  fqg_umi_params prm;
  memset(&prm, 0, sizeof(prm));
  memcpy(prm.feat_tag, feat_tag, 2);
  memcpy(prm.cell_tag, cell_tag, 2);
  memcpy(prm.umi_tag, umi_tag, 2);
  prm.sorted_by_cell = bam_sorted_by_cell;
  prm.uniq_mapped_only = uniq_mapped_only;
  prm.max_cells = (uint32_t)max_cells;
  prm.max_features = (uint32_t)max_features;
  prm.min_reads = min_num_reads;
  prm.min_umis = min_num_umis;
  static const uint64_t kNone = 0;
  if (known_umi_file) {
    prm.known_umis = kumi.empty() ? &kNone : kumi.data();
    prm.n_known_umis = kumi.size();
  }
  if (known_cells_file) {
    prm.known_cells = kcells.empty() ? &kNone : kcells.data();
    prm.n_known_cells = kcells.size();
  }
  fqg_umi_result res;
  prm.strict_set = getenv("FQGPU_UMI_STRICT_SET") ? 1 : 0;  // an extra: UMI sets as src/range_list.h documents them
  // FQGPU_DEVICES=0,1,..: a file that is read cell by cell goes over several GPUs by cell range (umi_multi.h).
  // Whatever that path does not take - a finding, a file that is not grouped by cell, a limit - runs on one device as
  // before: its words are the reference's.
  fqhost::UmiMultiResult multi;
  {
    std::vector<int> devs;
    if (const char* e = getenv("FQGPU_DEVICES"))
      for (const char* p = e; *p;) {
        char* q = nullptr;
        const long v = strtol(p, &q, 10);
        if (q == p) break;
        devs.push_back((int)v);
        p = *q == ',' ? q + 1 : q;
      }
    if (devs.size() > 1 && bam_sorted_by_cell && n_rec) {
      std::vector<fqg_ctx*> ctxs{g_ctx};
      for (size_t i = 1; i < devs.size(); ++i) {
        fqg_ctx* c = nullptr;
        if (fqg_open(devs[i], &c) != 0) {
          PRINT_ERROR("FQGPU_DEVICES: device %d is not a usable MI355X GPU", devs[i]);
          leave(2);
        }
        ctxs.push_back(c);
      }
      multi = fqhost::umi_count_multi(ctxs, stream, offsets, n_rec, used, prm);
      if (getenv("FQGPU_UMI_MULTI_DEBUG"))
        fprintf(stderr, "[umi multi] %s%s\n", multi.ok ? "counted over the devices" : "one device: ", multi.ok ? "" : multi.why.c_str());
      for (size_t i = 1; i < ctxs.size(); ++i) fqg_close(ctxs[i]);
    }
  }
  if (multi.ok) {
    memset(&res, 0, sizeof(res));
    res.n_alignments = multi.n_alignments;
    res.n_tags_found = multi.n_tags_found;
    res.n_umis_discarded = multi.n_umis_discarded;
    res.n_cells_discarded = multi.n_cells_discarded;
    res.n_features = multi.features.size();
    res.n_cells = multi.cells.size();
    for (int w = 0; w < 2; ++w) {
      res.n_entries[w] = multi.entries[w].size();
      res.total[w] = multi.total[w];
    }
    res.tot_reads = multi.tot_reads;
    res.tot_umi = multi.tot_umi;
    res.rl_replayed = multi.rl_replayed;
    res.rl_changed = multi.rl_changed;
    res.rl_undefined = multi.rl_undefined;
  } else
    LIB(fqg_umi_count(g_ctx, stream.data(), stream.size(), FQG_MEM_HOST, offsets.data(), n_rec, &prm, &res));
  if (res.rl_unresolved) {  // never silently different from the reference
    fprintf(stderr, "\nERROR: bam_umi_count: a UMI set could not be replayed within this build's limits\n");
    leave(2);
  }
  if (getenv("FQGPU_RL_DEBUG"))  // (where `undefined` is not 0 the reference's own counts depend on what its heap held: DESIGN.md 7.1)
    fprintf(stderr, "[rl] sets replayed %llu, alignments decided differently from a set %llu, reads of tree memory the reference never wrote: undefined %llu\n",
            (unsigned long long)res.rl_replayed, (unsigned long long)res.rl_changed, (unsigned long long)res.rl_undefined);

  // An alignment that is out of order in a file that is read cell by cell: the reference writes a cell's lines when the
  // cell changes (cell2MM, src/bam_umi_count.c:1009-1015) and meets that alignment BEFORE it writes the cell in front of
  // it (:1003-1006) - its files hold every cell but the last of the alignments in front, behind a header it never
  // comes back to.  The same here: those alignments are counted once more, alone, and the last of their cells left out.
  // (The limits of process_entry, :444-463, are not met by the reference in a state one could copy: it dies of a
  // signal before it says "Too many features" - DESIGN.md 7.1.)
  auto cells_in_front = [&] {
    fqg_umi_result part;
    if (!bam_sorted_by_cell || res.record == 0 ||
        fqg_umi_count(g_ctx, stream.data(), stream.size(), FQG_MEM_HOST, offsets.data(), res.record, &prm, &part) != 0 ||
        part.code != FQG_OK || part.rl_unresolved || part.n_cells < 2)
      return;
    const uint64_t written = part.n_cells - 1;
    for (uint64_t k = 10000; k <= written; k += 10000)
      fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b%-10llu", (unsigned long long)k);
    FILE* fds[2] = {counts_fd, rcounts_fd};
    for (int w = 0; w < 2; ++w) {
      if (!fds[w]) continue;
      std::vector<fqg_umi_entry> ent(part.n_entries[w]);
      LIB(fqg_umi_entries(g_ctx, w, ent.data(), ent.size()));
      for (const auto& e : ent)
        if (e.col <= written) fprintf(fds[w], "%u %u %u\n", e.row, e.col, e.value);
      fflush(fds[w]);
    }
  };
  const uint64_t alns_seen = res.code ? res.record + 1 : res.n_alignments;
  if (!bam_sorted_by_cell)
    for (uint64_t k = 100000; k <= alns_seen; k += 100000)
      fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%llu", (unsigned long long)k);
  switch (res.code) {
    case FQG_OK: break;
    case FQG_E_UMI_NOT_SORTED:
      cells_in_front();
      fprintf(stderr, "Error: The BAM file does not seem to be sorted by CR\n");
      leave(1);
    case FQG_E_UMI_FEATURE_NAME:
      fprintf(stderr, "bam_umi_count: src/bam_umi_count.c:1048: main: Assertion `len1+1 < FEAT_ID_MAX_LEN' failed.\n");
      fflush(nullptr);
      abort();
    case FQG_E_UMI_TOO_MANY_UMIS:
      PRINT_ERROR("Too many umi barcodes %u - please rerun and increase the maximum number of umis\n", (unsigned)res.aux);
      leave(1);
    case FQG_E_UMI_TOO_MANY_CELLS:
      PRINT_ERROR("Too many cells %u - please rerun and increase the cells using the --max_cells parameter\n", (unsigned)res.aux);
      leave(1);
    case FQG_E_UMI_TOO_MANY_FEATURES:
      PRINT_ERROR("Too many features %u - please rerun and increase the maximum number of features using the --max_feat parameter\n",
                  (unsigned)res.aux);
      leave(1);
    default:
      PRINT_ERROR("unexpected finding %d", res.code);
      leave(2);
  }
  if (bam_sorted_by_cell)
    for (uint64_t k = 10000; k <= res.n_cells; k += 10000)
      fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b%-10llu", (unsigned long long)k);
  fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b\n");
  fprintf(stderr, "Alignments processed: %llu\n", (unsigned long long)res.n_alignments);
  fprintf(stderr, "%s encountered  %llu times\n", feat_tag, (unsigned long long)res.n_tags_found);
  fprintf(stderr, "%lld UMIs discarded\n", (long long)res.n_umis_discarded);
  fprintf(stderr, "%lld cells discarded\n", (long long)res.n_cells_discarded);
  fprintf(stderr, "%u features\n", (unsigned)res.n_features);
  fprintf(stderr, "%u cells\n", (unsigned)res.n_cells);
  fprintf(stderr, "%u samples\n", 0u);
  fprintf(stderr, "%f total reads\n", (double)res.tot_reads);
  fprintf(stderr, "%f total UMI\n", (double)res.tot_umi);
  if (!res.n_tags_found) {
    fprintf(stderr, "ERROR: no valid alignments tagged with %s were found in %s.\n", feat_tag, bam_file);
    leave(1);
  }

  std::vector<char> names(res.n_features * 25 + 25);
  std::vector<uint64_t> cells(res.n_cells);
  std::vector<fqg_umi_entry> ent[2];
  if (multi.ok) {
    for (size_t i = 0; i < multi.features.size(); ++i) strncpy(&names[i * 25], multi.features[i].c_str(), 24);
    cells = multi.cells;
    ent[0] = multi.entries[0];
    ent[1] = multi.entries[1];
  } else {
    LIB(fqg_umi_features(g_ctx, names.data(), res.n_features));
    LIB(fqg_umi_cells(g_ctx, cells.data(), res.n_cells));
    for (int w = 0; w < 2; ++w) {
      ent[w].resize(res.n_entries[w]);
      LIB(fqg_umi_entries(g_ctx, w, ent[w].data(), ent[w].size()));
    }
  }
  auto put_lines = [](FILE* fd, const std::vector<fqg_umi_entry>& v) {
    for (const auto& e : v) fprintf(fd, "%u %u %u\n", e.row, e.col, e.value);
  };

  if (bam_sorted_by_cell) {
    // cell2MM wrote the lines while reading; the header is finished afterwards (:1093-1113)
    put_lines(counts_fd, ent[0]);
    fseek(counts_fd, header_loc, SEEK_SET);
    fprintf(counts_fd, "%-10u %-10u %-15llu", (unsigned)res.n_features, (unsigned)res.n_cells, (unsigned long long)res.total[0]);
    write_rows(ucounts_file, names, res.n_features);
    write_cols(ucounts_file, cells, cell_suffix);
    fclose(counts_fd);
    if (rcounts_fd) {
      put_lines(rcounts_fd, ent[1]);
      fseek(rcounts_fd, rheader_loc, SEEK_SET);
      fprintf(rcounts_fd, "%-10u %-10u %-15llu", (unsigned)res.n_features, (unsigned)res.n_cells, (unsigned long long)res.total[1]);
      write_rows(rcounts_file, names, res.n_features);
      write_cols(rcounts_file, cells, cell_suffix);
      fclose(rcounts_fd);
    }
    fqg_close(g_ctx);
    leave(0);
  }

  auto write2mm = [&](const char* file, int which) {  // write2MM :584-663
    FILE* fd = fopen(file, "w+");
    if (!fd) {
      PRINT_ERROR("Failed to open file %s", file);
      leave(1);
    }
    fprintf(stderr, "Saving MM file %s...\n", file);
    write_rows(file, names, res.n_features);
    write_cols(file, cells, cell_suffix);
    fputs(kHdr, fd);
    fprintf(fd, "%u %u ", (unsigned)res.n_features, (unsigned)res.n_cells);
    const long loc = ftell(fd);
    fprintf(fd, "%-15lu\n", 0L);
    put_lines(fd, ent[which]);
    if (ent[which].empty()) {
      fprintf(stderr, "ERROR: 0 quantified features.\n");
      leave(1);
    }
    fseek(fd, loc, SEEK_SET);
    fprintf(fd, "%-15llu", (unsigned long long)ent[which].size());
    fclose(fd);
    fprintf(stderr, "Saving MM file...done.\n");
    fprintf(stderr, "#cells/features: %llu\n", (unsigned long long)ent[which].size());
    fprintf(stderr, "#cells: %llu\n", 0ULL);
    fprintf(stderr, "#tot expr: %llu\n", (unsigned long long)res.total[which]);
  };
  write2mm(ucounts_file, 0);
  if (rcounts_file) write2mm(rcounts_file, 1);
  fqg_close(g_ctx);
  leave(0);
}
