// fastq_filterpair - drop-in for the reference program of the same name (reference src/fastq_filterpair.c:38-228):
// pair the reads of two FASTQ files by name and write paired1 / paired2 / unpaired.
//
// Same command line, stderr text and exit status; the three outputs hold the same records in the same
// order (gzip level 3 like the reference's "w3"; what a reader inflates is identical).  What runs where:
//   GPU    framing + validation + unique-name index of file 1 (fastq_index_readnames, src/fastq.c:396-439);
//          one lookup-and-delete per record of the other file with the serial loop's answers
//          (fqg_index_probe_delete); the ordered gather of the records of each output (fqg_records_gather)
//   host   reading / inflating the inputs, deflating the outputs, the order and wording of messages
// Each input is held as ONE image on the device (the gathers address records of the whole file).
#include <unistd.h>

#include <string>
#include <vector>

#include "fq_common.h"
#include "fq_parallel.h"

namespace {

constexpr unsigned kHashSize = 100000001u;  // src/fastq_filterpair.c:33

// The reference writes its three outputs through gzopen(path, "w3") / gzwrite on its one thread (fastq_new(path, FALSE,
// "w3") -> fastq_open, src/fastq.c:631-664) - 50 MB/s, which is all a run of this program would be.  Here the text
// of a batch is cut into blocks and every block becomes a gzip member on a core of its own (fq_parallel.h: what a reader
// inflates is the same bytes); the level is the reference's 3, zlib's default for "-" as gzdopen(stdout, "wb") has it.
using Out = fqhost::GzipMembers;
Out* open_out(const char* path) {
  Out* g = new Out();
  const bool to_stdout = path[0] == '-' && path[1] == '\0';
  if (!g->open(path, to_stdout ? Z_DEFAULT_COMPRESSION : 3)) {
    FQ_PRINT_ERROR("Unable to open %s", path);
    fqhost::leave(kExitParams);
  }
  return g;
}

// records `list` of `frame`, in that order, appended to the output
void emit(const fqg_frame* frame, const std::vector<uint64_t>& list, Out* out, std::vector<char>& host) {
  if (list.empty()) return;
  uint64_t bytes = 0;
  LIB(fqg_records_gather(g_ctx, frame, list.data(), list.size(), &bytes));
  if (host.size() < bytes) host.resize(bytes);
  LIB(fqg_records_gather_output(g_ctx, host.data(), bytes));
  if (!out->write(host.data(), bytes)) {
    FQ_PRINT_ERROR("%s.\n", out->error().c_str());  // GZ_WRITE's gzerror() text, src/fastq.c:211-235
    fqhost::leave(kExitSys);
  }
}

void close_out(Out* g) {  // fastq_destroy -> fastq_close (src/fastq.c:615-629)
  if (!g->close()) {
    FQ_PRINT_ERROR("unable to close file descriptor");
    fqhost::leave(kExitSys);
  }
  delete g;
}

}  // namespace

int main(int argc, char** argv) {
  fqhost::install_counted_output(argv);  // (fq_respawn.h: a run that starts over on input cut at the gzgets limits prints nothing twice)
  fprintf(stderr, "fastq_utils %s\n", "0.25.3");  // fastq_print_version
  if (argc != 6 && argc != 7) {
    fprintf(stderr, "Usage: filterpair fastq1 fastq2 paired1 paired2 unpaired [sorted]\n");
    fqhost::leave(kExitParams);
  }
  fprintf(stderr, "%d", argc);
  const char* dev = getenv("FQGPU_DEVICE");
  int rc = fqg_open(dev ? atoi(dev) : 0, &g_ctx);
  if (rc != 0) {
    FQ_PRINT_ERROR("no usable MI355X device (fqg_open: %d); this program has no CPU path", rc);
    fqhost::leave(kExitSys);
  }
  const char *path1 = argv[1], *path2 = argv[2];
  Input in1(g_ctx, path1, piece_bytes());
  Input in2(g_ctx, path2, piece_bytes());
  const bool sorted = argc == 7 && !strcmp(argv[6], "sorted");
  fprintf(stderr, "HASHSIZE=%u\n", kHashSize);
  if (sorted) fprintf(stderr, "Assuming sorted fastq files\n");

  Stats S1, S2;
  LIB(fqg_acc_create(g_ctx, &S1.acc1));
  LIB(fqg_acc_create(g_ctx, &S2.acc1));
  unsigned long index_mem = 0;
  IndexedFile F1, F2;
  fprintf(stderr, "Scanning and indexing all reads from %s\n", path1);
  run_index_input(in1, path1, 1, S1, F1, true);
  fprintf(stderr, "Scanning complete.\n");
  index_mem += F1.index_mem;
  fprintf(stderr, "Reads indexed: %llu\n", (unsigned long long)F1.entries);
  fprintf(stderr, "Memory used in indexing: %ld MB\n", (long)(index_mem / 1024 / 1024));
  Out *w1 = open_out(argv[3]), *w2 = open_out(argv[4]), *w3 = open_out(argv[5]);
  unsigned long up2 = 0, paired = 0;
  std::vector<char> host;
  const fqg_frame* frame1 = fqg_index_frame(F1.index, 0);  // (null for an empty file)

  if (sorted) {
    // src/fastq_filterpair.c:96-148: index file 2 too, then every file against the other file's index
    fprintf(stderr, "Scanning and indexing all reads from %s\n", path2);
    run_index_input(in2, path2, 1, S2, F2, true);
    fprintf(stderr, "Scanning complete.\n");
    index_mem += F2.index_mem;
    fprintf(stderr, "Reads indexed: %llu\n", (unsigned long long)F2.entries);
    fprintf(stderr, "Memory used in indexing: %ld MB\n", (long)(index_mem / 1024 / 1024));
    const fqg_frame* frame2 = fqg_index_frame(F2.index, 0);
    struct Side {
      const char* path;
      const fqg_frame* frame;
      fqg_file_state st;
      fqg_index* other;
      uint64_t n;
      Out* pair_out;
      bool counts;
    } sides[2] = {{path1, frame1, F1.st, F2.index, F1.n_records, w1, true},
                  {path2, frame2, F2.st, F1.index, F2.n_records, w2, false}};
    for (const Side& sd : sides) {
      fprintf(stderr, "Filtering %s...\n", sd.path);
      std::vector<uint64_t> match(sd.n ? sd.n : 1), yes, no;
      if (sd.n) {
        LIB(fqg_frame_make_current(g_ctx, sd.frame));
        fqg_index_result ir;
        LIB(fqg_index_probe_delete(g_ctx, sd.other, &sd.st, match.data(), &ir));
        for (uint64_t k = 0; k < sd.n; ++k) (match[k] < FQG_MATCH_WRONG_HEADER ? yes : no).push_back(k);
      }
      // PRINT_READS_PROCESSED(fd->cline/4, 10000) with cline = 1 after fastq_rewind: (1 + 4j) / 4 = j
      ticker(1, sd.n, 10000);
      if (sd.counts) paired += yes.size();
      up2 += no.size();
      // the unpaired file interleaves nothing here: all of this file's singletons in order
      emit(sd.frame, yes, sd.pair_out, host);
      emit(sd.frame, no, w3, host);
    }
  } else {
    // src/fastq_filterpair.c:149-216
    fprintf(stderr, "Processing %s\n", path2);
    fflush(stderr);
    in2.next(true);
    Probe pr2;
    probe_piece(pr2, in2.data(), in2.size(), 1);
    fqg_validate_result r2;
    LIB(fqg_validate(g_ctx, nullptr, in2.data(), in2.size(), FQG_MEM_HOST, 1, &pr2.st,
                     FQG_VALIDATE_FRAME_ONLY | FQG_VALIDATE_NO_STATS | FQG_VALIDATE_NAMES | in2.vflags(), &r2));
    if (r2.code == FQG_E_LINE_TOO_LONG) fail_too_long(path2, r2.record);
    const uint64_t n2 = r2.n_records;
    std::vector<uint64_t> match(n2 ? n2 : 1), p1, p2, u2;
    fqg_index_result ir{};
    fqg_frame* frame2 = nullptr;
    if (n2) {
      LIB(fqg_index_probe_delete(g_ctx, F1.index, &pr2.st, match.data(), &ir));
      LIB(fqg_frame_retain(g_ctx, &frame2));
    }
    uint64_t stop = n2;  // fastq_get_readname exits at the first header without '@' (src/fastq.c:448)
    for (uint64_t k = 0; k < n2; ++k)
      if (match[k] == FQG_MATCH_WRONG_HEADER) {
        stop = k;
        break;
      }
    if (n2 && !(stop == 0)) print_probe(pr2);  // the first record's name fixes fd2's format (printed once)
    // what fd1's read position would be: after fastq_rewind at 0, after a copy behind the copied record
    std::vector<fqg_record> rec1(F1.n_records ? F1.n_records : 1);
    if (F1.n_records) {
      LIB(fqg_frame_make_current(g_ctx, frame1));
      LIB(fqg_frame_records(g_ctx, 0, F1.n_records, rec1.data(), FQG_MEM_HOST));
    }
    auto end_of = [&](uint64_t g) {
      const fqg_record& d = rec1[g];
      return d.offset + d.hdr1_len + d.seq_len + d.hdr2_len + d.qual_len;
    };
    uint64_t cur = 0, next1 = 0;
    unsigned long ctr_seek = 0, ctr_noseek = 0;
    for (uint64_t k = 0; k < stop; ++k) {
      if (match[k] < FQG_MATCH_WRONG_HEADER) {
        ++paired;
        p2.push_back(k);
        p1.push_back(match[k]);
        // fastq_quick_copy_entry (src/fastq.c:125-157): seek unless the file already stands there
        if (cur != rec1[match[k]].offset) ++ctr_seek;
        else ++ctr_noseek;
        fprintf(stderr, "%lu / %lu\n", ctr_seek, ctr_noseek);
        cur = end_of(match[k]);
        next1 = match[k] + 1;
      } else {
        ++up2;
        u2.push_back(k);
      }
      if ((k + 1) % 10000 == 0) ticker(k + 1, k + 1, 10000);
    }
    emit(frame2, p2, w2, host);
    emit(frame1, p1, w1, host);
    emit(frame2, u2, w3, host);
    if (stop < n2) {
      const RecordText t = locate_record(in2.data(), in2.size(), stop);
      fail_wrong_header(path2, 4 * (stop + 1), t.l[0]);
    }
    if (r2.tail_lines > 0) fail_truncated(path2, 4 * n2);
    fprintf(stderr, "\n");
    const uint64_t left = F1.entries - paired;
    fprintf(stderr, "Recording %llu unpaired reads from %s\n", (unsigned long long)left, path1);
    fflush(stderr);
    // src/fastq_filterpair.c:196-216: file 1 is read on from where the last copy left it, while entries remain
    std::vector<uint64_t> u1;
    if (left && F1.n_records) {
      std::vector<uint8_t> alive(F1.n_records);
      LIB(fqg_index_alive(g_ctx, F1.index, alive.data(), alive.size()));
      uint64_t remaining = left, j = 0;
      for (uint64_t g = next1; g < F1.n_records && remaining; ++g) {
        ++j;
        if (alive[g]) {
          u1.push_back(g);
          --remaining;
        }
        if (j % 100000 == 0) ticker(j, j, 100000);  // cline = 1 + 4j after the rewind
      }
    }
    emit(frame1, u1, w3, host);
    fprintf(stderr, "Unpaired from %s: %llu\n", path1, (unsigned long long)left);
    fprintf(stderr, "Unpaired from %s: %ld\n", path2, (long)up2);
    if (frame2) fqg_frame_release(frame2);
  }
  fprintf(stderr, "\n");
  fprintf(stderr, "Paired: %ld\n", (long)paired);
  close_out(w1);
  close_out(w2);
  close_out(w3);
  if (paired == 0) {
    fprintf(stderr, "!!!WARNING!!! 0 paired reads! are the headers ok?\n");
    fqhost::leave(kExitFormat);
  }
  fqhost::leave(0);
}
