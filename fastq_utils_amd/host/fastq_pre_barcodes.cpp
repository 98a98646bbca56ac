// fastq_pre_barcodes - drop-in for the reference program of the same name
// (reference src/fastq_pre_barcodes.c): same options, same stderr / stdout text, same output bytes.
//
//   host   option parsing, (gz) reading of up to five inputs into pinned pieces, lock-step
//          bookkeeping across pieces, gzip / stdout writing of what the GPU produced
//   GPU    framing of every input, read-name agreement, barcode extraction with the quality
//          filter, header tagging, slicing, and the assembly of the output text (FASTQ or SAM)
// There is no CPU path for the per-read work.
#include <getopt.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <map>
#include <set>
#include <functional>
#include <string>
#include <vector>

#include "fq_blocks.h"
#include "fq_input.h"
#include "fq_multi.h"
#include "fq_parallel.h"

// SURVEY 5 "metrics", as bin/fastq_info has it: FQGPU_JSON_METRICS=<file> writes the machine-readable twin of the
// "Reads processed / discarded" lines - the command line and both output streams stay the reference's
static const std::chrono::steady_clock::time_point g_pb_start = std::chrono::steady_clock::now();
static void pb_json_metrics(long processed, long discarded, size_t devices) {
  const char* jm = getenv("FQGPU_JSON_METRICS");
  if (!jm) return;
  FILE* jf = fopen(jm, "w");
  if (!jf) return;
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_pb_start).count();
  const unsigned long long bytes = fqhost::bytes_handed_out().load();
  fprintf(jf, "{\"program\": \"fastq_pre_barcodes\", \"reads\": %ld, \"discarded\": %ld, \"input_bytes\": %llu, \"seconds\": %.6f, "
              "\"Mreads_per_s\": %.3f, \"GB_per_s\": %.3f, \"devices\": %zu}\n",
          processed, discarded, bytes, secs, secs > 0 ? (double)processed / secs / 1e6 : 0.0, secs > 0 ? (double)bytes / secs / 1e9 : 0.0,
          devices ? devices : (size_t)1);
  fclose(jf);
}

using namespace fqhost;

namespace {

fqg_ctx* g_ctx = nullptr;
[[noreturn]] void die_lib(const char* what, int rc) {
  FQ_PRINT_ERROR("GPU library failure in %s (%d): %s", what, rc, g_ctx ? fqg_last_error(g_ctx) : "no context");
  fqhost::leave(kExitSys);
}
#define LIB(call)                        \
  do {                                   \
    int rc__ = (call);                   \
    if (rc__ != 0) die_lib(#call, rc__); \
  } while (0)
#define FQ_PRINT_INFO(...)        \
  do {                            \
    fprintf(stderr, "INFO:");     \
    fprintf(stderr, __VA_ARGS__); \
    fprintf(stderr, "\n");        \
  } while (0)

size_t piece_bytes() {
  const char* e = getenv("FQGPU_CHUNK_MB");
  size_t mb = e ? strtoull(e, nullptr, 10) : 512;
  if (mb < 1) mb = 1;
  return mb << 20;
}

enum { READ1 = 1, READ2 = 2, INDEX1 = 3, INDEX2 = 4, INDEX3 = 5 };

int read_index2read_idx(const char* s) {  // src/fastq_pre_barcodes.c:79-89
  if (!strcmp(s, "read1")) return READ1;
  if (!strcmp(s, "read2")) return READ2;
  if (!strcmp(s, "index1")) return INDEX1;
  if (!strcmp(s, "index2")) return INDEX2;
  if (!strcmp(s, "index3")) return INDEX3;
  FQ_PRINT_ERROR("invalid file reference %s (valid values are read1,read2, index1,index2,index3)\n", s);
  fqhost::leave(1);
}

void print_usage() {  // src/fastq_pre_barcodes.c:311-346
  const char msg[] =
      "  --verbose    :increase level of messages printed to stderr\n"
      "  --brief      :decrease level of messages printed to stderr\n"
      "  --help       :print the usage\n"
      "  --read1 <filename> :fastq (optional gzipped) file name \n"
      "  --read2 <filename> :fastq (optional gzipped) file name \n"
      "  --index1 <filename> :fastq (optional gzipped) file name \n"
      "  --index2 <filename> :fastq (optional gzipped) file name \n"
      "  --index3 <filename> :fastq (optional gzipped) file name \n"
      "  --phred_encoding (33|64) :phred encoding used in the input files\n"
      "  --min_qual [0-40]        :defines the minimum quality that all bases in the UMI, CELL or Sample should "
      "have (reads that do not pass the criteria are discarded). 0 disables the filter. \n"
      "  --outfile1 <filename>    :file name for ouputing the reads from file1\n"
      "  --outfile2 <filename>    :file name for ouputing the reads from file2\n"
      "  --outfile3 <filename>    :file name for ouputing the reads from file3\n"
      "  --interleaved (read1|read2|index1|index2|index3),(read1|read2|index1|index2|index3)    :interleaved data\n"
      "  --umi_read (read1|read2|index1|index2|index3)       :in which input file can the UMI be found\n"
      "  --umi_offset integer     :offset \n"
      "  --umi_size               :number of bases after the offset\n"
      "  --cell_read (read1|read2|index1|index2|index3)      :in which input file can the cell be found\n"
      "  --cell_offset integer    :offset \n"
      "  --cell_size integer      :number of bases after the offset\n"
      "  --sample_read (read1|read2|index1|index2|index3)    :in which input file can the sample barcode be found\n"
      "  --sample_offset integer  :offset \n"
      "  --sample_size integer    :number of bases after the offset\n"
      "  --read1_offset integer   :\n"
      "  --read1_size integer     :\n"
      "  --read2_offset integer   :\n"
      "  --read2_size integer     :\n"
      "  --10x     : use 10X UMI tags (UB and UY) instead of the default tags defined in the SAM specification\n";
  fprintf(stderr, "usage: fastq_pre_barcodes --read1 fastq_file --outfile1 out_file [optional parameters]\n");
  fprintf(stderr, "%s\n", msg);
}

// one input: pieces, frames, and where the next iteration reads
struct Source {
  const char* path = nullptr;
  Input* in = nullptr;
  fqg_frame* frame = nullptr;
  fqg_file_state st{};
  bool probed = false;
  std::string format_line;
  uint64_t avail = 0;       // complete records in the current frame
  long use = 0;             // local index of the record the next iteration uses (may exceed avail)
  uint64_t records_before = 0;  // records in earlier frames
  bool final_piece = false;
  int tail_lines = 0;
  bool open_end = false;    // the file's last line has no '\n' and closes its last complete record: gzgets reads it up
                            // to the end of the file, and the file is at its end for gzeof from then on
  bool exhausted = false;   // no more data will come
  bool carry_pending = false;
  size_t carry_at = 0;      // bytes of the current piece covered by complete records
};

// state of the first record + the line fastq_get_readname prints on its first call for the file
bool probe_bytes(const char* data, size_t size, fqg_file_state* st, std::string* format_line) {
  st->is_pe = 1;
  if (fqg_probe_first_record(data, size, 1, st) != 0) return false;
  if (st->readname_format == FQG_NAME_CASAVA18) *format_line = "CASAVA=1.8\n";
  else if (st->readname_format == FQG_NAME_INTEGER) {
    // INTEGERNAME and NOP share a value; the text differs (src/fastq.c:465-474)
    const char* nl = static_cast<const char*>(memchr(data, '\n', size));
    std::string h(data + 1, nl ? (size_t)(nl - data) : size - 1);
    const std::string name = h.c_str();
    size_t i = 0;
    while (i < name.size() && name[i] >= '0' && name[i] <= '9') ++i;
    const std::string rest = name.substr(i);
    const bool all_digits = i > 0 && (rest.empty() || rest == "\n" || rest == "\r");
    *format_line = all_digits ? "Read name provided as an integer\n" : "Read name provided with no suffix\n";
  }
  return true;
}

// The reference reads a line beyond its gzgets buffers in pieces (src/fastq.c:249-253) and goes on out of step, every
// piece a line of its own.  Everything in front of this piece of input went the reference's way; the program runs itself
// again, as a child and on one device, on input that is cut where gzgets cuts it (fq_respawn.h, fq_reframe.h: inflated
// input is cut while it is read and never comes here) - every piece a line, a C string to the kernels as to the
// reference.  Only a stream that cannot be read twice is refused.
std::function<void()> g_before_respawn;  // (the one-device loop: text still on its way to stdout is written first)
[[noreturn]] void refuse_long_line(const char* path, uint64_t record) {
  if (fqhost::reframe_supported() && !fqhost::reframing() && strcmp(path, "-") != 0) {
    if (g_before_respawn) g_before_respawn();
    fqhost::respawn_reframed();
  }
  FQ_PRINT_ERROR("Error in file %s: record %lu has a line longer than the reference's line buffers (%d / %d bytes)", path,
                 (unsigned long)(record + 1), FQG_MAX_LABEL_LENGTH - 1, FQG_MAX_READ_LENGTH - 1);
  fflush(stdout);
  fqhost::leave(kExitSys);
}

void probe(Source& s) {
  if (s.probed || s.in->size() == 0) return;
  s.probed = probe_bytes(s.in->data(), s.in->size(), &s.st, &s.format_line);
}

// frame the next piece of `s`; false when the input is used up
bool refill(Source& s) {
  if (s.frame) {
    fqg_frame_release(s.frame);
    s.frame = nullptr;
    s.records_before += s.avail;
    s.use -= (long)s.avail;
    s.avail = 0;
  }
  if (s.exhausted) return false;
  if (s.carry_pending) {  // only now: the piece's bytes stay readable for messages until it is replaced
    s.in->carry_from(s.carry_at);
    s.carry_pending = false;
  }
  if (!s.in->next()) {
    s.exhausted = true;
    return false;
  }
  probe(s);
  fqg_validate_result r;
  LIB(fqg_validate(g_ctx, nullptr, s.in->data(), s.in->size(), FQG_MEM_HOST, s.in->final() ? 1 : 0, &s.st,
                   FQG_VALIDATE_FRAME_ONLY | s.in->vflags(), &r));
  if (r.code == FQG_E_LINE_TOO_LONG) refuse_long_line(s.in->path().c_str(), s.records_before + r.record);
  bool ends_here = r.stopped != 0;  // a header line that starts with a NUL byte: "no entry" (src/fastq.c:250), the input ends
  if (ends_here)  // the records in front of it, framed alone
    LIB(fqg_validate(g_ctx, nullptr, s.in->data(), r.consumed, FQG_MEM_HOST, 1, &s.st, FQG_VALIDATE_FRAME_ONLY | s.in->vflags(), &r));
  // a record with a sequence / second header / quality line that starts with NUL: an empty string to the reference - the
  // file is truncated THERE (src/fastq.c:254; tail_lines > 0), whatever follows
  const bool cut_short = !ends_here && r.code == FQG_E_TRUNCATED && !s.in->final();
  s.final_piece = s.in->final() || ends_here || cut_short;
  s.tail_lines = r.tail_lines;
  s.open_end = s.final_piece && !ends_here && r.tail_lines == 0 && r.n_records > 0 && s.in->size() > 0 && s.in->data()[s.in->size() - 1] != '\n';
  s.avail = r.n_records;
  if (r.n_records) LIB(fqg_frame_retain(g_ctx, &s.frame));
  if (!s.in->final() && !ends_here && !cut_short) {
    s.carry_pending = true;
    s.carry_at = r.consumed;
  } else s.exhausted = true;
  return r.n_records > 0 || !s.exhausted;
}

// ---- several GPUs (FQGPU_DEVICES=0,1,..): every input is cut into blocks of the same B records (fq_blocks.h), block j
// of all inputs is one unit, whichever device is free takes the next unit, and this thread takes the results in unit
// order - what the serial loop prints and writes, in its order.  Not for --interleaved input (a discarded read leaves
// the reference's file pointers out of step from there on, src/fastq_pre_barcodes.c:653 vs :722: a serial dependence).
struct BlockRun {
  const char* const* file;
  const fqg_barcode_params* P;
  GzipMembers* outgz;
  int out_sam, num_input_files;
};

[[noreturn]] void run_blocks(const BlockRun& A, const std::vector<int>& devs) {
  const size_t nd = devs.size();
  std::vector<fqg_ctx*> ctx(nd, nullptr);
  ctx[0] = g_ctx;  // (opened on devs[0])
  for (size_t i = 1; i < nd; ++i) {
    const int rc = fqg_open(devs[i], &ctx[i]);
    if (rc != 0) {
      FQ_PRINT_ERROR("FQGPU_DEVICES: device %d is not a usable MI355X GPU (fqg_open: %d)", devs[i], rc);
      fqhost::leave(kExitSys);
    }
  }
  RecordBlocks* cut[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  fqg_file_state st0[6];
  memset(st0, 0, sizeof(st0));
  std::string format_line[6];
  double record_bytes = 64;
  for (int x = READ1; x <= INDEX3; ++x)
    if (A.file[x]) {
      cut[x] = new RecordBlocks(g_ctx, A.file[x], (int)nd + 2);
      if (cut[x]->peek_size()) probe_bytes(cut[x]->peek(), cut[x]->peek_size(), &st0[x], &format_line[x]);
      else st0[x].is_pe = 1;
      const double rb = cut[x]->peek_lines() >= 4 ? 4.0 * (double)cut[x]->peek_size() / (double)cut[x]->peek_lines()
                                                  : (double)std::max<size_t>(cut[x]->peek_size(), 64);
      record_bytes = std::max(record_bytes, rb);
    }
  const size_t piece = getenv("FQGPU_CHUNK_MB") ? piece_bytes() : (size_t)128 << 20;
  uint64_t B = std::max<uint64_t>(1, (uint64_t)((double)piece / record_bytes));
  if (const char* e = getenv("FQGPU_BLOCK_RECORDS")) B = std::max<uint64_t>(1, strtoull(e, nullptr, 10));  // (tests: tiny blocks)
  for (int x = READ1; x <= INDEX3; ++x)
    if (cut[x]) cut[x]->start(B);

  struct Unit {
    uint64_t seq = 0;
    int rc = 0;
    std::string err;
    fqg_barcode_result r{};
    uint64_t n = 0;                               // iterations the unit had to offer
    uint64_t records[6] = {0, 0, 0, 0, 0, 0};     // complete records of every input's block
    int tail_lines[6] = {0, 0, 0, 0, 0, 0};
    bool final[6] = {false, false, false, false, false, false};
    bool open_end[6] = {false, false, false, false, false, false};  // see Source::open_end
    bool ends = false;                            // an input ends inside this unit although its block is not the last
    int long_line_file = 0;                       // an input of this unit has a line beyond the gzgets limits ...
    uint64_t long_line_record = 0;                // ... in this record of the file
    char* out[3] = {nullptr, nullptr, nullptr};   // the text of every output: pinned buffers that go round (OutPool)
    size_t out_cap[3] = {0, 0, 0};
    std::string wrong_header;                     // the text of the header line of a FQG_E_WRONG_HEADER finding
  };
  // The text a unit brings back from the GPU: pinned buffers that go round between the contexts' threads and the thread
  // that writes.  Into fresh pageable memory the copy ran at 7.5 GB/s per context (page faults, a bounce buffer) and the
  // copies of the next blocks TO the GPU waited behind it: 28 s for the 200 M pairs of the bench where the serial loop,
  // whose text has always travelled in pinned buffers, takes 6.7.  As many buffers as the units that may be under way
  // have outputs (`window` below); take() makes them as they are first asked for.
  struct OutPool {
    fqg_ctx* ctx;
    size_t limit, n_made = 0;
    std::vector<std::pair<char*, size_t>> free_;
    std::set<char*> pageable;  // buffers that are not pinned (the pinned allocation failed)
    std::mutex mu;
    std::condition_variable cv;
    bool quit = false;
    char* take(size_t bytes, size_t* cap) {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        size_t best = free_.size();  // a free one that is large enough: the smallest such
        for (size_t i = 0; i < free_.size(); ++i)
          if (free_[i].second >= bytes && (best == free_.size() || free_[i].second < free_[best].second)) best = i;
        if (best != free_.size()) {
          char* p = free_[best].first;
          *cap = free_[best].second;
          free_.erase(free_.begin() + (long)best);
          return p;
        }
        if (n_made < limit) {
          ++n_made;
          lk.unlock();
          const size_t want = bytes + bytes / 8 + 4096;
          char* p = static_cast<char*>(fqg_host_alloc(ctx, want));
          bool plain = false;
          if (!p) {  // (no pinned memory to be had: pageable memory does it, slower)
            p = static_cast<char*>(malloc(want));
            plain = p != nullptr;
          }
          *cap = p ? want : 0;
          if (!p || plain) {
            lk.lock();
            if (!p) --n_made;
            else pageable.insert(p);
          }
          return p;
        }
        if (!free_.empty()) {  // every buffer made, none of the free ones large enough: one of them makes room
          char* small = free_.back().first;
          free_.pop_back();
          --n_made;
          const bool plain = pageable.erase(small) != 0;
          lk.unlock();
          if (plain) free(small);
          else fqg_host_free(ctx, small);
          lk.lock();
          continue;
        }
        if (quit) return nullptr;
        cv.wait(lk);
      }
    }
    void give(char* p, size_t cap) {
      if (!p) return;
      std::lock_guard<std::mutex> lk(mu);
      free_.emplace_back(p, cap);
      cv.notify_all();
    }
    void stop() {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv.notify_all();
    }
  };
  // units under way or waiting for the writer: one per context and one more; none is begun beyond that (a context that
  // ran ahead would hold every buffer with units the writer cannot take yet, and the unit it waits for would find none)
  const uint64_t window = nd + 1;
  const size_t outs_per_unit = A.out_sam ? 1 : (size_t)((A.P->emit[1] ? 1 : 0) + (A.P->emit[2] ? 1 : 0));
  OutPool out_pool{g_ctx, (size_t)window * std::max<size_t>(outs_per_unit, 1)};
  uint64_t n_written = 0;  // units the writer is done with (under mu)
  std::map<uint64_t, Unit> done;
  std::mutex mu, fetch_mu;
  std::condition_variable cv;
  std::atomic<bool> stop{false};
  bool exhausted = false;     // (under fetch_mu)
  uint64_t next_seq = 0;      // (under fetch_mu)
  uint64_t n_units = ~0ull;   // known once a unit with the end of an input was handed out (under mu)

  const bool timing = getenv("FQGPU_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto work = [&](size_t di) {
    fqg_ctx* c = ctx[di];
    double t_wait = 0, t_frame = 0, t_transform = 0, t_out = 0, t_hand = 0;
    uint64_t n_units_done = 0;
    struct Report {
      bool on;
      size_t di;
      const double &w, &f, &t, &o, &h;
      const uint64_t& n;
      ~Report() {
        if (on) fprintf(fqhost::diag(), "fqgpu timing: context %zu: %llu units; waiting for blocks %.3f s, copy + framing %.3f s, transform %.3f s, output D2H %.3f s, handing over %.3f s\n",
                        di, (unsigned long long)n, w, f, t, o, h);
      }
    } report{timing, di, t_wait, t_frame, t_transform, t_out, t_hand, n_units_done};
    for (;;) {
      Unit u;
      Block b[6];
      const double t0 = timing ? now() : 0;
      {
        std::lock_guard<std::mutex> lk(fetch_mu);
        {
          std::unique_lock<std::mutex> lk2(mu);
          cv.wait(lk2, [&] { return stop.load() || next_seq < n_written + window; });
        }
        if (exhausted || stop) return;
        bool all = true;
        for (int x = READ1; x <= INDEX3 && all; ++x)
          if (cut[x] && !cut[x]->next(&b[x])) all = false;
        if (!all) {  // (only after an abort: a unit that holds the end of an input is the last one handed out)
          exhausted = true;
          return;
        }
        u.seq = next_seq++;
        for (int x = READ1; x <= INDEX3; ++x)
          if (cut[x] && b[x].final) exhausted = true;
        if (exhausted) {
          std::lock_guard<std::mutex> lk2(mu);
          n_units = u.seq + 1;
        }
      }
      const double t1 = timing ? now() : 0;
      t_wait += t1 - t0;
      const fqg_frame* frames[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      fqg_frame* held[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      fqg_file_state states[6];
      memcpy(states, st0, sizeof(states));
      uint64_t first[6] = {0, 0, 0, 0, 0, 0};
      auto lib_fail = [&](const char* what, int rc) {
        u.rc = rc;
        u.err = std::string(what) + ": " + fqg_last_error(c);
      };
      u.n = ~0ull;
      for (int x = READ1; x <= INDEX3 && !u.rc; ++x)
        if (cut[x]) {
          fqg_validate_result r;
          const int rc = fqg_validate(c, nullptr, b[x].data, b[x].size, FQG_MEM_HOST, b[x].final ? 1 : 0, &states[x],
                                      FQG_VALIDATE_FRAME_ONLY, &r);
          if (rc) {
            lib_fail("fqg_validate", rc);
            break;
          }
          if (r.code == FQG_E_LINE_TOO_LONG) {  // (the thread that takes the results starts the program over: refuse_long_line)
            u.long_line_file = x;
            u.long_line_record = u.seq * B + r.record;
            u.rc = FQG_ERR_ARG;
            break;
          }
          // a header line that starts with a NUL byte is "no entry" for the reference (src/fastq.c:250): this input ends
          // HERE, cleanly, whatever follows - the unit is the last one the consumer looks at
          const bool ends_here = r.stopped != 0;
          // ... and another line of a record that starts with NUL is an empty string: the file is truncated there
          // (src/fastq.c:254; tail_lines > 0) - the last unit as well
          const bool cut_short = !ends_here && r.code == FQG_E_TRUNCATED && !b[x].final;
          if (cut_short) u.ends = true;
          if (ends_here) {
            // frame the records in front of it once more, alone: what follows the NUL is not this file's any more
            u.ends = true;
            const int rc2 = fqg_validate(c, nullptr, b[x].data, r.consumed, FQG_MEM_HOST, 1, &states[x], FQG_VALIDATE_FRAME_ONLY, &r);
            if (rc2) {
              lib_fail("fqg_validate", rc2);
              break;
            }
          }
          if (!b[x].final && !ends_here && !cut_short && (r.n_records != B || r.consumed != b[x].size)) {
            u.rc = FQG_ERR_STATE;
            u.err = std::string("a block of ") + A.file[x] + " cut at a record boundary was not consumed whole";
            break;
          }
          u.records[x] = r.n_records;
          u.tail_lines[x] = ends_here ? 0 : r.tail_lines;
          u.final[x] = b[x].final || ends_here || cut_short;
          u.open_end[x] = b[x].final && !ends_here && r.tail_lines == 0 && r.n_records > 0 && b[x].size > 0 && b[x].data[b[x].size - 1] != '\n';
          u.n = std::min<uint64_t>(u.n, r.n_records);
          if (r.n_records) {
            const int rc2 = fqg_frame_retain(c, &held[x]);
            if (rc2) {
              lib_fail("fqg_frame_retain", rc2);
              break;
            }
            frames[x] = held[x];
          }
        }
      const double t2 = timing ? now() : 0;
      t_frame += t2 - t1;
      double t3 = t2;
      if (!u.rc && u.n > 0) {
        fqg_barcode_params Pb = *A.P;
        const int rc = fqg_barcodes_transform(c, frames, states, first, &Pb, u.n, u.seq * B, &u.r);
        if (rc) lib_fail("fqg_barcodes_transform", rc);
        t3 = timing ? now() : 0;
        t_transform += t3 - t2;
        for (int which = 0; which < 3 && !u.rc; ++which)
          if (u.r.out_bytes[which]) {
            u.out[which] = out_pool.take(u.r.out_bytes[which], &u.out_cap[which]);
            if (!u.out[which]) {
              u.rc = FQG_ERR_NOMEM;
              u.err = "no pinned memory for the output text";
              break;
            }
            const int rc2 = fqg_barcodes_output(c, which, u.out[which], u.r.out_bytes[which]);
            if (rc2) lib_fail("fqg_barcodes_output", rc2);
          }
        if (!u.rc && u.r.code == FQG_E_WRONG_HEADER) {
          const Block& bb = b[u.r.file];
          const char *p = bb.data, *end = bb.data + bb.size;
          for (uint64_t line = 0; p < end && line < 4 * u.r.n_done; ++line) {
            const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            p = nl ? nl + 1 : end;
          }
          const char* nl = p < end ? static_cast<const char*>(memchr(p, '\n', (size_t)(end - p))) : nullptr;
          u.wrong_header.assign(p, nl ? nl + 1 : end);
        }
      } else if (!u.rc) u.n = 0;
      const double t4 = timing ? now() : 0;
      t_out += t4 - t3;
      for (int x = READ1; x <= INDEX3; ++x)
        if (cut[x]) {
          if (held[x]) fqg_frame_release(held[x]);
          cut[x]->release(b[x]);
        }
      {
        std::lock_guard<std::mutex> lk(mu);
        const uint64_t seq = u.seq;
        done.emplace(seq, std::move(u));
        cv.notify_all();
      }
      if (timing) t_hand += now() - t4, ++n_units_done;
    }
  };
  std::vector<std::thread> th;
  for (size_t i = 0; i < nd; ++i) th.emplace_back(work, i);
  auto join_all = [&] {
    stop = true;
    out_pool.stop();
    {
      std::lock_guard<std::mutex> lk(mu);
      cv.notify_all();
    }
    for (int x = READ1; x <= INDEX3; ++x)
      if (cut[x]) cut[x]->abort();
    for (auto& t : th)
      if (t.joinable()) t.join();
  };

  unsigned long processed = 0, discarded = 0;
  bool first_batch = true;
  Unit last;
  bool have_last = false;
  double t_main_wait = 0;
  const double t_loop = now();
  for (uint64_t k = 0;; ++k) {
    Unit u;
    {
      const double tw = timing ? now() : 0;
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return done.count(k) || k >= n_units; });
      if (timing) t_main_wait += now() - tw;
      if (!done.count(k)) break;
      u = std::move(done[k]);
      done.erase(k);
    }
    if (u.rc) {
      join_all();
      if (u.long_line_file) refuse_long_line(A.file[u.long_line_file], u.long_line_record);
      FQ_PRINT_ERROR("GPU library failure in %s (%d)", u.err.c_str(), u.rc);
      fqhost::leave(kExitSys);
    }
    if (u.n > 0) {
      const fqg_barcode_result& r = u.r;
      if (first_batch && A.num_input_files > 1) {
        // format lines of the first fastq_get_readname call per file, in file order (src/fastq.c:459-485)
        for (int x = READ1; x <= INDEX3; ++x)
          if (A.file[x]) {
            if (r.code == FQG_E_WRONG_HEADER && r.iteration == 0 && r.file == x) break;
            fputs(format_line[x].c_str(), stderr);
            if (st0[x].space == FQG_SPACE_COLOUR) fputs("Color space\n", stderr);
          }
      }
      first_batch = false;
      for (uint64_t w = 0; w < r.n_short; ++w) fputs("Warning: Read too short - barcode not found\n", stderr);
      if (r.out_bytes[0]) fwrite(u.out[0], 1, r.out_bytes[0], stdout);
      for (int which = 1; which < 3; ++which)
        if (r.out_bytes[which] && !A.outgz[which].write(u.out[which], r.out_bytes[which])) {
          join_all();
          FQ_PRINT_ERROR("%s.\n", A.outgz[which].error().c_str());  // GZ_WRITE's gzerror() text, src/fastq.c:211-235
          fqhost::leave(kExitSys);
        }
      const unsigned long before = processed;
      processed += r.n_done;
      discarded += r.n_discarded;
      for (unsigned long c = (before / 100000 + 1) * 100000; c <= processed; c += 100000) {
        fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", c);
        fflush(stderr);
      }
      if (r.code != FQG_OK) {
        join_all();
        if (r.code == FQG_E_WRONG_HEADER) {  // src/fastq.c:448-451, with the file's own line counter
          const uint64_t reads_of_file = u.seq * B + r.n_done + 1;
          FQ_PRINT_ERROR("Error in file %s: line %lu: wrong header %s", A.file[r.file], (unsigned long)(4 * reads_of_file),
                         u.wrong_header.c_str());
          fqhost::leave(kExitFormat);
        }
        FQ_PRINT_ERROR("Readnames do not match across files (read #%ld)", (long)(processed + 1));
        fqhost::leave(kExitFormat);
      }
    }
    for (int which = 0; which < 3; ++which) {
      out_pool.give(u.out[which], u.out_cap[which]);
      u.out[which] = nullptr;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      ++n_written;
      cv.notify_all();
    }
    const bool ends = u.ends;
    last = std::move(u);
    have_last = true;
    if (ends) break;  // (later units, if any were handed out, are dropped)
  }
  join_all();
  if (timing) fprintf(fqhost::diag(), "fqgpu timing: the thread that writes: %.3f s in the loop, %.3f s of them waiting for the next unit\n", now() - t_loop, t_main_wait);
  // an incomplete record where the next read would have happened is a truncated file (src/fastq.c:254-257); a clean
  // end of any input just ends the loop.  The first input, in file order, that has nothing left decides - unless the
  // loop's own condition ends it first (fastq_files_eof, src/fastq_pre_barcodes.c:288-297, :594): an input whose last
  // line has no '\n' is at its end for gzeof once that line has been read, and no input is read again.
  bool loop_condition_ends_it = false;
  if (have_last)
    for (int x = READ1; x <= INDEX3; ++x)
      if (A.file[x] && last.final[x] && last.open_end[x] && last.records[x] == last.n) loop_condition_ends_it = true;
  if (have_last && !loop_condition_ends_it)
    for (int x = READ1; x <= INDEX3; ++x)
      if (A.file[x]) {
        if (!(last.final[x] && last.records[x] == last.n)) continue;
        if (last.tail_lines[x] > 0) {
          FQ_PRINT_ERROR("Error in file %s: line %lu: file truncated", A.file[x],
                         (unsigned long)(4 * (last.seq * B + last.records[x])));
          fqhost::leave(1);
        }
        break;
      }
  FQ_PRINT_INFO("Reads processed: %ld", (long)processed);
  FQ_PRINT_INFO("Reads discarded: %ld", (long)discarded);
  if (!A.out_sam)
    for (int x = READ1; x <= READ2; ++x)
      if (A.P->emit[x] && !A.outgz[x].close()) {
        FQ_PRINT_ERROR("unable to close file descriptor");
        fqhost::leave(kExitSys);
      }
  fflush(stdout);
  pb_json_metrics((long)processed, (long)discarded, std::set<int>(devs.begin(), devs.end()).size());
  fqhost::leave(0);
}

}  // namespace

int main(int argc, char** argv) {
  fqhost::install_counted_output(argv);  // (fq_respawn.h: a run that starts over on input cut at the gzgets limits prints nothing twice)
  static int verbose = 0, paired = 0, help = 0, out_sam = 0, tenx = 0;
  fqg_barcode_params P;
  memset(&P, 0, sizeof(P));
  const char* file[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const char* outfile[3] = {nullptr, nullptr, nullptr};
  P.phred_encoding = 64;
  P.umi_read = P.cell_read = P.sample_read = -1;
  P.umi_offset = P.cell_offset = P.sample_offset = -1;
  P.read_offset[1] = P.read_offset[2] = -1;
  int num_input_files = 0;
  bool has_interleaved = false;
  opterr = 0;
  fprintf(stderr, "fastq_utils %s\n", "0.25.3");

  static struct option long_options[] = {{"verbose", no_argument, &verbose, 1},
                                         {"brief", no_argument, &verbose, 0},
                                         {"paired_end", no_argument, &paired, 1},
                                         {"single_end", no_argument, &paired, 0},
                                         {"sam", no_argument, &out_sam, 1},
                                         {"fastq", no_argument, &out_sam, 0},
                                         {"help", no_argument, &help, 1},
                                         {"umi_read", required_argument, 0, 'a'},
                                         {"umi_offset", required_argument, 0, 'b'},
                                         {"umi_size", required_argument, 0, 'c'},
                                         {"read1_offset", required_argument, 0, 'd'},
                                         {"read1_size", required_argument, 0, 'e'},
                                         {"read2_offset", required_argument, 0, 'f'},
                                         {"read2_size", required_argument, 0, 'g'},
                                         {"min_qual", required_argument, 0, 'h'},
                                         {"cell_read", required_argument, 0, 'i'},
                                         {"cell_offset", required_argument, 0, 'j'},
                                         {"cell_size", required_argument, 0, 'k'},
                                         {"read1", required_argument, 0, 'l'},
                                         {"read2", required_argument, 0, 'm'},
                                         {"index1", required_argument, 0, 't'},
                                         {"index2", required_argument, 0, 'v'},
                                         {"index3", required_argument, 0, 'u'},
                                         {"outfile1", required_argument, 0, 'n'},
                                         {"outfile2", required_argument, 0, 'o'},
                                         {"interleaved", required_argument, 0, 'z'},
                                         {"sample_read", required_argument, 0, 'p'},
                                         {"sample_offset", required_argument, 0, 'q'},
                                         {"sample_size", required_argument, 0, 'r'},
                                         {"phred_encoding", required_argument, 0, 's'},
                                         {"10x", no_argument, &tenx, 1},
                                         {0, 0, 0, 0}};
  auto set_input = [&](const char* name, int idx) {
    if (name && !file[idx]) num_input_files++;
    file[idx] = name;
  };
  for (;;) {
    int option_index = 0;
    const int c = getopt_long(argc, argv, "a:b:c:d:e:f:g:h:i:j:k:l:m:n:o:p:q:r:s:t:u:z:X", long_options, &option_index);
    if (c == -1) break;
    switch (c) {
      case 'X': tenx = 1; break;
      case 'z': {
        char tmps[1025];
        strncpy(tmps, optarg, 1024);
        tmps[1024] = 0;
        int xx = 0;
        char* token = strtok(tmps, ",");
        int refs[3] = {0, 0, 0};
        while (token != nullptr) {
          refs[xx] = read_index2read_idx(token);
          token = xx == 2 ? nullptr : strtok(nullptr, ",");
          ++xx;
        }
        if (xx != 2) {
          FQ_PRINT_ERROR("two file references should be passed to --interleaved");
          fqhost::leave(1);
        }
        P.interleaved[0] = refs[0];
        P.interleaved[1] = refs[1];
        has_interleaved = true;
        break;
      }
      case 'a': P.umi_read = read_index2read_idx(optarg); break;
      case 'b': P.umi_offset = atol(optarg); break;
      case 'c': P.umi_size = atol(optarg); break;
      case 'd': P.read_offset[READ1] = atol(optarg); break;
      case 'e': P.read_size[READ1] = atol(optarg); break;
      case 'f': P.read_offset[READ2] = atol(optarg); break;
      case 'g': P.read_size[READ2] = atol(optarg); break;
      case 'h': P.min_qual = atoi(optarg); break;
      case 'i': P.cell_read = read_index2read_idx(optarg); break;
      case 'j': P.cell_offset = atol(optarg); break;
      case 'k': P.cell_size = atol(optarg); break;
      case 'l': set_input(optarg, READ1); break;
      case 'm': set_input(optarg, READ2); break;
      case 't': set_input(optarg, INDEX1); break;
      case 'v': set_input(optarg, INDEX2); break;
      case 'u': set_input(optarg, INDEX3); break;
      case 'n': outfile[READ1] = optarg; break;
      case 'o': outfile[READ2] = optarg; break;
      case 'p': P.sample_read = read_index2read_idx(optarg); break;
      case 'q': P.sample_offset = atol(optarg); break;
      case 'r': P.sample_size = atol(optarg); break;
      case 's': P.phred_encoding = atoi(optarg); break;
      default: break;
    }
  }
  if (help) {
    print_usage();
    fqhost::leave(0);
  }
  FQ_PRINT_INFO("Validating options...");
  if (!file[READ1]) {  // validate_options, src/fastq_pre_barcodes.c:91-107
    FQ_PRINT_ERROR("missing input file (-read1)");
    fqhost::leave(1);
  }
  if (paired && !file[READ2]) {
    FQ_PRINT_ERROR("if paired_end is used then two fastq files should be provided - missing input file (-read2)");
    fqhost::leave(kExitParams);
  }
  if (!outfile[READ1]) {
    FQ_PRINT_ERROR("if single_end then -outfile1 should be provided");
    fqhost::leave(kExitParams);
  }
  FQ_PRINT_INFO("Options OK.");
  FQ_PRINT_INFO("input files %d", num_input_files);

  const char* dev = getenv("FQGPU_DEVICE");
  std::vector<int> devices = devices_from_env();  // FQGPU_DEVICES=0,1,..: record blocks over several GPUs
  if (has_interleaved) devices.clear();
  // One GPU, nothing said: the loop over record blocks all the same - every input has a reader of its own there (the serial
  // loop below reads them one after the other: 1.4 s against 0.9 - 1.0 s for 40 M reads and their index reads from tmpfs,
  // profiles/r07_multi_dev_legs.txt).  FQGPU_SERIAL_LOOP=1 keeps the serial loop, which is also what --interleaved input
  // runs through and a run that was started over on input cut at the gzgets limits (fq_respawn.h).
  const bool block_loop = !has_interleaved && !fqhost::reframing() && !getenv("FQGPU_SERIAL_LOOP");
  // (one context: a second one on the same GPU brought nothing that could be told from the noise - 200 M pairs to SAM 5.0 -
  // 6.1 s with one, 5.4 - 5.5 with two; 40 M reads 1.01 against 1.10 - and costs 30 ms to open; FQGPU_DEVICES=0,0 asks for it)
  if (devices.empty() && block_loop) devices.assign(1, dev ? atoi(dev) : 0);
  const bool multi = block_loop ? !devices.empty() : devices.size() > 1;
  int rc = fqg_open(multi ? devices[0] : (dev ? atoi(dev) : 0), &g_ctx);
  if (rc != 0) {
    FQ_PRINT_ERROR("no usable MI355X GPU (fqg_open: %d); this build has no CPU path", rc);
    fqhost::leave(kExitSys);
  }
  P.out_sam = out_sam;
  P.tenx = tenx;
  Source src[6];
  for (int x = READ1; x <= INDEX3; ++x)
    if (file[x]) {
      P.present[x] = 1;
      src[x].path = file[x];
      if (!multi) src[x].in = new Input(g_ctx, file[x], piece_bytes());
    }
  if (has_interleaved && (!file[P.interleaved[0]] || !file[P.interleaved[1]])) {
    FQ_PRINT_ERROR("--interleaved refers to an input that was not given");
    fqhost::leave(kExitParams);
  }
  GzipMembers outgz[3];  // gzip level 4 like the reference's "w4", one member per 4 MiB block, all cores
  bool out_open[3] = {false, false, false};
  if (!out_sam) {
    for (int x = READ1; x <= READ2; ++x)
      if (outfile[x]) {
        if (!file[x]) {
          FQ_PRINT_ERROR("--outfile%d needs --read%d", x, x);
          fqhost::leave(kExitParams);
        }
        P.emit[x] = 1;
        // ("-": the reference's gzdopen(stdout, "wb") compresses at the default level; "w4" otherwise)
        if (!outgz[x].open(outfile[x], (outfile[x][0] == '-' && outfile[x][1] == 0) ? Z_DEFAULT_COMPRESSION : 4)) {
          FQ_PRINT_ERROR("Unable to open %s", outfile[x]);
          fqhost::leave(kExitParams);
        }
        out_open[x] = true;
      }
  } else {
    printf("@HD\tVN:1.0 SO:unknown\n");
    printf("@PG\tID:1 PN:fastq_pre_barcodes CL:%s", argv[0]);
    for (int c = 1; c < argc - 1; c++) printf(" %s", argv[c]);  // (the reference drops the last word)
    printf("\n");
  }

  if (multi) {
    BlockRun A{file, &P, outgz, out_sam, num_input_files};
    run_blocks(A, devices);
  }
  unsigned long processed = 0, discarded = 0;
  bool first_batch = true;
  // What a batch prints goes to a writer thread (in order: one thread, one queue): gzip'ing and writing batch k - the
  // reference's whole cost in FASTQ mode - runs beside reading, framing and transforming batch k + 1.  Two batches may
  // wait; drain() before anything else may be said or the program leaves.
  // (the text travels in pinned buffers that go round: the copy from the GPU into fresh pageable memory - page faults, a
  // bounce buffer - took four times as long as everything else the program does)
  struct OutJob {
    int which = 0;
    char* text = nullptr;
    size_t size = 0, cap = 0;
  };
  struct AsyncOut {
    GzipMembers* gz;
    std::deque<OutJob> q;
    std::vector<OutJob> spare;  // buffers that have been written out
    std::mutex mu;
    std::condition_variable cv;
    bool quit = false, failed = false, busy = false;
    std::thread th;
    double t_write = 0;
    void start() {
      th = std::thread([this] {
        for (;;) {
          OutJob j;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return quit || !q.empty(); });
            if (q.empty()) return;
            j = std::move(q.front());
            q.pop_front();
            busy = true;
          }
          const auto t0 = std::chrono::steady_clock::now();
          bool ok = true;
          if (j.which == 0) ok = fwrite(j.text, 1, j.size, stdout) == j.size;
          else ok = gz[j.which].write(j.text, j.size);
          {
            std::lock_guard<std::mutex> lk(mu);
            t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (!ok) failed = true;
            busy = false;
            spare.push_back(j);
          }
          cv.notify_all();
        }
      });
    }
    // a pinned buffer of at least `bytes` for the next batch's text (a spare one when it is large enough)
    OutJob buffer(fqg_ctx* ctx, size_t bytes) {
      {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < spare.size(); ++i)
          if (spare[i].cap >= bytes) {
            OutJob j = spare[i];
            spare.erase(spare.begin() + (long)i);
            return j;
          }
        if (!spare.empty()) {  // too small: replace it
          fqhost::slot_release(ctx, spare.back().text);
          spare.pop_back();
        }
      }
      OutJob j;
      j.cap = bytes + bytes / 8 + (1u << 20);
      j.text = fqhost::slot_alloc(ctx, j.cap);
      if (!j.text) {
        FQ_PRINT_ERROR("unable to allocate %zu bytes of pinned memory", j.cap);
        fqhost::leave(kExitSys);
      }
      return j;
    }
    void push(const OutJob& j) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return q.size() < 2; });
      q.push_back(j);
      lk.unlock();
      cv.notify_all();
    }
    bool drain() {  // everything handed over is written; false: a write failed
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return q.empty() && !busy; });
      return !failed;
    }
    void stop() {
      {
        std::lock_guard<std::mutex> lk(mu);
        quit = true;
      }
      cv.notify_all();
      if (th.joinable()) th.join();
    }
  } outq;
  outq.gz = outgz;
  outq.start();
  auto drain_or_die = [&] {
    if (!outq.drain()) {
      const char* why = "write error";
      for (int which = 1; which < 3; ++which)
        if (!outgz[which].error().empty()) why = outgz[which].error().c_str();
      FQ_PRINT_ERROR("%s.\n", why);  // GZ_WRITE's gzerror() text, src/fastq.c:211-235
      fqhost::leave(kExitSys);
    }
  };
  const bool timing = getenv("FQGPU_TIMING") != nullptr;
  double t_refill = 0, t_transform = 0, t_fetch = 0, t_hand = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto step_of = [&](int x) { return (has_interleaved && (x == P.interleaved[0] || x == P.interleaved[1])) ? 2L : 1L; };
  for (int x = READ1; x <= INDEX3; ++x)
    if (file[x]) src[x].use = (has_interleaved && x == P.interleaved[1]) ? 1 : 0;

  std::vector<OutJob> in_flight;  // output of the last transform, on its way to the host
  auto land_output = [&] {
    if (in_flight.empty()) return;
    const double t_d = now();
    LIB(fqg_barcodes_output_wait(g_ctx));
    const double t_e = now();
    for (const OutJob& job : in_flight) outq.push(job);
    in_flight.clear();
    t_fetch += t_e - t_d;
    t_hand += now() - t_e;
  };
  g_before_respawn = [&] {
    land_output();
    drain_or_die();
  };
  for (;;) {
    // every input needs a frame that holds the record its next iteration uses
    const double t_a = now();
    bool out_of_data = false;
    for (int x = READ1; x <= INDEX3 && !out_of_data; ++x)
      if (file[x]) {
        Source& s = src[x];
        while (!s.frame || s.use >= (long)s.avail) {
          if (!refill(s) && !s.frame) {
            out_of_data = true;
            break;
          }
          if (s.frame && s.use < (long)s.avail) break;
          if (s.exhausted && (!s.frame || s.use >= (long)s.avail)) {
            out_of_data = true;
            break;
          }
        }
      }
    if (out_of_data) break;
    uint64_t n = ~0ull;
    for (int x = READ1; x <= INDEX3; ++x)
      if (file[x]) {
        const Source& s = src[x];
        const uint64_t left = s.avail - (uint64_t)s.use;
        const long st = step_of(x);
        n = std::min<uint64_t>(n, (left + st - 1) / st);
      }
    if (n == 0 || n == ~0ull) break;
    const fqg_frame* frames[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    fqg_file_state states[6];
    uint64_t first[6] = {0, 0, 0, 0, 0, 0};
    memset(states, 0, sizeof(states));
    fqg_barcode_params Pb = P;
    for (int x = READ1; x <= INDEX3; ++x)
      if (file[x]) {
        frames[x] = src[x].frame;
        states[x] = src[x].st;
        // the library adds +1 for the second interleaved reference itself
        first[x] = (uint64_t)src[x].use - ((has_interleaved && x == P.interleaved[1]) ? 1 : 0);
      }
    fqg_barcode_result r;
    land_output();  // (the transform writes the device buffers the previous batch's text is copied from)
    const double t_b = now();
    LIB(fqg_barcodes_transform(g_ctx, frames, states, first, &Pb, n, processed, &r));
    const double t_c = now();
    t_refill += t_b - t_a;
    t_transform += t_c - t_b;
    if (first_batch && num_input_files > 1) {
      drain_or_die();  // (nothing is in flight yet: the first batch)
      // format lines of the first fastq_get_readname call per file, in file order (src/fastq.c:459-485)
      for (int x = READ1; x <= INDEX3; ++x)
        if (file[x]) {
          if (r.code == FQG_E_WRONG_HEADER && r.iteration == 0 && r.file == x) break;
          fputs(src[x].format_line.c_str(), stderr);
          if (src[x].st.space == FQG_SPACE_COLOUR) fputs("Color space\n", stderr);
        }
    }
    first_batch = false;
    for (uint64_t w = 0; w < r.n_short; ++w) fputs("Warning: Read too short - barcode not found\n", stderr);
    // The text comes back on a stream of its own, beside the upload and framing of the NEXT pieces of input (the link
    // carries both directions at once: tools/kbench/duplex.hip); it is handed to the writer (stdout / gzip) when the
    // next batch is about to be transformed - land_output(), at the top of the loop - or the loop ends.
    for (int which = 0; which < 3; ++which)
      if (r.out_bytes[which]) {
        const double t_d = now();
        OutJob job = outq.buffer(g_ctx, r.out_bytes[which]);
        job.which = which;
        job.size = r.out_bytes[which];
        LIB(fqg_barcodes_output_begin(g_ctx, which, job.text, job.size));
        in_flight.push_back(job);
        t_fetch += now() - t_d;
      }
    const unsigned long before = processed;
    processed += r.n_done;
    discarded += r.n_discarded;
    for (unsigned long c = (before / 100000 + 1) * 100000; c <= processed; c += 100000) {
      fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", c * (has_interleaved ? 2 : 1));
      fflush(stderr);
    }
    if (r.code != FQG_OK) {
      land_output();
      drain_or_die();
      if (r.code == FQG_E_WRONG_HEADER) {
        // src/fastq.c:448-451, with the file's own line counter
        Source& s = src[r.file];
        const uint64_t rec = s.records_before + (uint64_t)s.use + r.n_done * step_of(r.file);
        const char* b = s.in->data();
        // the header text: first line of that record in the current piece
        const char* p = b;
        const char* end = b + s.in->size();
        const uint64_t local = (uint64_t)s.use + r.n_done * step_of(r.file);
        for (uint64_t line = 0; p < end && line < 4 * local; ++line) {
          const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
          p = nl ? nl + 1 : end;
        }
        const char* nl = p < end ? static_cast<const char*>(memchr(p, '\n', (size_t)(end - p))) : nullptr;
        const std::string hdr(p, nl ? nl + 1 : end);
        const uint64_t reads_of_file = rec + 1;  // records this file has handed out so far
        FQ_PRINT_ERROR("Error in file %s: line %lu: wrong header %s", s.path, (unsigned long)(4 * reads_of_file),
                       hdr.c_str());
        fqhost::leave(kExitFormat);
      }
      FQ_PRINT_ERROR("Readnames do not match across files (read #%ld)", (long)(processed + 1));
      fqhost::leave(kExitFormat);
    }
    const bool ended_on_discard = has_interleaved && r.n_done < n;
    for (int x = READ1; x <= INDEX3; ++x)
      if (file[x]) {
        src[x].use += (long)r.n_done * step_of(x);
        if (ended_on_discard && x == P.interleaved[0]) src[x].use -= 1;  // no re-synchronising read after a discard
      }
  }
  land_output();
  drain_or_die();
  outq.stop();
  if (timing)
    fprintf(fqhost::diag(), "\nfqgpu timing: reading + framing %.3f s, transform %.3f s, output D2H %.3f s, waiting for the writer %.3f s; "
                    "the writer (gzip / stdout) worked %.3f s beside them\n", t_refill, t_transform, t_fetch, t_hand, outq.t_write);
  // an incomplete record where the next read would have happened is a truncated file
  // (src/fastq.c:254-257); a clean end of any input just ends the loop - and so does the loop's own condition
  // (fastq_files_eof, src/fastq_pre_barcodes.c:288-297, :594), before any input is read again: an input whose last line
  // has no '\n' is at its end for gzeof once that line has been read
  bool loop_condition_ends_it = false;
  for (int x = READ1; x <= INDEX3; ++x)
    if (file[x]) {
      const Source& s = src[x];
      if (s.exhausted && s.open_end && (!s.frame || s.use == (long)s.avail)) loop_condition_ends_it = true;
    }
  for (int x = READ1; x <= INDEX3 && !loop_condition_ends_it; ++x)
    if (file[x]) {
      Source& s = src[x];
      const bool drained = s.exhausted && (!s.frame || s.use >= (long)s.avail);
      if (!drained) continue;
      if (s.tail_lines > 0 && (!s.frame || s.use == (long)s.avail ||
                               (has_interleaved && x == P.interleaved[1] && s.use == (long)s.avail + 1))) {
        FQ_PRINT_ERROR("Error in file %s: line %lu: file truncated", s.path,
                       (unsigned long)(4 * (s.records_before + s.avail)));
        fqhost::leave(1);
      }
      break;
    }
  FQ_PRINT_INFO("Reads processed: %ld", (long)processed);
  FQ_PRINT_INFO("Reads discarded: %ld", (long)discarded);
  if (!out_sam)
    for (int x = READ1; x <= READ2; ++x)
      if (out_open[x] && !outgz[x].close()) {
        FQ_PRINT_ERROR("unable to close file descriptor");
        fqhost::leave(kExitSys);
      }
  fflush(stdout);
  pb_json_metrics((long)processed, (long)discarded, 1);
  fqhost::leave(0);
}
