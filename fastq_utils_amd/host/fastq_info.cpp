// fastq_info - drop-in for the reference program of the same name (reference src/fastq_info.c),
// with the per-record loops replaced by bulk calls into libfqgpu.so (include/fqg.h).
//
// Same command line, same stdout/stderr text, same exit status.  What runs where:
//   host   option parsing, (gz) reading into pinned pieces, the once-per-file probes, the order in
//          which findings of different kinds win, message text, the final summary
//   GPU    framing, validation, statistics, read-name index / pairing / name comparison
// There is no CPU path for the record work: without a GPU the program fails at start-up.
#include <getopt.h>
#include <regex.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include <chrono>
#include <map>

#include "fq_common.h"
#include "fq_multi.h"
#include "fq_names_multi.h"

namespace {

void print_usage(int verbose) {
  printf("Usage: fastq_info [-r -e -s -q -h] fastq1 [fastq2 file|pe]\n");
  if (verbose) {
    printf(" -h  : print this help message\n");
    printf(" -s  : the reads in the two fastq files have the same ordering\n");
    printf(" -e  : do not fail with empty files\n");
    printf(" -q  : do not fail if quality encoding cannot be determined\n");
    printf(" -r  : skip check for duplicated readnames\n");
  }
}

static const std::chrono::steady_clock::time_point g_t_start = std::chrono::steady_clock::now();
static double since_start() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t_start).count(); }

// ---- -r, one file: validate_single_fastq_file (src/fastq_info.c:155-176) ------------------
void run_single_noindex(const char* path, Stats& S) {
  Input in(g_ctx, path, piece_bytes());
  Probe pr;
  uint64_t base = 0;
  bool info_pending = true;
  // FQGPU_TIMING=1: where the wall time of the loop goes (waiting for the reader / copying + validating), on stderr
  const bool timing = getenv("FQGPU_TIMING") != nullptr;
  double t_wait = 0, t_gpu = 0;
  uint64_t n_pieces = 0, n_bytes = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  struct Report {
    const bool on;
    const double& w;
    const double& g;
    const uint64_t& p;
    const uint64_t& b;
    ~Report() {
      if (on) fprintf(fqhost::diag(), "\nfqgpu timing: %llu pieces, %.3f GB; waiting for the reader %.3f s, copy + validate %.3f s; %.3f s since the program started\n",
                      (unsigned long long)p, b / 1e9, w, g, (double)clock() / CLOCKS_PER_SEC >= 0 ? since_start() : 0.0);
    }
  } report{timing, t_wait, t_gpu, n_pieces, n_bytes};
  for (;;) {
    const double t0 = timing ? now() : 0;
    if (!in.next()) break;
    const double t1 = timing ? now() : 0;
    probe_piece(pr, in.data(), in.size(), 1);
    fqg_validate_result r;
    LIB(fqg_validate(g_ctx, S.acc1, in.data(), in.size(), FQG_MEM_HOST, in.final() ? 1 : 0, &pr.st, in.vflags(), &r));
    if (timing) {
      if (!n_pieces) fprintf(fqhost::diag(), "fqgpu timing: first piece in hand %.3f s, validated %.3f s after the program started\n",
                             since_start() - (now() - t1), since_start());
      t_wait += t1 - t0;
      t_gpu += now() - t1;
      ++n_pieces;
      n_bytes += in.size();
    }
    if (info_pending && base == 0 && r.n_records > 0) {
      if (!(r.code && r.record == 0 && is_early_code(r.code))) print_probe(pr);
      info_pending = false;
    }
    if (r.code) {
      const uint64_t R = base + r.record;
      ticker(base + 1, R, 100000);
      if (r.code == FQG_E_TRUNCATED) fail_truncated(path, 4 * R);
      if (r.code == FQG_E_LINE_TOO_LONG) fail_too_long(path, R);
      print_validation_error(path, 4 * (R + 1), r, locate_record(in.data(), in.size(), r.record));
      fqhost::leave(kExitFormat);
    }
    ticker(base + 1, base + r.n_records, 100000);
    base += r.n_records;
    if (r.stopped) break;
    if (!in.final()) in.carry_from(r.consumed);
  }
  printf("\n");
  fqg_file_stats fs;
  LIB(fqg_acc_read(S.acc1, &fs));
  S.num_reads1 = fs.num_rds;
}

// Pieces and slots of the loops over several contexts.  Pinned memory is what such a run pays for at both ends - 23 ms per
// 128 MiB when a slot is made, and again when the process leaves - so the pieces are half the one-context loop's and the
// slots as many as can be in use at once: two with the cutter (the piece whose last record is still open, the one being
// read), one with every context, one on its way back.
size_t multi_piece_bytes() { return getenv("FQGPU_CHUNK_MB") ? piece_bytes() : (size_t)64 << 20; }
int multi_slots(size_t n_contexts) { return (int)n_contexts + 3; }

// ---- -r, one file, several GPUs (FQGPU_DEVICES=0,1,..): the same loop over record-aligned pieces that have no
// order among them (fq_multi.h).  One thread + context + accumulator per device; this thread takes the results in
// file order, so the ticker, the first finding and its text are the serial loop's; the statistics of a clean file
// are the element-wise merge of the devices' accumulators (fqg_acc_export / fqg_acc_merge).
void run_single_noindex_multi(const char* path, Stats& S, const std::vector<int>& devs) {
  struct Dev {
    fqg_ctx* ctx = nullptr;
    fqg_acc* acc = nullptr;
  };
  std::vector<Dev> D(devs.size());
  D[0].ctx = g_ctx;  // (opened on devs[0])
  D[0].acc = S.acc1;
  // (before the piece cutter starts to pin its slots: an allocation of the runtime waits for the one in front of it, and
  // the seven small ones of a context behind slots of tens of MiB took 0.19 s)
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: opening %zu more contexts %.3f s after the program started\n", devs.size() - 1, since_start());
  for (size_t i = 1; i < devs.size(); ++i) {
    const int rc = fqg_open(devs[i], &D[i].ctx);
    if (rc != 0) {
      FQ_PRINT_ERROR("FQGPU_DEVICES: device %d is not a usable MI355X GPU (fqg_open: %d)", devs[i], rc);
      fqhost::leave(kExitSys);
    }
    if (fqg_acc_create(D[i].ctx, &D[i].acc) != 0) die_lib("fqg_acc_create", -1);
  }
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: ... opened %.3f s after the program started\n", since_start());
  const size_t piece = multi_piece_bytes();
  struct Done {
    Piece p;
    fqg_validate_result r{};
    int rc = 0;
    std::string err;
  };
  std::map<uint64_t, Done> done;
  std::mutex mu, fetch_mu;
  std::condition_variable cv;
  std::atomic<bool> stop{false};
  bool exhausted = false;  // (under fetch_mu)
  uint64_t n_pieces = ~0ull;  // known once the final piece was handed out (under mu)
  Probe pr;
  bool rerun_serial = false, probe_printed = false;
  {
    AlignedPieces src(g_ctx, path, piece, multi_slots(devs.size()));
    const bool timing = getenv("FQGPU_TIMING") != nullptr;

    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto work = [&](size_t di) {
      double t_wait = 0, t_gpu = 0;
      uint64_t n = 0;
      struct Report {
        bool on;
        size_t di;
        const double &w, &g;
        const uint64_t& n;
        ~Report() {
          if (on) fprintf(fqhost::diag(), "fqgpu timing: context %zu: %llu pieces; waiting for a piece %.3f s, copy + validate %.3f s; %.3f s since the program started\n",
                          di, (unsigned long long)n, w, g, since_start());
        }
      } report{timing, di, t_wait, t_gpu, n};
      for (;;) {
        Done d;
        fqg_file_state st;
        const double t0 = timing ? now() : 0;
        {
          std::lock_guard<std::mutex> lk(fetch_mu);
          if (exhausted || stop || !src.next(&d.p)) {
            exhausted = true;
            return;
          }
          if (d.p.final) exhausted = true;
          probe_piece(pr, d.p.data, d.p.size, 1);  // piece 0 is handed out first: the state is the first record's
          st = pr.st;
        }
        const double t1 = timing ? now() : 0;
        d.rc = fqg_validate(D[di].ctx, D[di].acc, d.p.data, d.p.size, FQG_MEM_HOST, d.p.final ? 1 : 0, &st, 0, &d.r);
        if (timing) t_wait += t1 - t0, t_gpu += now() - t1, ++n;
        if (d.rc) d.err = fqg_last_error(D[di].ctx);
        else if (!d.p.final && d.r.code == FQG_OK && !d.r.stopped && d.r.consumed != d.p.size) {
          d.rc = FQG_ERR_ARG;
          d.err = "a piece cut at a record boundary was not consumed whole";
        }
        std::lock_guard<std::mutex> lk(mu);
        if (d.p.final) n_pieces = d.p.seq + 1;
        done.emplace(d.p.seq, std::move(d));
        cv.notify_all();
      }
    };
    std::vector<std::thread> th;
    for (size_t i = 0; i < devs.size(); ++i) th.emplace_back(work, i);
    auto join_all = [&] {
      stop = true;
      src.abort();  // (a worker may be waiting for a piece, the producer for a slot that stays held after a finding)
      for (auto& t : th)
        if (t.joinable()) t.join();
    };
    bool info_pending = true;
    for (uint64_t k = 0;; ++k) {
      Done d;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done.count(k) || k >= n_pieces; });
        if (!done.count(k)) break;
        d = std::move(done[k]);
        done.erase(k);
      }
      const uint64_t base = d.p.first_record;
      const fqg_validate_result& r = d.r;
      if (d.rc) {
        join_all();
        FQ_PRINT_ERROR("GPU library failure in fqg_validate (%d): %s", d.rc, d.err.c_str());
        fqhost::leave(kExitSys);
      }
      if (info_pending && base == 0 && r.n_records > 0) {
        if (!(r.code && r.record == 0 && is_early_code(r.code))) {
          print_probe(pr);
          probe_printed = true;
        }
        info_pending = false;
      }
      if (r.code) {
        join_all();
        const uint64_t R = base + r.record;
        ticker(base + 1, R, 100000);
        if (r.code == FQG_E_TRUNCATED) fail_truncated(path, 4 * R);
        if (r.code == FQG_E_LINE_TOO_LONG) fail_too_long(path, R);
        print_validation_error(path, 4 * (R + 1), r, locate_record(d.p.data, d.p.size, r.record));
        fqhost::leave(kExitFormat);
      }
      if (r.stopped) {
        // a NUL at a record start ends the file here (src/fastq.c:250): what later pieces added to the accumulators
        // does not belong to it.  Rare enough to simply run the serial loop again on one device.
        join_all();
        rerun_serial = true;
        break;
      }
      ticker(base + 1, base + r.n_records, 100000);
      src.release(d.p);
      if (d.p.final) break;
    }
    join_all();
  }
  if (rerun_serial) {
    for (size_t i = 1; i < D.size(); ++i) {
      fqg_acc_destroy(D[i].acc);
      fqg_close(D[i].ctx);
    }
    if (strcmp(path, "-") == 0) {
      FQ_PRINT_ERROR("Error in file %s: a NUL byte at a record start with FQGPU_DEVICES on a stream: use one device",
                     path);
      fqhost::leave(kExitSys);
    }
    LIB(fqg_acc_reset(S.acc1));
    probe_line_is_out() = probe_printed;
    run_single_noindex(path, S);
    probe_line_is_out() = false;
    return;
  }
  {
    std::vector<char> buf;
    for (size_t i = 1; i < D.size(); ++i) {
      size_t used = 0;
      if (fqg_acc_export(D[i].acc, nullptr, 0, &used) != 0) die_lib("fqg_acc_export", -1);
      buf.resize(used);
      if (fqg_acc_export(D[i].acc, buf.data(), buf.size(), &used) != 0) die_lib("fqg_acc_export", -1);
      LIB(fqg_acc_merge(S.acc1, buf.data(), used));
      fqg_acc_destroy(D[i].acc);
      fqg_close(D[i].ctx);
    }
    printf("\n");
    fqg_file_stats fs;
    LIB(fqg_acc_read(S.acc1, &fs));
    S.num_reads1 = fs.num_rds;
  }
}

// ---- default and paired modes over several GPUs (FQGPU_DEVICES=0,1,..) --------------------------------------------
// The index modes with the records of a file spread over several contexts: every piece is validated (and its frame
// kept) by whichever context is free, and the names are tested ACROSS the contexts by the fingerprint exchange of
// fq_names_multi.h instead of one device's index.  Findings are ordered as the serial loops order them: per record
// read (truncation), name (wrong header), duplicate / unpaired, validation.
struct MultiDev {
  fqg_ctx* ctx = nullptr;
  fqg_acc* acc = nullptr;
};
struct MultiPass {
  bool stopped = false;       // a NUL at a record start ends the file there: the caller passes again with that limit
  uint64_t stop_offset = 0;   // ... bytes of the (inflated) file in front of that record
  bool probe_printed = false;
  uint64_t n_records = 0;     // records of the pieces looked at
  bool have = false;          // a finding of the validation side (stages 0, 1, 3) - the first in file order
  uint64_t rec = 0;
  int stage = 0;
  fqg_validate_result r{};
  RecordText text;
  Probe pr;
  std::vector<fqhost::NameShard> shards;  // per device: the frames it kept
};

// one pass of all devices over a file.  validate_as: the state the records are validated under (null: the file's
// own, decided from its first record); acc_of_first: the accumulator device 0 adds to
MultiPass multi_pass(const char* path, std::vector<MultiDev>& D, int is_pe, uint32_t flags, const fqg_file_state* validate_as,
                     uint64_t limit = ~0ull, bool print_info = true) {
  MultiPass out;
  out.shards.resize(D.size());
  for (size_t i = 0; i < D.size(); ++i) out.shards[i].ctx = D[i].ctx;
  const size_t piece = multi_piece_bytes();
  struct Done {
    Piece p;
    fqg_validate_result r{};
    int rc = 0;
    std::string err;
  };
  std::map<uint64_t, Done> done;
  std::mutex mu, fetch_mu, shard_mu;
  std::condition_variable cv;
  std::atomic<bool> stop{false};
  bool exhausted = false;        // (under fetch_mu)
  uint64_t n_pieces = ~0ull;     // known once the final piece was handed out (under mu)
  AlignedPieces src(g_ctx, path, piece, multi_slots(D.size()), limit);
  const bool timing = getenv("FQGPU_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto work = [&](size_t di) {
    double t_wait = 0, t_gpu = 0, t_keep = 0;
    uint64_t n = 0;
    struct Report {
      bool on;
      size_t di;
      const double &w, &g, &k;
      const uint64_t& n;
      ~Report() {
        if (on) fprintf(fqhost::diag(), "fqgpu timing: context %zu: %llu pieces; waiting for a piece %.3f s, copy + validate %.3f s, keeping the frame %.3f s\n",
                        di, (unsigned long long)n, w, g, k);
      }
    } report{timing, di, t_wait, t_gpu, t_keep, n};
    for (;;) {
      Done d;
      fqg_file_state st;
      const double t0 = timing ? now() : 0;
      {
        std::lock_guard<std::mutex> lk(fetch_mu);
        if (exhausted || stop || !src.next(&d.p)) {
          exhausted = true;
          return;
        }
        if (d.p.final) exhausted = true;
        probe_piece(out.pr, d.p.data, d.p.size, is_pe);  // piece 0 is handed out first: the state is the first record's
        st = validate_as ? *validate_as : out.pr.st;
      }
      const double t1 = timing ? now() : 0;
      d.rc = fqg_validate(D[di].ctx, D[di].acc, d.p.data, d.p.size, FQG_MEM_HOST, d.p.final ? 1 : 0, &st, flags, &d.r);
      const double t2 = timing ? now() : 0;
      if (d.rc) d.err = fqg_last_error(D[di].ctx);
      else if (!d.p.final && d.r.code == FQG_OK && !d.r.stopped && d.r.consumed != d.p.size) {
        d.rc = FQG_ERR_ARG;
        d.err = "a piece cut at a record boundary was not consumed whole";
      } else if (d.r.n_records) {
        fqg_frame* fr = nullptr;
        d.rc = fqg_frame_retain(D[di].ctx, &fr);
        if (d.rc) d.err = fqg_last_error(D[di].ctx);
        else {
          std::lock_guard<std::mutex> lk(shard_mu);
          out.shards[di].pieces.push_back(fqhost::NameShard::Piece{fr, d.p.first_record, d.r.n_records});
        }
      }
      if (timing) t_wait += t1 - t0, t_gpu += t2 - t1, t_keep += now() - t2, ++n;
      std::lock_guard<std::mutex> lk(mu);
      if (d.p.final) n_pieces = d.p.seq + 1;
      done.emplace(d.p.seq, std::move(d));
      cv.notify_all();
    }
  };
  std::vector<std::thread> th;
  for (size_t i = 0; i < D.size(); ++i) th.emplace_back(work, i);
  auto join_all = [&] {
    stop = true;
    src.abort();
    for (auto& t : th)
      if (t.joinable()) t.join();
  };
  bool info_pending = true;
  for (uint64_t k = 0;; ++k) {
    Done d;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return done.count(k) || k >= n_pieces; });
      if (!done.count(k)) break;
      d = std::move(done[k]);
      done.erase(k);
    }
    const uint64_t base = d.p.first_record;
    const fqg_validate_result& r = d.r;
    if (d.rc) {
      join_all();
      FQ_PRINT_ERROR("GPU library failure in fqg_validate (%d): %s", d.rc, d.err.c_str());
      fqhost::leave(kExitSys);
    }
    int stage = -1;
    if (r.code == FQG_E_TRUNCATED || r.code == FQG_E_LINE_TOO_LONG) stage = 0;
    else if (r.code == FQG_E_HDR1_AT) stage = 1;
    else if (r.code) stage = 3;
    if (info_pending && base == 0 && r.n_records > 0) {
      if (!(stage >= 0 && stage <= 1 && r.record == 0)) {
        if (print_info) print_probe(out.pr);
        out.probe_printed = true;
      }
      info_pending = false;
    }
    if (r.stopped && stage < 0) {
      join_all();
      out.stopped = true;
      out.stop_offset = d.p.stream_offset + r.consumed;
      return out;
    }
    if (stage >= 0) {
      join_all();
      out.have = true;
      out.rec = base + r.record;
      out.stage = stage;
      out.r = r;
      out.text = locate_record(d.p.data, d.p.size, r.record);
      out.n_records = base + r.n_records;
      return out;
    }
    out.n_records = base + r.n_records;
    src.release(d.p);
    if (d.p.final) break;
  }
  join_all();
  return out;
}

void merge_device_stats(std::vector<MultiDev>& D, Stats& S) {
  std::vector<char> buf;
  for (size_t i = 1; i < D.size(); ++i) {
    size_t used = 0;
    if (fqg_acc_export(D[i].acc, nullptr, 0, &used) != 0) die_lib("fqg_acc_export", -1);
    buf.resize(used);
    if (fqg_acc_export(D[i].acc, buf.data(), buf.size(), &used) != 0) die_lib("fqg_acc_export", -1);
    LIB(fqg_acc_merge(S.acc1, buf.data(), used));
    if (fqg_acc_reset(D[i].acc) != 0) die_lib("fqg_acc_reset", -1);
  }
}

void release_shards(std::vector<fqhost::NameShard>& sh) {
  for (auto& s : sh)
    for (auto& p : s.pieces) fqg_frame_release(const_cast<fqg_frame*>(p.frame));
  sh.clear();
}

struct MultiIndexed {
  std::vector<MultiDev> D;
  fqhost::NamesOfFile f1;
};

// file 1 of the default / paired mode.  Always returns true (a file that holds a NUL at a record start is passed over
// once more with the limit the first pass found, below; nothing falls back to the one-device loop any more).
bool run_index_multi(const char* path, int is_pe, Stats& S, IndexedFile& F, MultiIndexed& M, const std::vector<int>& devs) {
  M.D.resize(devs.size());
  M.D[0].ctx = g_ctx;
  M.D[0].acc = S.acc1;
  for (size_t i = 1; i < devs.size(); ++i) {
    const int rc = fqg_open(devs[i], &M.D[i].ctx);
    if (rc != 0) {
      FQ_PRINT_ERROR("FQGPU_DEVICES: device %d is not a usable MI355X GPU (fqg_open: %d)", devs[i], rc);
      fqhost::leave(kExitSys);
    }
    if (fqg_acc_create(M.D[i].ctx, &M.D[i].acc) != 0) die_lib("fqg_acc_create", -1);
  }
  MultiPass pass = multi_pass(path, M.D, is_pe, FQG_VALIDATE_COUNT_TWICE | FQG_VALIDATE_INDEX, nullptr);
  if (pass.stopped) {
    // a NUL byte at a record start ends the file there (src/fastq.c:250): what later pieces added to the accumulators
    // and to the shards does not belong to it - once more, with the file ending where the reference stops reading
    release_shards(pass.shards);
    if (strcmp(path, "-") == 0) {
      FQ_PRINT_ERROR("Error in file %s: a NUL byte at a record start with FQGPU_DEVICES on a stream: use one device", path);
      fqhost::leave(kExitSys);
    }
    for (auto& d : M.D)
      if (fqg_acc_reset(d.acc) != 0) die_lib("fqg_acc_reset", -1);
    pass = multi_pass(path, M.D, is_pe, FQG_VALIDATE_COUNT_TWICE | FQG_VALIDATE_INDEX, nullptr, pass.stop_offset, !pass.probe_printed);
    if (pass.stopped) {
      FQ_PRINT_ERROR("Error in file %s: the file changed while it was read", path);
      fqhost::leave(kExitSys);
    }
  }
  M.f1.shards = std::move(pass.shards);
  M.f1.st = pass.pr.st;
  M.f1.flag = 0;
  fqhost::NamesExchange ex;
  bool dup = false;
  uint64_t dup_rec = 0, name_bytes = 0;
  std::string dup_name;
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: file 1 passed %.3f s after the program started\n", since_start());
  const bool exchanged = ex.first_duplicate(M.f1, &dup, &dup_rec, &dup_name, &name_bytes);
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: names of file 1 exchanged %.3f s after the program started\n", since_start());
  if (!exchanged) {
    FQ_PRINT_ERROR("GPU library failure in the read-name exchange: %s", ex.error.c_str());
    fqhost::leave(kExitSys);
  }
  // which finding does the serial loop hit first?  per record: read (0), name (1), duplicate (2), validation (3)
  uint64_t best = ~0ull;
  int stage = 9;
  if (pass.have) {
    best = pass.rec;
    stage = pass.stage;
  }
  if (dup && (dup_rec < best || (dup_rec == best && 2 < stage))) {
    best = dup_rec;
    stage = 2;
  }
  if (best != ~0ull) {
    ticker(1, best, 100000);
    if (stage == 0) {
      if (pass.r.code == FQG_E_LINE_TOO_LONG) fail_too_long(path, best);
      fail_truncated(path, 4 * best);
    }
    if (stage == 1) fail_wrong_header(path, 4 * (best + 1), pass.text.l[0]);
    if (stage == 2) {
      FQ_PRINT_ERROR("Error in file %s: line %lu: duplicated sequence %s", path, (unsigned long)(4 * (best + 1)), dup_name.c_str());
      fqhost::leave(kExitFormat);
    }
    print_validation_error(path, 4 * (best + 1), pass.r, pass.text);
    fqhost::leave(kExitFormat);
  }
  ticker(1, pass.n_records, 100000);
  merge_device_stats(M.D, S);
  F.st = pass.pr.st;
  F.n_records = pass.n_records;
  F.entries = pass.n_records;
  F.index_mem = 8 + pass.n_records * (16 + 1 + 24) + name_bytes;  // what the reference adds up (src/fastq.c:609, src/fastq_info.c:293)
  return true;
}

// the second file of a pair over the same devices (src/fastq_info.c:322-362)
void run_pair_second_file_multi(const char* path1, const char* path2, Stats& S, IndexedFile& F, MultiIndexed& M) {
  const unsigned long cline1 = 4 * F.n_records;  // fd1->cline stays where indexing left it
  // file-2 records are validated against file 1's state and counters (src/fastq_info.c:345); names under file 2's own
  // (the accumulators as file 1 left them: a second file that ends at a NUL byte is passed over twice)
  std::vector<std::vector<char>> before(M.D.size());
  for (size_t i = 0; i < M.D.size(); ++i) {
    size_t used = 0;
    if (fqg_acc_export(M.D[i].acc, nullptr, 0, &used) != 0) die_lib("fqg_acc_export", -1);
    before[i].resize(used);
    if (fqg_acc_export(M.D[i].acc, before[i].data(), before[i].size(), &used) != 0) die_lib("fqg_acc_export", -1);
  }
  MultiPass pass = multi_pass(path2, M.D, 1, FQG_VALIDATE_INDEX, &F.st);
  if (pass.stopped) {
    release_shards(pass.shards);
    if (strcmp(path2, "-") == 0) {
      FQ_PRINT_ERROR("Error in file %s: a NUL byte at a record start with FQGPU_DEVICES on a stream: use one device", path2);
      fqhost::leave(kExitSys);
    }
    for (size_t i = 0; i < M.D.size(); ++i) {
      if (fqg_acc_reset(M.D[i].acc) != 0) die_lib("fqg_acc_reset", -1);
      if (fqg_acc_merge(M.D[i].acc, before[i].data(), before[i].size()) != 0) die_lib("fqg_acc_merge", -1);
    }
    pass = multi_pass(path2, M.D, 1, FQG_VALIDATE_INDEX, &F.st, pass.stop_offset, !pass.probe_printed);
    if (pass.stopped) {
      FQ_PRINT_ERROR("Error in file %s: the file changed while it was read", path2);
      fqhost::leave(kExitSys);
    }
  }
  fqhost::NamesOfFile f2;
  f2.shards = std::move(pass.shards);
  f2.st = pass.pr.st;
  f2.flag = FQG_FP_FILE2;
  fqhost::NamesExchange ex;
  fqhost::PairingOutcome po;
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: file 2 passed %.3f s after the program started\n", since_start());
  // (mates in one order, every name at its place: nothing to exchange.  FQGPU_NO_POSITIONAL_MATCH=1: always exchange)
  bool by_position = false;
  bool paired_ok = getenv("FQGPU_NO_POSITIONAL_MATCH") ? true : ex.paired_by_position(M.f1, f2, &by_position);
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: names compared by position (%s) %.3f s after the program started\n", ex.why.c_str(), since_start());
  if (paired_ok && by_position) po.matched = pass.n_records;
  else if (paired_ok) paired_ok = ex.pairing(M.f1, f2, &po);
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: names of the two files paired %.3f s after the program started\n", since_start());
  if (!paired_ok) {
    FQ_PRINT_ERROR("GPU library failure in the read-name exchange: %s", ex.error.c_str());
    fqhost::leave(kExitSys);
  }
  uint64_t best = ~0ull;
  int stage = 9;
  if (pass.have) {
    best = pass.rec;
    stage = pass.stage;
  }
  if (po.has_first && (po.first_unpaired < best || (po.first_unpaired == best && 2 < stage))) {
    best = po.first_unpaired;
    stage = 2;
  }
  if (best != ~0ull) {
    ticker(1, best, 100000);
    if (stage == 0) {
      if (pass.r.code == FQG_E_LINE_TOO_LONG) fail_too_long(path2, best);
      fail_truncated(path2, 4 * best);
    }
    if (stage == 1) fail_wrong_header(path2, 4 * (best + 1), pass.text.l[0]);
    if (stage == 2) {
      FQ_PRINT_ERROR("Error in file %s: line %lu: unpaired read - %s", path2, (unsigned long)(4 * (best + 1)), po.first_name.c_str());
      fqhost::leave(kExitFormat);
    }
    print_validation_error(path1, cline1, pass.r, pass.text);  // named after file 1, like the reference
    fqhost::leave(kExitFormat);
  }
  ticker(1, pass.n_records, 100000);
  merge_device_stats(M.D, S);
  printf("\n");
  if (po.leftover > 0) {
    FQ_PRINT_ERROR("Error in file %s: found %llu unpaired reads", path1, (unsigned long long)po.leftover);
    fqhost::leave(kExitFormat);
  }
}

// ---- second file of a pair: src/fastq_info.c:322-362 ----------------------------------------
void run_pair_second_file(const char* path1, const char* path2, Stats& S, IndexedFile& F) {
  Input in(g_ctx, path2, piece_bytes());
  Probe pr2;
  uint64_t base = 0;
  bool info_pending = true;
  const unsigned long cline1 = 4 * F.n_records;  // fd1->cline stays where indexing left it
  while (in.next()) {
    probe_piece(pr2, in.data(), in.size(), 1);
    fqg_validate_result r;
    // file-2 records are validated against file 1's state and counters (src/fastq_info.c:345)
    LIB(fqg_validate(g_ctx, S.acc1, in.data(), in.size(), FQG_MEM_HOST, in.final() ? 1 : 0, &F.st, FQG_VALIDATE_NAMES | in.vflags(), &r));
    fqg_index_result ir{};
    ir.n_entries = F.entries;
    if (r.n_records > 0) LIB(fqg_index_match_delete(g_ctx, F.index, &pr2.st, &ir));
    uint64_t best_rec = ~0ull;
    int best_stage = 9;
    auto offer = [&](uint64_t rec, int stage) {
      if (rec < best_rec || (rec == best_rec && stage < best_stage)) {
        best_rec = rec;
        best_stage = stage;
      }
    };
    if (r.code == FQG_E_TRUNCATED || r.code == FQG_E_LINE_TOO_LONG) offer(r.record, 0);
    else if (r.code == FQG_E_HDR1_AT) offer(r.record, 1);
    else if (r.code) offer(r.record, 3);
    if (ir.code == FQG_E_WRONG_HEADER) offer(ir.record, 1);
    if (ir.code == FQG_E_UNPAIRED) offer(ir.record, 2);
    if (info_pending && base == 0 && r.n_records > 0) {
      if (!(best_rec == 0 && best_stage <= 1)) print_probe(pr2);
      info_pending = false;
    }
    if (best_rec != ~0ull) {
      const uint64_t R = base + best_rec;
      ticker(base + 1, R, 100000);
      const RecordText t = locate_record(in.data(), in.size(), best_rec);
      if (best_stage == 0) {
        if (r.code == FQG_E_LINE_TOO_LONG) fail_too_long(path2, R);
        fail_truncated(path2, 4 * R);
      }
      if (best_stage == 1) fail_wrong_header(path2, 4 * (R + 1), t.l[0]);
      if (best_stage == 2) {
        FQ_PRINT_ERROR("Error in file %s: line %lu: unpaired read - %s", path2, (unsigned long)(4 * (R + 1)),
                       canonical_name(t.l[0], pr2.st).c_str());
        fqhost::leave(kExitFormat);
      }
      print_validation_error(path1, cline1, r, t);  // named after file 1, like the reference
      fqhost::leave(kExitFormat);
    }
    ticker(base + 1, base + r.n_records, 100000);
    base += r.n_records;
    if (r.n_records > 0) F.entries = ir.n_entries;
    if (r.stopped) break;
    if (!in.final()) in.carry_from(r.consumed);
  }
  printf("\n");
  if (F.entries > 0) {
    FQ_PRINT_ERROR("Error in file %s: found %llu unpaired reads", path1, (unsigned long long)F.entries);
    fqhost::leave(kExitFormat);
  }
}

// ---- "pe": validate_interleaved (src/fastq_info.c:57-106) ------------------------------------
void run_interleaved(const char* path, Stats& S) {
  fprintf(stderr, "Paired-end interleaved\n");
  Input in(g_ctx, path, piece_bytes());
  in.next(true);  // pairs are compared inside one frame: the whole file is one image
  Probe pr;
  probe_piece(pr, in.data(), in.size(), 1);
  fqg_validate_result r;
  LIB(fqg_validate(g_ctx, S.acc1, in.data(), in.size(), FQG_MEM_HOST, 1, &pr.st, FQG_VALIDATE_INDEX | in.vflags(), &r));
  const uint64_t n = r.n_records;
  fqg_index_result cr{};
  if (n >= 2) {
    fqg_frame* fr = nullptr;
    LIB(fqg_frame_retain(g_ctx, &fr));
    LIB(fqg_names_compare(g_ctx, fr, &pr.st, nullptr, nullptr, &cr));
    fqg_frame_release(fr);
  }
  // order inside pair k: read m1, read m2, name m1, name m2, names equal, validate m1, validate m2
  uint64_t best_pair = ~0ull;
  int best_stage = 99;
  auto offer = [&](uint64_t pair, int stage) {
    if (pair < best_pair || (pair == best_pair && stage < best_stage)) {
      best_pair = pair;
      best_stage = stage;
    }
  };
  const bool trunc = r.code == FQG_E_TRUNCATED || r.code == FQG_E_LINE_TOO_LONG;
  uint64_t trunc_rec = trunc ? r.record : ~0ull;
  if (trunc) offer(r.record / 2, r.record % 2 == 0 ? 0 : 2);
  else if (r.code == FQG_E_HDR1_AT) offer(r.record / 2, r.record % 2 == 0 ? 3 : 4);
  else if (r.code) offer(r.record / 2, r.record % 2 == 0 ? 6 : 7);
  if (r.tail_lines > 0) {  // an incomplete last record, even when an earlier record has a finding
    offer(n / 2, n % 2 == 0 ? 0 : 2);
    if (trunc_rec == ~0ull) trunc_rec = n;
  } else if (n % 2 == 1) {
    offer(n / 2, 1);  // a first mate without a second one
  }
  if (cr.code == FQG_E_WRONG_HEADER) {
    const RecordText t = locate_record(in.data(), in.size(), cr.record);
    offer(cr.record / 2, (!t.l[0].empty() && t.l[0][0] != '@') ? 3 : 4);
  }
  if (cr.code == FQG_E_UNPAIRED) offer(cr.record / 2, 5);
  // format lines: printed by the first fastq_get_readname call, i.e. for mate 1 of pair 0
  if (n >= 2 && !(best_pair == 0 && best_stage <= 3)) print_probe(pr);
  if (best_pair != ~0ull) {
    ticker(1, best_pair, 50000, 2);
    const uint64_t k = best_pair;
    const unsigned long cline_pair = 4 * (2 * k + 2);
    switch (best_stage) {
      case 0:
      case 2:
        if (r.code == FQG_E_LINE_TOO_LONG) fail_too_long(path, r.record);
        fail_truncated(path, 4 * (best_pair * 2 + (best_stage == 2 ? 1 : 0)));
      case 1:
        FQ_PRINT_ERROR("Error in file %s: line %lu: file truncated?", path, (unsigned long)(4 * n));
        fqhost::leave(kExitFormat);
      case 3:
        fail_wrong_header(path, cline_pair, locate_record(in.data(), in.size(), 2 * k).l[0]);
      case 4:
        fail_wrong_header(path, cline_pair, locate_record(in.data(), in.size(), 2 * k + 1).l[0]);
      case 5:
        FQ_PRINT_ERROR("Error in file %s: line %lu: unpaired read - %s", path, cline_pair,
                       canonical_name(locate_record(in.data(), in.size(), 2 * k).l[0], pr.st).c_str());
        fqhost::leave(kExitFormat);
      default:
        print_validation_error(path, cline_pair, r, locate_record(in.data(), in.size(), r.record));
        fqhost::leave(kExitFormat);
    }
  }
  ticker(1, n / 2, 50000, 2);
  printf("\n");
  fqg_file_stats fs;
  LIB(fqg_acc_read(S.acc1, &fs));
  S.num_reads1 = fs.num_rds;
}

// What fastq_read_entry (src/fastq.c:245-261) returns on the bytes BEHIND a line that started with a NUL byte.  That
// line made the call before return 0 ("no entry"), which ends a loop - but not the file: the paired loop below reads
// once more from each file afterwards (src/fastq_info.c:142-149), and that read starts at the next line.
enum { kBehindNothing = 0, kBehindRecord = 1, kBehindTruncated = 2 };
int read_behind_stop(const char* d, size_t n, size_t at) {
  auto line = [&](size_t& p) {  // consumes a line; false when gzgets hands back an empty string (NUL first, or nothing left)
    if (p >= n) return false;
    const bool text = d[p] != 0;
    const char* nl = static_cast<const char*>(memchr(d + p, '\n', n - p));
    p = nl ? (size_t)(nl - d) + 1 : n;
    return text;
  };
  size_t p = at;
  line(p);  // the NUL line itself
  if (!line(p)) return kBehindNothing;
  const bool seq = line(p), hdr2 = line(p), qual = line(p);
  return (seq && hdr2 && qual) ? kBehindRecord : kBehindTruncated;
}

// ---- -r -s, two files: validate_paired_sorted_fastq_file (src/fastq_info.c:108-152) ---------
void run_paired_sorted(const char* p1, const char* p2, Stats& S) {
  Input in1(g_ctx, p1, piece_bytes()), in2(g_ctx, p2, piece_bytes());
  in1.next(true);
  in2.next(true);
  Probe pr1, pr2;
  probe_piece(pr1, in1.data(), in1.size(), 1);
  probe_piece(pr2, in2.data(), in2.size(), 1);
  fqg_validate_result r1, r2;
  LIB(fqg_validate(g_ctx, S.acc1, in1.data(), in1.size(), FQG_MEM_HOST, 1, &pr1.st, FQG_VALIDATE_INDEX | in1.vflags(), &r1));
  fqg_frame *f1 = nullptr, *f2 = nullptr;
  if (r1.n_records) LIB(fqg_frame_retain(g_ctx, &f1));
  LIB(fqg_validate(g_ctx, S.acc2, in2.data(), in2.size(), FQG_MEM_HOST, 1, &pr2.st, FQG_VALIDATE_INDEX | in2.vflags(), &r2));
  if (r2.n_records) LIB(fqg_frame_retain(g_ctx, &f2));
  fqg_index_result cr{};
  if (f1 && f2) LIB(fqg_names_compare(g_ctx, f1, &pr1.st, f2, &pr2.st, &cr));
  const uint64_t n1 = r1.stopped ? r1.n_records : r1.n_records, n2 = r2.n_records;
  // read failures: a NUL-started line inside the file (reported as the first finding), or an
  // incomplete last record (always known through tail_lines)
  const bool t1 = r1.code == FQG_E_TRUNCATED || r1.code == FQG_E_LINE_TOO_LONG;
  const bool t2 = r2.code == FQG_E_TRUNCATED || r2.code == FQG_E_LINE_TOO_LONG;
  const bool tail1 = r1.tail_lines > 0, tail2 = r2.tail_lines > 0;
  // per index k: read 1, validate 1, read 2, validate 2, names
  enum { READ1 = 0, VAL1 = 1, READ2 = 2, VAL2 = 3, NAMES = 4, END1 = 5, END2 = 6, STOP1 = 7, STOP2 = 8 };
  uint64_t best_k = ~0ull;
  int best_stage = 99, best_kind = -1;
  auto offer = [&](uint64_t k, int stage, int kind) {
    if (k < best_k || (k == best_k && stage < best_stage)) {
      best_k = k;
      best_stage = stage;
      best_kind = kind;
    }
  };
  if (t1) offer(r1.record, 0, READ1);
  else if (r1.code) offer(r1.record, 1, VAL1);
  if (t2) offer(r2.record, 2, READ2);
  else if (r2.code) offer(r2.record, 3, VAL2);
  if (cr.code == FQG_E_NAME_MISMATCH) offer(cr.record, 4, NAMES);
  if (r1.stopped) offer(n1, 0, STOP1);  // a line that starts with a NUL byte: "no entry", the loop ends there
  else if (tail1) offer(n1, 0, READ1);
  else offer(n1, 0, END1);  // clean end of file 1: the loop ends before reading file 2
  if (r2.stopped) offer(n2, 2, STOP2);
  else if (tail2) offer(n2, 2, READ2);
  else offer(n2, 2, END2);  // clean end of file 2, after record n2 of file 1 was handled
  // format lines come from inside validation (src/fastq.c:364) of each file's first record
  const bool reach1 = n1 > 0 && !(r1.code && r1.record == 0 && is_early_code(r1.code));
  if (reach1) print_probe(pr1);
  const bool reach2 = n1 > 0 && !(r1.code && r1.record == 0) && n2 > 0 &&
                      !(r2.code && r2.record == 0 && is_early_code(r2.code));
  if (reach2) print_probe(pr2);
  auto finish_frames = [&]() {
    if (f1) fqg_frame_release(f1);
    if (f2) fqg_frame_release(f2);
  };
  const uint64_t k = best_k;
  switch (best_kind) {
    case READ1:
      if (r1.code == FQG_E_LINE_TOO_LONG) fail_too_long(p1, k);
      fail_truncated(p1, 4 * k);
    case VAL1:
      print_validation_error(p1, 4 * (k + 1), r1, locate_record(in1.data(), in1.size(), k));
      fqhost::leave(kExitFormat);
    case READ2:
      if (r2.code == FQG_E_LINE_TOO_LONG) fail_too_long(p2, k);
      fail_truncated(p2, 4 * k);
    case VAL2:
      print_validation_error(p2, 4 * (k + 1), r2, locate_record(in2.data(), in2.size(), k));
      fqhost::leave(kExitFormat);
    case NAMES:
      FQ_PRINT_ERROR("Readnames do not match across files (read #%ld)", (long)(k + 1 + 1));
      fqhost::leave(kExitFormat);
    case END1:
      // file 1 is exhausted at record n1; anything left in file 2?
      if (n2 > n1) {
        FQ_PRINT_ERROR("Premature end of file1");
        fqhost::leave(kExitFormat);
      }
      if (tail2 && n2 == n1) fail_truncated(p2, 4 * n1);
      break;
    case END2:
      // file 2 is exhausted at record n2 (record n2 of file 1 has been read and validated)
      if (n1 >= n2 + 2) {
        FQ_PRINT_ERROR("Premature end of file2");
        fqhost::leave(kExitFormat);
      }
      if (tail1 && n1 == n2 + 1) fail_truncated(p1, 4 * (n2 + 1));
      break;
    case STOP1: {
      // the loop ended on the NUL line of file 1; the read after the loop goes on behind it
      const int x = read_behind_stop(in1.data(), in1.size(), (size_t)r1.consumed);
      if (x == kBehindRecord) {
        FQ_PRINT_ERROR("Premature end of file2");
        fqhost::leave(kExitFormat);
      }
      if (x == kBehindTruncated) fail_truncated(p1, 4 * n1);
      if (n2 > n1) {  // ... and file 2 still has record n1
        FQ_PRINT_ERROR("Premature end of file1");
        fqhost::leave(kExitFormat);
      }
      if (tail2 && n2 == n1) fail_truncated(p2, 4 * n1);
      break;
    }
    case STOP2: {
      // record n2 of file 1 has been read and validated; the loop ended on the NUL line of file 2
      if (n1 >= n2 + 2) {
        FQ_PRINT_ERROR("Premature end of file2");
        fqhost::leave(kExitFormat);
      }
      if (tail1 && n1 == n2 + 1) fail_truncated(p1, 4 * (n2 + 1));
      const int y = read_behind_stop(in2.data(), in2.size(), (size_t)r2.consumed);
      if (y == kBehindRecord) {
        FQ_PRINT_ERROR("Premature end of file1");
        fqhost::leave(kExitFormat);
      }
      if (y == kBehindTruncated) fail_truncated(p2, 4 * n2);
      break;
    }
  }
  finish_frames();
  printf("\n");
  fqg_file_stats fs;
  LIB(fqg_acc_read(S.acc1, &fs));
  S.num_reads1 = fs.num_rds;
}

}  // namespace

int main(int argc, char** argv) {
  int is_paired_data = 0, is_interleaved = 0, is_sorted = 0, empty_ok = 0, no_encoding_ok = 0, skip_readname_check = 0;
  int nopt = 0, c;
  opterr = 0;
  fqhost::install_counted_output(argv);  // (fq_respawn.h: a run that has to start over does not print anything twice)
  fprintf(stderr, "fastq_utils %s\n", "0.25.3");  // fastq_print_version
  // option handling as in src/fastq_info.c:214-255 (nopt counts option letters)
  while ((c = getopt(argc, argv, "esfrhq")) != -1) switch (c) {
      case 'q': no_encoding_ok = 1; ++nopt; break;
      case 'e': empty_ok = 1; ++nopt; break;
      case 's': is_sorted = 1; ++nopt; break;
      case 'r': skip_readname_check = 1; ++nopt; break;
      case 'h': print_usage(1); fqhost::leave(0);
      case 'f':
        fprintf(stderr, "Fixing (-f) enabled: Replacing . by N (creating .fix.gz files)\n");
        FQ_PRINT_ERROR("-f option is no longer valid.");
        fqhost::leave(kExitParams);
      default:
        ++nopt;
        FQ_PRINT_ERROR("Option -%c invalid", optopt);
        fqhost::leave(kExitParams);
    }
  if (argc - nopt < 2 || argc - nopt > 3) {
    FQ_PRINT_ERROR("Invalid number of arguments");
    print_usage(0);
    fqhost::leave(kExitParams);
  }
  if (argc - nopt == 3) {
    is_paired_data = 1;
    is_interleaved = strncmp(argv[2 + nopt], "pe", 2) == 0;
  }
  const char* file1 = argv[1 + nopt];
  const char* file2 = is_paired_data ? argv[2 + nopt] : nullptr;

  const char* dev = getenv("FQGPU_DEVICE");
  const std::vector<int> devices = devices_from_env();  // FQGPU_DEVICES=0,1,..: the -r pass over several GPUs
  fqhost::keep_slots_until_exit() = true;
  int rc = fqg_open(!devices.empty() ? devices[0] : dev ? atoi(dev) : 0, &g_ctx);
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: context open %.3f s after the program started\n", since_start());
  if (rc != 0) {
    FQ_PRINT_ERROR("no usable MI355X GPU (fqg_open: %d); this build has no CPU path", rc);
    fqhost::leave(kExitSys);
  }
  Stats S;
  LIB(fqg_acc_create(g_ctx, &S.acc1));
  IndexedFile F;
  MultiIndexed multi;
  bool merged_second_file = false;

  if (is_interleaved) {
    run_interleaved(file1, S);
  } else if (is_paired_data && is_sorted && skip_readname_check) {
    fprintf(stderr, "-s option used: assuming that reads have the same ordering in both files\n");
    LIB(fqg_acc_create(g_ctx, &S.acc2));
    run_paired_sorted(file1, file2, S);
    fqg_acc_destroy(S.acc2);
    S.acc2 = nullptr;  // the summary only looks at file 1 (src/fastq_info.c:316-319)
  } else if (!is_paired_data && skip_readname_check) {
    fprintf(stderr, "Skipping check for duplicated read names\n");
    // One GPU, nothing said, a LARGE regular file: the loop over record-aligned pieces with TWO contexts on that GPU - one
    // piece's copy runs beside another's kernels (the 100 M-read file of the bench from tmpfs: 94 Mreads/s against 87,
    // 78 against 73 on a slower box; small files would only pay for the second context).  FQGPU_ONE_CONTEXT=1: the loop of
    // one context, which is also what the index modes, streams and re-framed input (fq_respawn.h) run through.
    std::vector<int> r_devs = devices;
    if (r_devs.empty() && !getenv("FQGPU_ONE_CONTEXT") && !fqhost::reframing()) {
      struct stat sb;
      if (strcmp(file1, "-") != 0 && stat(file1, &sb) == 0 && S_ISREG(sb.st_mode)) {
        unsigned char magic[2] = {0, 0};
        const int fd = open(file1, O_RDONLY);
        const bool gz = fd >= 0 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (fd >= 0) close(fd);
        // (16 GiB plain, 4 GiB gzip'd: on a file of 3.8 GB the second context - 30 ms to open, its slots to pin, both to take
        // down - cost more than it brought, 0.45 - 0.48 s against 0.38 - 0.41; on 12.7 GB it is even, above that it wins)
        if ((uint64_t)sb.st_size >= (gz ? 4ull << 30 : 16ull << 30)) r_devs.assign(2, dev ? atoi(dev) : 0);
      }
    }
    if (r_devs.size() > 1) run_single_noindex_multi(file1, S, r_devs);
    else run_single_noindex(file1, S);
  } else {
    fprintf(stderr, "DEFAULT_HASHSIZE=%lu\n", 39000001ul);
    fprintf(stderr, "Scanning and indexing all reads from %s\n", file1);
    F.lookups = is_paired_data && file2 != nullptr;  // (one file, or "pe": the index is only the uniqueness test)
    if (!(devices.size() > 1 && run_index_multi(file1, is_paired_data, S, F, multi, devices))) {
      multi.D.clear();
      run_index_file(file1, is_paired_data, S, F);
    }
    fprintf(stderr, "Scanning complete.\n");
    S.num_reads1 = F.entries;
    fprintf(stderr, "\n");
    fprintf(stderr, "Reads processed: %llu\n", (unsigned long long)F.entries);
    fprintf(stderr, "Memory used in indexing: ~%ld MB\n", (long)(F.index_mem / 1024 / 1024));
  }
  if (S.num_reads1 == 0) {
    if (empty_ok) {
      fprintf(stdout, "Number of reads: %lu\n", 0L);
      fprintf(stdout, "Quality encoding range: %lu %lu\n", 0L, 0L);
      fprintf(stdout, "Quality encoding: %s\n", "");
      fprintf(stdout, "Read length: %lu %lu %u\n", 0L, 0L, 0);
      fqhost::leave(0);
    }
    FQ_PRINT_ERROR("No reads found in %s.", file1);
    fqhost::leave(kExitFormat);
  }
  // min/max lengths and qualities are taken BEFORE the second file goes through fd1
  // (src/fastq_info.c:316-319); only the length histogram keeps growing
  fqg_file_stats fs;
  LIB(fqg_acc_read(S.acc1, &fs));
  if (is_paired_data && !is_interleaved && !is_sorted) {
    fprintf(stderr, "File %s processed\n", file1);
    fprintf(stderr, "Next file %s\n", file2);
    if (!multi.D.empty()) run_pair_second_file_multi(file1, file2, S, F, multi);
    else run_pair_second_file(file1, file2, S, F);
    // fd2's own counters are never touched (file-2 records went through fd1): an untouched
    // accumulator stands in for it in min/max and in median_rl()
    LIB(fqg_acc_create(g_ctx, &S.acc2));
    merged_second_file = true;
  }
  unsigned long min_rl = fs.min_rl, max_rl = fs.max_rl, min_qual = fs.min_qual, max_qual = fs.max_qual;
  if (merged_second_file) {
    fqg_file_stats f2s;
    LIB(fqg_acc_read(S.acc2, &f2s));
    min_rl = std::min<unsigned long>(f2s.min_rl, min_rl);
    max_rl = std::max<unsigned long>(f2s.max_rl, max_rl);
    min_qual = std::min<unsigned long>(f2s.min_qual, min_qual);
    max_qual = std::max<unsigned long>(f2s.max_qual, max_qual);
  }
  fprintf(stderr, "------------------------------------\n");
  fprintf(stderr, "Number of reads: %lu\n", S.num_reads1);
  const char* enc = qual_range_to_enc(min_qual, max_qual);
  if (!enc && !no_encoding_ok) {
    if (max_qual > FQG_MAX_PHRED_QUAL)
      FQ_PRINT_ERROR("Unable to determine quality encoding - unknown range [%lu,>%u]", min_qual, FQG_MAX_PHRED_QUAL);
    else
      FQ_PRINT_ERROR("Unable to determine quality encoding - unknown range [%lu,%lu]", min_qual, max_qual);
    fqhost::leave(kExitFormat);
  }
  fprintf(stderr, "Quality encoding range: %lu %lu\n", min_qual, max_qual);
  if (!enc) fprintf(stderr, "Quality encoding: NA\n");
  else fprintf(stderr, "Quality encoding: %s\n", enc);
  uint64_t med = 0;
  LIB(fqg_acc_median(S.acc1, S.acc2, &med));
  fprintf(stderr, "Read length: %lu %lu %u\n", min_rl - 1, max_rl - 1, (unsigned)(med - 1));
  fprintf(stderr, "OK\n");
  if (getenv("FQGPU_TIMING")) fprintf(fqhost::diag(), "fqgpu timing: summary printed %.3f s after the program started\n", since_start());
  if (const char* jm = getenv("FQGPU_JSON_METRICS")) {
    // SURVEY 5 "metrics": the machine-readable twin of the summary above, as an EXTRA that leaves the command line and
    // both output streams as the reference has them - one JSON object written to the file the variable names
    if (FILE* jf = fopen(jm, "w")) {
      const double secs = since_start();
      const unsigned long long reads = (unsigned long long)S.num_reads1;  // (what "Number of reads" says)
      const unsigned long long bytes = fqhost::bytes_handed_out().load();
      fprintf(jf, "{\"program\": \"fastq_info\", \"reads\": %llu, \"input_bytes\": %llu, \"seconds\": %.6f, "
                  "\"Mreads_per_s\": %.3f, \"GB_per_s\": %.3f, \"devices\": %zu, \"min_quality\": %lu, \"max_quality\": %lu, "
                  "\"min_read_length\": %lu, \"max_read_length\": %lu, \"median_read_length\": %u}\n",
              reads, bytes, secs, secs > 0 ? (double)reads / secs / 1e6 : 0.0, secs > 0 ? (double)bytes / secs / 1e9 : 0.0,
              std::max<size_t>(devices.size(), 1), min_qual, max_qual, min_rl - 1, max_rl - 1, (unsigned)(med - 1));
      fclose(jf);
    }
  }
  // everything is said and nothing is open for writing: leave without the HIP runtime's tear-down (0.1 s)
  fflush(stdout);
  fflush(stderr);
  _exit(0);
}
