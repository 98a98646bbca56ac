// fq_input.h - host side of the drop-in programs: reading (optionally gzipped) FASTQ files into
// pinned staging buffers, piece by piece, with the incomplete tail of one piece carried into the
// next.  Decompression stays on the host (zlib), as in the reference (src/fastq.c:631-661).
//
// Reading runs AHEAD of the GPU: a producer thread fills a ring of pinned slots while the caller has
// the previous piece copied to the device and validated (what the reference does serially with four
// gzgets per record, src/fastq.c:245-261).  A plain (not gzipped) regular file is read with pread() by
// several threads at once - 50 Mreads/s of 150 bp reads are 17.5 GB/s, more than one core copies.  A gzip
// file is inflated on every core the process may use: a bgzip'd one block by block (read_bgzf), any other one by
// chunks whose first blocks are searched for (fq_pgzip.h); stdin and small files by one zlib thread - all ahead of the GPU.
//
// Layout of a slot: [ headroom | raw bytes ].  The producer writes raw file bytes behind the headroom
// without knowing where the previous piece's last complete record ended; the consumer learns that from
// the validation of the previous piece and copies the few carried bytes in FRONT of the raw bytes.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fqg.h"
#include "fq_parallel.h"
#include "fq_pgzip.h"
#include "fq_reframe.h"
#include "fq_respawn.h"

namespace fqhost {

// src/fastq.h:68-80
#define FQ_PRINT_ERROR(...)       \
  do {                            \
    fprintf(stderr, "\nERROR: "); \
    fprintf(stderr, __VA_ARGS__); \
    fprintf(stderr, "\n");        \
  } while (0)
constexpr int kExitParams = 1, kExitSys = 2, kExitFormat = 3;

// The programs run one job and exit: un-pinning hundreds of megabytes of staging slots first (0.1 - 0.2 s) buys nothing -
// the operating system takes the memory back.  Set once by a program's main(); library-style users leave it alone.
// A slot that outlives its owner this way is NOT lost: it goes to a process-wide pool and the next reader (the second
// pass over a file, the second file of a pair, the one-device loop after a multi-device attempt) takes it over instead
// of pinning a new one - a program holds as many pinned slots as its busiest reader needs, however many readers it
// builds (the slots are portable pinned memory: any device's context may copy from them).
inline bool& keep_slots_until_exit() {
  static bool v = false;
  return v;
}

// input bytes handed to the GPU so far (every piece an Input gives out, carried tails not counted twice): what the
// machine-readable metrics of a program are made of (FQGPU_JSON_METRICS)
inline std::atomic<unsigned long long>& bytes_handed_out() {
  static std::atomic<unsigned long long> v{0};
  return v;
}

class SlotPool {
 public:
  static SlotPool& get() {
    static SlotPool* p = new SlotPool;  // (never destroyed: reader threads may still hold slots when the program leaves)
    return *p;
  }
  // pinned memory of at least `bytes`; nullptr when the allocation fails
  char* take(fqg_ctx* ctx, size_t bytes) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].second >= bytes && free_[i].second <= bytes + bytes / 2 + (1u << 20) &&
            (best == free_.size() || free_[i].second < free_[best].second))
          best = i;
      if (best != free_.size()) {
        char* p = free_[best].first;
        free_.erase(free_.begin() + (long)best);
        return p;
      }
    }
    char* p = static_cast<char*>(fqg_host_alloc(ctx, bytes));
    if (p) {
      std::lock_guard<std::mutex> lk(mu_);
      size_[p] = bytes;
    }
    return p;
  }
  // the owner is done with it: back to the pool while the program keeps its slots, freed otherwise
  void give(fqg_ctx* ctx, char* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu_);
    auto it = size_.find(p);
    if (keep_slots_until_exit() && it != size_.end()) {
      free_.emplace_back(p, it->second);
      return;
    }
    if (it != size_.end()) size_.erase(it);
    fqg_host_free(ctx, p);
  }
  size_t pinned_bytes() {  // (tests) everything this pool has handed out or holds
    std::lock_guard<std::mutex> lk(mu_);
    size_t t = 0;
    for (auto& kv : size_) t += kv.second;
    return t;
  }

 private:
  std::mutex mu_;
  std::vector<std::pair<char*, size_t>> free_;
  std::map<char*, size_t> size_;
};
// The piece size for a regular file of `file_bytes` bytes on disk: a slot for a small file need not have the size of a
// piece - pinning memory takes as long as filling it (128 MiB: 22 ms of every start, 512 MiB: 90 ms, per input file).
// A compressed file is given what 24 times its size inflates to, at least 1 MiB; one that inflates to more comes in
// several pieces, like any large file.  FQGPU_CHUNK_MB set: that size, exactly (the tests cut files where they want).
inline size_t piece_for_file(size_t piece, uint64_t file_bytes, bool compressed) {
  if (getenv("FQGPU_CHUNK_MB")) return piece;
  const uint64_t may = compressed ? file_bytes * 24 : file_bytes;
  const uint64_t mib = 1u << 20;
  const uint64_t want = std::max<uint64_t>(mib, (may + mib) & ~(mib - 1));
  return (size_t)std::min<uint64_t>(piece, want);
}
inline char* slot_alloc(fqg_ctx* ctx, size_t bytes) { return SlotPool::get().take(ctx, bytes); }
inline void slot_release(fqg_ctx* ctx, char* p) { SlotPool::get().give(ctx, p); }

// How the programs leave: with everything they said flushed, and WITHOUT exit()'s hooks.  The HIP runtime tears itself
// down in one of them, and it must not meet a thread of ours that is still inside a HIP call (a reader pinning its next
// slot while the main thread has found the file's first error): that is a crash after the error message, i.e. a wrong
// exit status.  Nothing is lost: outputs are closed by those who write them before they leave.
[[noreturn]] inline void leave(int code) {
  fflush(stdout);
  fflush(stderr);
  if (getenv("FQGPU_PLAIN_EXIT")) exit(code);  // (tools/exit_stress.py: does the process survive exit()'s hooks?)
  _exit(code);
}

inline unsigned host_read_threads() {
  if (const char* e = getenv("FQGPU_HOST_THREADS")) return (unsigned)std::max(1L, strtol(e, nullptr, 10));
  const unsigned hw = std::thread::hardware_concurrency();
  // (a dozen copy a tmpfs file faster than PCIe takes it; more of them only compete with the DMA for host memory:
  // 8 / 16 / 32 / 64 threads -> 1.06 / 1.16 / 1.24 / 1.46 s for the 100 M-read file of the bench)
  return std::max(1u, std::min(12u, hw ? hw : 1u));
}

// (ReaderPool - a few threads that stay around - lives in fq_parallel.h)

// gzip files below this size stay with one zlib thread (FQGPU_PGZIP_MIN: tests send tiny files through the chunked reader)
inline uint64_t pgzip_min_bytes() {
  if (const char* e = getenv("FQGPU_PGZIP_MIN")) return (uint64_t)std::max(0L, atol(e));
  return 1u << 20;
}

// The many-core reader for the gzip file open on fd (fq_pgzip.h), or nothing when one zlib thread is to read it: small
// files, a single usable core, FQGPU_NO_PARALLEL_INFLATE.  (What that reader does not want to decide it leaves to one
// zlib stream of its own, so every file gzopen reads is read.)
inline std::unique_ptr<ParallelGunzip> open_pgzip(int fd, uint64_t size, const char* path) {
  if (size < pgzip_min_bytes() || host_threads() <= 1 || getenv("FQGPU_NO_PARALLEL_INFLATE")) return nullptr;
  const unsigned T = std::min(host_threads(), 64u);
  // (tools/pgzip_scan.sh on the 16-core share of an EPYC 9575F: 2.6 / 3.0 / 3.5 GB/s inflated with chunks of 1 / 2 / 4 MiB)
  size_t chunk = std::max<size_t>(512u << 10, std::min<size_t>(4u << 20, (128u << 20) / T));
  chunk = std::min<size_t>(chunk, std::max<size_t>((size_t)size / T, 128u << 10));
  if (const char* e = getenv("FQGPU_PGZIP_CHUNK")) chunk = (size_t)std::max(4096L, atol(e));
  return std::unique_ptr<ParallelGunzip>(new ParallelGunzip(fd, size, path, T, chunk));
}
inline void pgzip_report(const ParallelGunzip* pg, const std::string& path) {
  if (!pg || !(getenv("FQGPU_PGZIP_DEBUG") || getenv("FQGPU_TIMING"))) return;
  const ParallelGunzip::Stats& st = pg->stats();
  fprintf(fqhost::diag(), "fqgpu timing: %s inflated by chunks: %llu rounds, %llu chunks joined, %llu without a block start, %llu wrong guesses, "
          "%llu members%s%s; reading %.3f s, finding + inflating %.3f s, joining %.3f s, markers -> bytes + CRC-32 %.3f s\n",
          path.c_str(), (unsigned long long)st.batches, (unsigned long long)st.chunks_joined, (unsigned long long)st.chunks_not_found,
          (unsigned long long)st.chunks_discarded, (unsigned long long)st.members, st.fell_back ? "; one zlib stream from: " : "",
          st.fell_back ? st.why.c_str() : "", st.s_load, st.s_decode, st.s_join + st.s_windows, st.s_narrow);
}

// Readers that run ahead of the GPU, and exit().  A program that links the per-record library (libfastq_gpu.so under the
// reference's own main()) leaves through exit() whenever it likes - on its first finding, say, while a producer thread is
// pinning or filling the next slot, i.e. is INSIDE a HIP call.  exit() runs the HIP runtime's own teardown from one of
// its hooks; a thread of ours inside the runtime at that moment is a crash after everything has been said (a wrong exit
// status).  So the hooks stop the readers first: every Input with a live producer is registered here, and the handler
// - registered with atexit() when the first of them starts, i.e. AFTER the runtime was initialised by fqg_open, and
// therefore run BEFORE the runtime's handlers (exit() runs them last-registered first) - tells them to stop and joins
// them.  The drop-in programs themselves leave through _exit() (leave(), above) and never get here.
class ExitQuiesce {
 public:
  typedef void (*StopFn)(void*);
  static ExitQuiesce& get() {
    static ExitQuiesce* p = new ExitQuiesce;  // (never destroyed: it is used from an exit handler)
    return *p;
  }
  void add(void* who, StopFn stop) {
    std::lock_guard<std::mutex> lk(mu_);
    live_.emplace_back(who, stop);
    if (!hooked_) {
      hooked_ = true;
      // (on_exit, glibc: the handler learns the status exit() was called with - it needs it when it gives up waiting)
      on_exit([](int status, void*) { ExitQuiesce::get().stop_all(status); }, nullptr);
    }
  }
  void remove(void* who) {
    std::lock_guard<std::mutex> lk(mu_);
    for (size_t i = 0; i < live_.size(); ++i)
      if (live_[i].first == who) {
        live_.erase(live_.begin() + (long)i);
        return;
      }
  }
  // Stops and joins every live reader - for at most kPatience: a reader blocked in read() / gzread() on a pipe that
  // has stalled (stdin, a slow upstream) only sees its stop flag when the read returns, and a program that leaves on its
  // first finding must not hang in exit() behind it.  When the patience runs out the process ends at once with the
  // status it was leaving with (_exit: no further handlers - the runtime's teardown is exactly what must not run beside
  // a thread that is still inside a HIP call).
  void stop_all(int status) {
    std::vector<std::pair<void*, StopFn>> all;
    {
      std::lock_guard<std::mutex> lk(mu_);
      all.swap(live_);
    }
    if (all.empty()) return;
    struct Wait {
      std::mutex mu;
      std::condition_variable cv;
      bool done = false;
    };
    auto w = std::make_shared<Wait>();
    std::thread t([all, w] {
      for (auto& e : all) e.second(e.first);
      {
        std::lock_guard<std::mutex> lk(w->mu);
        w->done = true;
      }
      w->cv.notify_all();
    });
    {
      std::unique_lock<std::mutex> lk(w->mu);
      if (!w->cv.wait_for(lk, std::chrono::seconds(kPatienceSeconds), [&] { return w->done; })) {
        fflush(nullptr);
        _exit(status);
      }
    }
    t.join();
  }
  static constexpr int kPatienceSeconds = 3;

 private:
  std::mutex mu_;
  std::vector<std::pair<void*, StopFn>> live_;
  bool hooked_ = false;
};

class Input {
 public:
  Input(fqg_ctx* ctx, const char* path, size_t piece_bytes) : ctx_(ctx), path_(path), cap_(piece_bytes) {
    // fastq_open, src/fastq.c:631-661
    if (path_ == "-") gz_ = gzdopen(fileno(stdin), "rb");
    else {
      // a regular file that does not start with the gzip magic is what zlib would pass through unchanged
      const int fd = open(path, O_RDONLY);
      struct stat sb;
      if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
        unsigned char magic[18];
        memset(magic, 0, sizeof(magic));
        const ssize_t got = pread(fd, magic, sizeof(magic), 0);
        if (!(got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
          plain_fd_ = fd;
          plain_size_ = (uint64_t)sb.st_size;
        } else if (cap_ = piece_for_file(cap_, (uint64_t)sb.st_size, true);
                   got == 18 && bgzf_block_size(magic, 18) > 0 && !getenv("FQGPU_NO_PARALLEL_INFLATE")) {
          // bgzip'd FASTQ: a sequence of small gzip members that say how long they are (SAM/BAM specification 4.1) -
          // inflated on all cores (read_bgzf below) instead of by one zlib thread
          bgzf_fd_ = fd;
          bgzf_size_ = (uint64_t)sb.st_size;
        } else if ((pgz_ = open_pgzip(fd, (uint64_t)sb.st_size, path))) {
          pgz_fd_ = fd;  // any other gzip file of some size: chunks of it are inflated side by side
        }
      }
      if (plain_fd_ < 0 && bgzf_fd_ < 0 && pgz_fd_ < 0) {
        if (fd >= 0) close(fd);
        gz_ = gzopen(path, "r");
      }
    }
    if (!gz_ && plain_fd_ < 0 && bgzf_fd_ < 0 && pgz_fd_ < 0) {
      FQ_PRINT_ERROR("Unable to open %s", path);
      leave(kExitParams);
    }
    if (gz_) gzbuffer(gz_, 1 << 20);
    if (plain_fd_ >= 0 && plain_size_ < cap_) cap_ = std::max<size_t>(plain_size_, 1);  // small file: one small slot
    if (bgzf_fd_ >= 0) cap_ = std::max<size_t>(cap_, 1u << 17);  // (whole blocks of up to 64 KiB are inflated into a slot)
    // The reference's gzgets limits (fq_reframe.h).  Inflated input and stdin pass through one thread anyway: it cuts
    // as it goes (a memchr per line beside the inflate).  A plain file is read by many threads and handed over as it
    // is; the GPU reports a line beyond the limits (FQG_E_LINE_TOO_LONG) and the program starts over with
    // FQGPU_REFRAME set (fq_respawn.h), which brings it here.
    reframe_ = ((gz_ != nullptr || bgzf_fd_ >= 0 || pgz_fd_ >= 0) && reframe_supported()) || reframing();
  }
  // the producer is told to stop and joined (the destructor; exit(): ExitQuiesce)
  void stop_reading() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    if (producer_.joinable() && producer_.get_id() != std::this_thread::get_id()) producer_.join();
  }
  ~Input() {
    ExitQuiesce::get().remove(this);
    stop_reading();
    if (gz_) gzclose(gz_);
    if (plain_fd_ >= 0) close(plain_fd_);
    if (bgzf_fd_ >= 0) close(bgzf_fd_);
    pgzip_report(pgz_.get(), path_);
    pgz_.reset();
    if (pgz_fd_ >= 0) close(pgz_fd_);
    free(bz_raw_);
    for (Slot& s : slots_) slot_release(ctx_, s.buf);
    slot_release(ctx_, whole_);
    slot_release(ctx_, big_);
  }
  Input(const Input&) = delete;
  Input& operator=(const Input&) = delete;

  // Next piece: the carried tail of the previous one followed by fresh bytes.  Returns false once
  // the final piece has been handed out.  An empty file yields one empty, final piece.
  bool next(bool whole_file = false) {
    if (finished_) return false;
    if (whole_file) return next_whole();
    if (!producer_.joinable()) {
      ExitQuiesce::get().add(this, [](void* in) { static_cast<Input*>(in)->stop_reading(); });
      producer_ = std::thread([this] { produce(); });
    }
    const int prev = cur_;
    // carried bytes of the piece the caller is done with
    const char* carry_src = nullptr;
    size_t carry = 0;
    if (prev >= 0 && have_carry_) {
      carry_src = data_ + carry_at_;
      carry = len_ - carry_at_;
    }
    have_carry_ = false;
    const int want = (prev + 1) % kSlots;
    Slot& s = slots_[want];
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return s.ready || failed_; });
      if (failed_) {
        FQ_PRINT_ERROR("%s.\n", fail_msg_.c_str());
        leave(kExitSys);
      }
    }
    if (carry > s.head) {  // a tail longer than the headroom (a record of megabases): rebuild this one piece
      char* nb = alloc(carry + s.len + 1);
      memcpy(nb, carry_src, carry);
      memcpy(nb + carry, s.buf + s.head, s.len);
      slot_release(ctx_, big_);
      big_ = nb;
      data_ = nb;
    } else {
      if (carry) memcpy(s.buf + s.head - carry, carry_src, carry);
      data_ = s.buf + s.head - carry;
    }
    len_ = carry + s.len;
    bytes_handed_out() += s.len;
    eof_ = s.last;
    if (prev >= 0) {  // the previous slot may be refilled
      std::lock_guard<std::mutex> lk(mu_);
      slots_[prev].ready = false;
      cv_.notify_all();
    }
    cur_ = want;
    if (eof_) finished_ = true;
    return true;
  }
  // keep bytes [consumed, size) for the next piece (only meaningful for non-final pieces); the bytes of
  // the current piece stay readable until the next call of next()
  void carry_from(size_t consumed) {
    carry_at_ = consumed;
    have_carry_ = true;
    if (whole_mode_) whole_carry_ = len_ - consumed;
  }
  void stop() { finished_ = true; }
  const char* data() const { return data_; }
  size_t size() const { return len_; }
  bool final() const { return eof_; }
  // what fqg_validate must be told about this input's pieces
  uint32_t vflags() const { return reframe_ ? FQG_VALIDATE_REFRAMED : 0u; }
  // bytes of a plain (uncompressed, seekable) input, 0 when unknown: a size hint for whoever sizes tables from it
  uint64_t plain_bytes() const { return plain_fd_ >= 0 ? plain_size_ : 0; }
  const std::string& path() const { return path_; }

 private:
  static constexpr int kSlots = 3;
  static constexpr size_t kHead = 8u << 20;  // > the longest record the reference's line buffers admit
  struct Slot {
    char* buf = nullptr;
    size_t head = 0, len = 0;
    bool ready = false, last = false, allocated = false;
  };

  // bytes a slot holds behind its headroom: the piece, and in re-framing mode the two bytes per cut on top of a piece
  // of at least kReframeMin bytes (the bytes held back from one round to the next stay below the longest limit)
  static constexpr size_t kReframeMin = 4u << 20;
  size_t slot_room() const {
    if (!reframe_) return cap_;
    const size_t c = std::max(cap_, kReframeMin);
    return c + c / 256 + 64;
  }
  char* alloc(size_t n) {
    char* p = slot_alloc(ctx_, n);
    if (!p) {
      FQ_PRINT_ERROR("unable to allocate %zu bytes of pinned memory", n);
      leave(kExitSys);
    }
    return p;
  }

  // one read of up to `want` bytes behind what the slot holds; false at end of input
  size_t read_gz(char* dst, size_t want, bool* at_end) {
    size_t len = 0;
    while (len < want) {
      const size_t ask = std::min<size_t>(want - len, 1u << 30);
      const int got = gzread(gz_, dst + len, (unsigned)ask);
      if (got < 0) {
        int en = 0;
        std::lock_guard<std::mutex> lk(mu_);
        fail_msg_ = gzerror(gz_, &en);
        failed_ = true;
        cv_.notify_all();
        return len;
      }
      if (got == 0) {
        *at_end = true;
        return len;
      }
      len += (size_t)got;
    }
    const int c = gzgetc(gz_);  // a file that ends exactly where the buffer does
    if (c < 0) *at_end = true;
    else gzungetc(c, gz_);
    return len;
  }
  size_t read_plain(char* dst, size_t want, bool* at_end) {
    const uint64_t left = plain_size_ - plain_off_;
    const size_t len = (size_t)std::min<uint64_t>(want, left);
    const unsigned T = (unsigned)std::min<uint64_t>(host_read_threads(), std::max<uint64_t>(1, len >> 22));
    if (T > 1 && !pool_) pool_.reset(new ReaderPool(host_read_threads()));
    std::atomic<bool> bad{false};
    auto part = [&](unsigned t) {
      const size_t a = (len * t / T) & ~(size_t)4095, b = t + 1 == T ? len : (len * (t + 1) / T) & ~(size_t)4095;
      size_t done = a;
      while (done < b) {
        const ssize_t got = pread(plain_fd_, dst + done, b - done, (off_t)(plain_off_ + done));
        if (got <= 0) {
          bad = true;
          return;
        }
        done += (size_t)got;
      }
    };
    if (T <= 1) part(0);
    else pool_->run(T, part);
    if (bad) {
      std::lock_guard<std::mutex> lk(mu_);
      fail_msg_ = "read error";
      failed_ = true;
      cv_.notify_all();
    }
    plain_off_ += len;
    if (plain_off_ >= plain_size_) *at_end = true;
    return len;
  }

  // The reference's reads return whole lines as long as no line reaches the smallest of its buffers: then there is nothing
  // to cut, whichever call reads which line, and all the line reader's state needs is the number of lines (fq_reframe.h).
  // That is every ordinary file, and looking for the longest line is work for many threads - the walk line by line on the
  // one thread that feeds the GPU was a sixth of the time a gzip'd file of 100 M reads took.  true: raw[0, *taken) stands
  // as it is (everything but an unfinished last line; everything when at_end) and the reader's state is up to date.
  bool short_lines_only(const char* raw, size_t n, bool at_end, size_t* taken) {
    const size_t limit = Reframer::room(0);  // 999: a line of that many bytes with its newline still comes back whole
    if (n < (1u << 20)) return false;        // (small pieces: the one thread is as fast as asking the others)
    if (!scan_pool_) scan_pool_.reset(new ReaderPool(std::min(host_threads(), 32u)));
    const unsigned T = (unsigned)std::min<size_t>(scan_pool_->size(), n >> 20);
    struct Part {
      size_t first = 0, last = 0, count = 0, longest = 0;  // first / last newline (when count), longest stretch between two of them
    };
    std::vector<Part> parts(T);
    scan_pool_->run(T, [&](unsigned t) {
      const size_t a = n * t / T, b = n * (t + 1) / T;
      Part& p = parts[t];
      size_t at = a;
      while (at < b) {
        const char* nl = static_cast<const char*>(memchr(raw + at, '\n', b - at));
        if (!nl) break;
        const size_t q = (size_t)(nl - raw);
        if (!p.count) p.first = q;
        else p.longest = std::max(p.longest, q - p.last);
        p.last = q;
        ++p.count;
        at = q + 1;
      }
    });
    // the stretches that cross from one thread's part into the next, the first line and the unfinished last one
    size_t prev_nl = (size_t)-1, lines = 0, longest = 0;  // (position of the newline in front of the current line; -1: raw's first byte starts it)
    for (const Part& p : parts) {
      if (!p.count) continue;
      longest = std::max(longest, std::max(p.longest, p.first - prev_nl));  // (unsigned: first - (-1) = first + 1 = the line with its newline)
      prev_nl = p.last;
      lines += p.count;
    }
    const size_t tail = n - (prev_nl + 1);  // bytes behind the last newline
    if (longest > limit || tail >= limit) return false;
    rf_.phase = (unsigned)((rf_.phase + lines) & 3u);
    *taken = at_end ? n : prev_nl + 1;
    return true;
  }

  size_t read_some(char* dst, size_t want, bool* at_end) {
    if (plain_fd_ >= 0) return read_plain(dst, want, at_end);
    if (bgzf_fd_ >= 0) return read_bgzf(dst, want, at_end);
    if (pgz_) return read_pgz(dst, want, at_end);
    return read_gz(dst, want, at_end);
  }
  size_t read_pgz(char* dst, size_t want, bool* at_end) {
    const size_t len = pgz_->read(dst, want, at_end);
    if (pgz_->failed()) {  // (zlib's text, as gzerror gives it)
      std::lock_guard<std::mutex> lk(mu_);
      fail_msg_ = pgz_->error();
      failed_ = true;
      cv_.notify_all();
    }
    return len;
  }

  // ---- BGZF input (bgzip'd FASTQ; SAM/BAM specification 4.1) ------------------------------------------------------
  // total size of the block that starts at p when p is a BGZF block header (gzip member, FEXTRA with the 'B' 'C'
  // subfield), 0 otherwise
  static size_t bgzf_block_size(const unsigned char* p, size_t avail) {
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = p[10] | ((size_t)p[11] << 8);
    size_t q = 12;
    while (q + 4 <= 12 + xlen && q + 4 <= avail) {
      const size_t slen = p[q + 2] | ((size_t)p[q + 3] << 8);
      if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= avail) {
        const size_t bsize = (p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
        return bsize >= 12 + xlen + 8 ? bsize : 0;
      }
      q += 4 + slen;
    }
    return 0;
  }
  // up to `want` inflated bytes: compressed bytes are read in pieces of 32 MiB, the blocks in them are listed
  // (their sizes are in their headers and trailers) and every block is inflated to its own place, many at a time.
  // Whole blocks only: fewer than 64 KiB short of `want` is "full".
  size_t read_bgzf(char* dst, size_t want, bool* at_end) {
    struct Block {
      size_t at, size, xlen, out_at, isize;
    };
    size_t len = 0;
    auto fail = [&](const char* what) {
      std::lock_guard<std::mutex> lk(mu_);
      fail_msg_ = what;
      failed_ = true;
      cv_.notify_all();
    };
    for (;;) {
      // refill the compressed window [bz_at_, bz_buf_.size()): what is left of it to the front, then up to 128 MiB of
      // the file behind it, read by the pool (one thread reads a tmpfs file at a few GB/s - less than the pool inflates)
      if (bz_buf_.size() - bz_at_ < (1u << 17) && bgzf_off_ < bgzf_size_) {
        if (!inflate_pool_) {
          inflate_pool_.reset(new ReaderPool(host_threads()));  // (fq_parallel.h: the cores this process may use)
        }
        const size_t old = bz_buf_.size() - bz_at_, add = (size_t)std::min<uint64_t>(128u << 20, bgzf_size_ - bgzf_off_);
        if (bz_raw_cap_ < old + add) {  // (plain memory, never zero-filled: a vector's resize would write it first)
          unsigned char* nb = static_cast<unsigned char*>(malloc(old + (128u << 20)));
          if (!nb) {
            fail("out of memory");
            return len;
          }
          if (old) memcpy(nb, bz_buf_.data() + bz_at_, old);
          free(bz_raw_);
          bz_raw_ = nb;
          bz_raw_cap_ = old + (128u << 20);
        } else if (old) memmove(bz_raw_, bz_buf_.data() + bz_at_, old);
        const unsigned T = (unsigned)std::min<size_t>(inflate_pool_->size(), std::max<size_t>(1, add >> 22));
        std::atomic<bool> bad_read{false};
        inflate_pool_->run(T, [&](unsigned t) {
          const size_t a = (add * t / T) & ~(size_t)4095, b = t + 1 == T ? add : (add * (t + 1) / T) & ~(size_t)4095;
          size_t done = a;
          while (done < b) {
            const ssize_t got = pread(bgzf_fd_, bz_raw_ + old + done, b - done, (off_t)(bgzf_off_ + done));
            if (got <= 0) {
              bad_read = true;
              return;
            }
            done += (size_t)got;
          }
        });
        if (bad_read) {
          fail("read error");
          return len;
        }
        bz_buf_ = Span{bz_raw_, old + add};
        bz_at_ = 0;
        bgzf_off_ += add;
      }
      if (bz_at_ == bz_buf_.size()) {
        *at_end = true;
        return len;
      }
      std::vector<Block> blocks;
      size_t p = bz_at_, total = 0;
      while (p < bz_buf_.size()) {
        const size_t bsize = bgzf_block_size(bz_buf_.data() + p, bz_buf_.size() - p);
        if (!bsize) {
          if (bz_buf_.size() - p < 18 && bgzf_off_ < bgzf_size_) break;  // a header cut by the window: next round
          fail("not a BGZF block where one was expected (a bgzip'd file followed by other data?)");
          return len;
        }
        if (p + bsize > bz_buf_.size()) {
          if (bgzf_off_ < bgzf_size_) break;
          fail("truncated BGZF block");
          return len;
        }
        const unsigned char* t = bz_buf_.data() + p + bsize - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        if (isize > 65536) {
          fail("BGZF block larger than 64 KiB");
          return len;
        }
        if (len + total + isize > want) break;
        const size_t xlen = bz_buf_[p + 10] | ((size_t)bz_buf_[p + 11] << 8);
        blocks.push_back({p, bsize, xlen, len + total, isize});
        total += isize;
        p += bsize;
      }
      if (blocks.empty()) {
        if (p < bz_buf_.size() && bgzf_off_ >= bgzf_size_ && len + 65536 > want) return len;  // no room for the next block
        if (p < bz_buf_.size() && len + 65536 > want) return len;
        if (p >= bz_buf_.size() && bgzf_off_ >= bgzf_size_) {
          *at_end = true;
          return len;
        }
        if (bz_buf_.size() - bz_at_ >= (1u << 17)) return len;  // (cannot be: a window of 128 KiB holds a block)
        continue;
      }
      // (inflating is all this input costs - zlib gives a few hundred MB/s per core, the GPU takes tens of GB/s: every
      // core the host has, FQGPU_HOST_THREADS caps it)
      const unsigned T = (unsigned)std::min<size_t>(inflate_pool_->size(), std::max<size_t>(1, blocks.size() / 4));
      std::atomic<bool> bad{false};
      const unsigned char* src = bz_buf_.data();
      inflate_pool_->run(T, [&](unsigned t) {
        z_stream zs;  // one inflate state per thread and batch, reset per block (setting one up allocates its window)
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) {
          bad = true;
          return;
        }
        for (size_t i = blocks.size() * t / T; i < blocks.size() * (t + 1) / T && !bad; ++i) {
          const Block& b = blocks[i];
          if (b.isize == 0) continue;  // (the end-of-file marker, or an empty block)
          if (inflateReset(&zs) != Z_OK) {
            bad = true;
            break;
          }
          zs.next_in = const_cast<Bytef*>(src + b.at + 12 + b.xlen);
          zs.avail_in = (uInt)(b.size - 12 - b.xlen - 8);
          zs.next_out = reinterpret_cast<Bytef*>(dst + b.out_at);
          zs.avail_out = (uInt)b.isize;
          const int rc = inflate(&zs, Z_FINISH);
          const bool good = rc == Z_STREAM_END && zs.total_out == b.isize;
          const unsigned char* c = src + b.at + b.size - 8;
          const uint32_t want_crc = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
          if (!good || (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(dst + b.out_at), (uInt)b.isize) != want_crc)
            bad = true;
        }
        inflateEnd(&zs);
      });
      if (bad) {
        fail("corrupt BGZF block (inflate or CRC-32 failed)");
        return len;
      }
      len += total;
      bz_at_ = p;
      if (bz_at_ == bz_buf_.size() && bgzf_off_ >= bgzf_size_) {
        *at_end = true;
        return len;
      }
      if (len + 65536 > want) return len;
    }
  }

  void produce() {
    // pinning a slot takes as long as filling it: the slots behind the first are allocated by a helper while the first
    // is being read (the file may well end inside the first)
    const size_t head = std::min(kHead, std::max<size_t>(cap_, 4096));  // (tiny files: tiny slots)
    std::thread helper;
    const bool more = plain_fd_ >= 0 && plain_size_ > cap_;  // (gz input, stdin: unknown - the slots are pinned as they are needed)
    if (more)
      helper = std::thread([this, head] {
        for (int i = 1; i < kSlots; ++i) {
          char* b = slot_alloc(ctx_, head + slot_room() + 1);
          std::lock_guard<std::mutex> lk(mu_);
          slots_[i].buf = b;
          slots_[i].allocated = true;
          cv_.notify_all();
          if (!b || quit_) return;
        }
      });
    struct Join {
      std::thread& t;
      ~Join() {
        if (t.joinable()) t.join();
      }
    } join{helper};
    for (int i = 0;; i = (i + 1) % kSlots) {
      Slot& s = slots_[i];
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return (!s.ready && (i == 0 || !more || s.allocated)) || quit_; });
        if (quit_) return;
      }
      if (!s.buf) {
        if (i > 0 && more) {
          std::lock_guard<std::mutex> lk(mu_);
          fail_msg_ = "unable to allocate pinned memory";
          failed_ = true;
          cv_.notify_all();
          return;
        }
        s.buf = alloc(head + slot_room() + 1);
      }
      s.head = head;
      bool at_end = false;
      size_t len;
      if (!reframe_) len = read_some(s.buf + s.head, cap_, &at_end);
      else {
        // raw bytes = what the last round held back + a fresh read; the cut form of what can be judged goes out
        char* raw = s.buf + s.head;
        const size_t held = rf_tail_.size(), want = std::max(cap_, kReframeMin) - held;
        if (held) memcpy(raw, rf_tail_.data(), held);
        const size_t n = held + read_some(raw + held, want, &at_end);
        bool clean = true;
        size_t taken;
        if (!short_lines_only(raw, n, at_end, &taken)) taken = rf_.run(raw, n, at_end, rf_out_, &clean);
        rf_tail_.assign(raw + taken, n - taken);
        len = taken;
        if (!clean) {
          if (rf_out_.size() > slot_room()) {  // (cannot be: two bytes per limit - 1 >= 999 bytes were allowed for)
            std::lock_guard<std::mutex> lk(mu_);
            fail_msg_ = "internal: a re-framed piece outgrew its slot";
            failed_ = true;
            cv_.notify_all();
            return;
          }
          memcpy(raw, rf_out_.data(), rf_out_.size());
          len = rf_out_.size();
        }
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        s.len = len;
        s.last = at_end;
        s.ready = true;
        cv_.notify_all();
        if (at_end || failed_) return;
      }
    }
  }

  // the whole (rest of the) file as one image, read on the calling thread
  bool next_whole() {
    whole_mode_ = true;
    size_t cap = std::max<size_t>(cap_, 4096), len = whole_carry_;
    // a plain file's size is known: one allocation of exactly what is left (growing by doubling would hold the old and
    // the new pinned buffer at once and copy a 32 GB file seven times); gz input and stdin grow as they go
    if (plain_fd_ >= 0 && plain_size_ >= plain_off_) cap = std::max<size_t>((size_t)(plain_size_ - plain_off_) + len + 1, 4096);
    char* buf = alloc(cap + 1);
    if (len) memcpy(buf, data_ + carry_at_, len);
    whole_carry_ = 0;
    bool at_end = false;
    while (!at_end) {
      if (len == cap || (bgzf_fd_ >= 0 && cap - len < 65536)) {  // (read_bgzf fills whole blocks only)
        char* nb = alloc(cap * 2 + 1);
        memcpy(nb, buf, len);
        slot_release(ctx_, buf);
        buf = nb;
        cap *= 2;
      }
      if (plain_fd_ >= 0) len += read_plain(buf + len, cap - len, &at_end);
      else if (bgzf_fd_ >= 0) len += read_bgzf(buf + len, cap - len, &at_end);
      else if (pgz_) len += read_pgz(buf + len, cap - len, &at_end);
      else {
        const int got = gzread(gz_, buf + len, (unsigned)std::min<size_t>(cap - len, 1u << 30));
        if (got < 0) {
          int en = 0;
          FQ_PRINT_ERROR("%s.\n", gzerror(gz_, &en));
          leave(kExitSys);
        }
        if (got == 0) at_end = true;
        len += (size_t)got;
      }
      if (failed_) {
        FQ_PRINT_ERROR("%s.\n", fail_msg_.c_str());
        leave(kExitSys);
      }
    }
    if (reframe_) {
      bool clean = true;
      rf_.run(buf, len, true, rf_out_, &clean);  // (the whole file at once: this object hands out nothing else)
      if (!clean) {
        char* nb = alloc(rf_out_.size() + 1);
        memcpy(nb, rf_out_.data(), rf_out_.size());
        slot_release(ctx_, buf);
        buf = nb;
        len = rf_out_.size();
      }
    }
    slot_release(ctx_, whole_);
    whole_ = buf;
    data_ = buf;
    len_ = len;
    bytes_handed_out() += len;
    eof_ = true;
    finished_ = true;
    return true;
  }

  fqg_ctx* ctx_;
  std::string path_;
  gzFile gz_ = nullptr;
  int plain_fd_ = -1;
  uint64_t plain_size_ = 0, plain_off_ = 0;
  int bgzf_fd_ = -1;  // bgzip'd input: blocks inflated on many threads (read_bgzf)
  uint64_t bgzf_size_ = 0, bgzf_off_ = 0;
  int pgz_fd_ = -1;  // any other gzip file: chunks inflated on many threads (fq_pgzip.h)
  std::unique_ptr<ParallelGunzip> pgz_;
  struct Span {  // the compressed window (bytes of bz_raw_)
    const unsigned char* p = nullptr;
    size_t n = 0;
    const unsigned char* data() const { return p; }
    size_t size() const { return n; }
    unsigned char operator[](size_t i) const { return p[i]; }
  } bz_buf_;
  unsigned char* bz_raw_ = nullptr;
  size_t bz_raw_cap_ = 0, bz_at_ = 0;
  std::unique_ptr<ReaderPool> inflate_pool_;
  size_t cap_;
  Slot slots_[kSlots];
  std::unique_ptr<ReaderPool> pool_, scan_pool_;
  std::thread producer_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool quit_ = false, failed_ = false;
  std::string fail_msg_;
  // the piece the caller holds
  int cur_ = -1;
  const char* data_ = nullptr;
  char *big_ = nullptr, *whole_ = nullptr;
  size_t len_ = 0, carry_at_ = 0, whole_carry_ = 0;
  bool have_carry_ = false, whole_mode_ = false;
  bool eof_ = false, finished_ = false;
  // the reference's gzgets limits (fq_reframe.h); the state belongs to whichever thread reads (producer or next_whole)
  bool reframe_ = false;
  Reframer rf_;
  std::string rf_tail_, rf_out_;
};

}  // namespace fqhost
