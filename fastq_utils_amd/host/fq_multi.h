// fq_multi.h - the -r pass of fastq_info (validate_single_fastq_file, reference src/fastq_info.c:155-176) over
// several GPUs of one node: FQGPU_DEVICES=0,1,...  ("validation shards naturally").
//
// One host thread + one fqg_ctx + one statistics accumulator per device.  The file is cut into pieces that START AND
// END AT RECORD BOUNDARIES without looking at a GPU: a record starts at every fourth line, so counting newlines on
// the host (the reader threads do it while the bytes are in their cache) gives every piece its first line number,
// and the bytes of the record that straddles a cut are moved to the piece it began in.  Pieces then have no order
// among them: whichever device is free takes the next one.  What the serial loop would report - the first finding
// in file order, the progress ticker, the statistics of a clean file - is put together on the host: findings by
// piece order, statistics by fqg_acc_export / fqg_acc_merge.  No collective is needed on this path.
#pragma once
#include <deque>
#include <functional>

#include "fq_input.h"

namespace fqhost {

struct Piece {
  char* data = nullptr;        // record-aligned image (points into a slot)
  size_t size = 0;
  uint64_t first_record = 0;   // records of the file before it
  uint64_t stream_offset = 0;  // bytes of the (inflated) file before it
  bool final = false;
  int slot = -1;
  uint64_t seq = 0;            // position in file order
};

class AlignedPieces {
 public:
  // limit: the file is taken to end after this many (inflated) bytes
  AlignedPieces(fqg_ctx* ctx, const char* path, size_t piece_bytes, int n_slots, uint64_t limit = ~0ull)
      : ctx_(ctx), cap_(piece_bytes), slots_((size_t)n_slots), limit_(limit) {
    path_ = path;
    if (path_ == "-") gz_ = gzdopen(fileno(stdin), "rb");
    else {
      const int fd = open(path, O_RDONLY);
      struct stat sb;
      if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
          cap_ = piece_for_file(cap_, (uint64_t)sb.st_size, true);  // (small files: small slots - pinning is start-up time)
          if ((pgz_ = open_pgzip(fd, (uint64_t)sb.st_size, path))) pgz_fd_ = fd;  // inflated on many cores (fq_pgzip.h)
        } else {
          plain_fd_ = fd;
          plain_size_ = std::min<uint64_t>((uint64_t)sb.st_size, limit_);
          cap_ = piece_for_file(cap_, plain_size_, false);
        }
      }
      if (plain_fd_ < 0 && pgz_fd_ < 0) {
        if (fd >= 0) close(fd);
        gz_ = gzopen(path, "r");
      }
    }
    if (!gz_ && plain_fd_ < 0 && pgz_fd_ < 0) {
      FQ_PRINT_ERROR("Unable to open %s", path);
      fqhost::leave(kExitParams);
    }
    if (gz_) gzbuffer(gz_, 1 << 20);
    want_slots_ = slots_.size();  // (all of them, while the cutter still runs: a file of one piece is done before the second is pinned)
    producer_ = std::thread([this] { produce(); });
    pinner_ = std::thread([this] { pin_slots(); });
  }
  ~AlignedPieces() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    if (producer_.joinable()) producer_.join();
    if (pinner_.joinable()) pinner_.join();
    if (getenv("FQGPU_TIMING"))
      fprintf(fqhost::diag(), "fqgpu timing: piece cutter: %llu slots filled; waiting for a free (pinned) slot %.3f s, pinning (beside it) %.3f s, reading + counting lines %.3f s, cutting %.3f s\n",
              (unsigned long long)t_slots_, t_wait_, t_pin_, t_read_, t_cut_);
    if (gz_) gzclose(gz_);
    if (plain_fd_ >= 0) close(plain_fd_);
    pgzip_report(pgz_.get(), path_);
    pgz_.reset();
    if (pgz_fd_ >= 0) close(pgz_fd_);
    for (auto& s : slots_) slot_release(ctx_, s.buf);
  }
  // next piece in file order; false when the file is exhausted.  Thread-safe.
  bool next(Piece* out) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !ready_.empty() || done_ || failed_ || quit_; });
    if (quit_) return false;
    if (failed_) {
      FQ_PRINT_ERROR("%s.\n", fail_msg_.c_str());
      fqhost::leave(kExitSys);
    }
    if (ready_.empty()) return false;
    *out = ready_.front();
    ready_.pop_front();
    bytes_handed_out() += out->size;
    return true;
  }
  void release(const Piece& p) {
    std::lock_guard<std::mutex> lk(mu_);
    slots_[(size_t)p.slot].busy = false;
    cv_.notify_all();
  }
  // Stop handing out pieces: wakes the producer (which may be waiting for a free slot that nobody will release any
  // more) and every consumer waiting in next(), which then returns false.  For the error paths of the consumers: they
  // stop with pieces still held, and joining them without this would wait forever.
  void abort() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
  }

 private:
  static constexpr size_t kTail = 8u << 20;  // room behind a slot's bytes for the rest of a straddling record
  struct Slot {
    char* buf = nullptr;
    bool busy = false;
  };

  size_t read_some(char* dst, size_t want, bool* at_end, uint64_t* newlines) {
    size_t len = 0;
    if (plain_fd_ >= 0) {
      const uint64_t left = plain_size_ - plain_off_;
      len = (size_t)std::min<uint64_t>(want, left);
      const unsigned T = (unsigned)std::min<uint64_t>(host_read_threads(), std::max<uint64_t>(1, len >> 22));
      std::vector<uint64_t> cnt(T, 0);
      std::atomic<bool> bad{false};
      auto part = [&](unsigned t) {
        const size_t a = (len * t / T) & ~(size_t)4095, b = t + 1 == T ? len : (len * (t + 1) / T) & ~(size_t)4095;
        // (256 KiB at a time: the lines are counted while the bytes are still in this core's cache - counting the part
        // after reading all of it is a second pass over memory)
        size_t done = a;
        uint64_t c = 0;
        while (done < b) {
          const ssize_t got = pread(plain_fd_, dst + done, std::min<size_t>(b - done, 256u << 10), (off_t)(plain_off_ + done));
          if (got <= 0) {
            bad = true;
            return;
          }
          const char* const end = dst + done + (size_t)got;
          for (const char* p = dst + done; (p = (const char*)memchr(p, '\n', (size_t)(end - p))) != nullptr; ++p) ++c;
          done += (size_t)got;
        }
        cnt[t] = c;
      };
      if (T <= 1) part(0);
      else {
        if (!pool_) pool_.reset(new ReaderPool(host_read_threads()));
        pool_->run(T, part);
      }
      if (bad) {
        fail("read error");
        return 0;
      }
      for (uint64_t c : cnt) *newlines += c;
      plain_off_ += len;
      if (plain_off_ >= plain_size_) *at_end = true;
      return len;
    }
    want = (size_t)std::min<uint64_t>(want, limit_ - gz_total_);
    if (pgz_) {
      len = pgz_->read(dst, want, at_end);
      if (pgz_->failed()) {
        fail(pgz_->error().c_str());
        return len;
      }
    }
    while (!pgz_ && len < want) {
      const int got = gzread(gz_, dst + len, (unsigned)std::min<size_t>(want - len, 1u << 30));
      if (got < 0) {
        int en = 0;
        fail(gzerror(gz_, &en));
        return len;
      }
      if (got == 0) {
        *at_end = true;
        break;
      }
      len += (size_t)got;
    }
    gz_total_ += len;
    if (gz_total_ >= limit_) *at_end = true;
    if (!*at_end && !pgz_) {
      const int c = gzgetc(gz_);
      if (c < 0) *at_end = true;
      else gzungetc(c, gz_);
    }
    for (const char* p = dst; (p = (const char*)memchr(p, '\n', (size_t)(dst + len - p))) != nullptr; ++p) ++*newlines;
    return len;
  }
  void fail(const char* msg) {
    std::lock_guard<std::mutex> lk(mu_);
    fail_msg_ = msg;
    failed_ = true;
    cv_.notify_all();
  }
  int free_slot() {
    std::unique_lock<std::mutex> lk(mu_);
    int s = -1;
    cv_.wait(lk, [&] {
      if (quit_ || failed_) return true;
      for (size_t i = 0; i < slots_.size(); ++i)
        if (slots_[i].buf && !slots_[i].busy) {
          s = (int)i;
          return true;
        }
      return false;
    });
    if (s >= 0) slots_[(size_t)s].busy = true;
    return s;
  }
  // Pinning a slot takes six times as long as filling it (128 MiB: 23 ms against 3.6 ms from tmpfs): the slots are pinned
  // by a thread of their own, one after the other, while the cutter fills - and fills again - the ones it has.  (The
  // cutter pinned them itself, on its way: 0.18 s of a 0.5 s job in front of every byte read after them.)
  void pin_slots() {
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < want_slots_; ++i) {
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (quit_ || failed_ || done_) break;
      }
      char* buf = slot_alloc(ctx_, cap_ + kTail + 1);
      std::lock_guard<std::mutex> lk(mu_);
      if (!buf) {
        if (i == 0) {  // (with fewer slots than asked for the loop still runs; with none it cannot)
          fail_msg_ = "unable to allocate pinned memory";
          failed_ = true;
        }
        cv_.notify_all();
        break;
      }
      slots_[i].buf = buf;
      cv_.notify_all();
    }
    t_pin_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  void produce() {
    // `held`: the piece whose end is not known yet (the record that straddles the next cut still has to be added)
    Piece held;
    bool have_held = false;
    size_t held_tail = 0;       // bytes of later slots already added behind it
    uint64_t lines_before = 0;  // newlines in the file before the raw bytes being read
    bool mid_line = false;      // the previous raw byte was not a newline
    bool at_end = false;
    uint64_t seq = 0;
    uint64_t raw_before = 0;    // bytes of the file before the raw bytes being read
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (!at_end) {
      const double t0 = now();
      const int si = free_slot();
      if (si < 0) return;
      const double t1 = now();
      Slot& s = slots_[(size_t)si];
      uint64_t nl = 0;
      const size_t len = read_some(s.buf, cap_, &at_end, &nl);
      if (failed_) return;
      const double t3 = now();
      t_wait_ += t1 - t0, t_read_ += t3 - t1, ++t_slots_;
      struct Cut {
        double& acc;
        double from;
        std::function<double()> clock;
        ~Cut() { acc += clock() - from; }
      } cut_timer{t_cut_, t3, now};
      // where the first record of these bytes starts: at the first line whose number is a multiple of four
      uint64_t skip_lines = (4 - lines_before % 4) % 4;
      if (mid_line && skip_lines == 0) skip_lines = 4;
      if (!have_held) skip_lines = 0;  // the file starts with a record
      size_t skip = 0;
      bool found = true;
      for (uint64_t k = 0; k < skip_lines; ++k) {
        const char* p = (const char*)memchr(s.buf + skip, '\n', len - skip);
        if (!p) {
          found = false;
          break;
        }
        skip = (size_t)(p - s.buf) + 1;
      }
      if (!found) skip = len;  // no record starts in these bytes: all of them belong to the held piece
      if (have_held) {
        if (held_tail + skip > kTail) {
          fail("more than 8 MiB of one record straddle two pieces (FQGPU_DEVICES): use one device");
          return;
        }
        memcpy(held.data + held.size, s.buf, skip);
        held.size += skip;
        held_tail += skip;
      }
      if (found && (skip < len || at_end)) {
        if (have_held) publish(held);
        held = Piece();
        held.data = s.buf + skip;
        held.size = len - skip;
        held.first_record = (lines_before + skip_lines) / 4;
        held.stream_offset = raw_before + skip;
        held.slot = si;
        held.seq = seq++;
        held_tail = 0;
        have_held = true;
      } else {
        // nothing of this slot starts a piece: give it back (its bytes were appended to the held piece)
        std::lock_guard<std::mutex> lk(mu_);
        s.busy = false;
      }
      lines_before += nl;
      raw_before += len;
      if (len) mid_line = s.buf[len - 1] != '\n';
    }
    if (have_held) {
      held.final = true;
      publish(held);
    } else {
      // an empty file: one empty, final piece
      const int si = free_slot();
      if (si < 0) return;
      Slot& s = slots_[(size_t)si];
      Piece p;
      p.data = s.buf;
      p.final = true;
      p.slot = si;
      publish(p);
    }
    std::lock_guard<std::mutex> lk(mu_);
    done_ = true;
    cv_.notify_all();
  }
  void publish(const Piece& p) {
    std::lock_guard<std::mutex> lk(mu_);
    ready_.push_back(p);
    cv_.notify_all();
  }

  fqg_ctx* ctx_;
  std::string path_;
  gzFile gz_ = nullptr;
  int pgz_fd_ = -1;  // a gzip file inflated on many cores (fq_pgzip.h)
  std::unique_ptr<ParallelGunzip> pgz_;
  int plain_fd_ = -1;
  uint64_t plain_size_ = 0, plain_off_ = 0;
  size_t cap_;
  std::vector<Slot> slots_;
  uint64_t limit_ = ~0ull, gz_total_ = 0;
  std::deque<Piece> ready_;
  std::thread producer_, pinner_;
  size_t want_slots_ = 0;
  std::mutex mu_;
  std::condition_variable cv_;
  std::unique_ptr<ReaderPool> pool_;
  bool quit_ = false, failed_ = false, done_ = false;
  std::string fail_msg_;
  double t_wait_ = 0, t_pin_ = 0, t_read_ = 0, t_cut_ = 0;  // FQGPU_TIMING (written by the producer, read after its join)
  uint64_t t_slots_ = 0;
};

inline std::vector<int> devices_from_env() {
  std::vector<int> d;
  const char* e = getenv("FQGPU_DEVICES");
  if (!e) return d;
  for (const char* p = e; *p;) {
    char* end = nullptr;
    const long v = strtol(p, &end, 10);
    if (end == p) break;
    d.push_back((int)v);
    p = *end == ',' ? end + 1 : end;
  }
  return d;
}

}  // namespace fqhost
