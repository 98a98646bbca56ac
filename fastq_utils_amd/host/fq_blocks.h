// fq_blocks.h - one (gz) FASTQ input cut into BLOCKS OF EXACTLY B RECORDS (the last block: what is left), without
// looking at a GPU.  For programs that walk several inputs in lock step (fastq_pre_barcodes, reference
// src/fastq_pre_barcodes.c:594-727: iteration k uses record k of every input): when every input is cut at the same
// record numbers, block j of all inputs is a unit of work with no order among the units, and whichever device is free
// takes the next one ("shards naturally" by record block).
//
// A record is four lines, so the cut behind record R is the byte behind the 4R-th newline.  The reader threads count
// the newlines of what they read while the bytes are in their cache (as host/fq_multi.h does); the cut is then
// searched only inside the one part whose count crosses 4R.  Bytes read beyond a cut are carried into the next block.
#pragma once
#include <deque>

#include "fq_input.h"

namespace fqhost {

struct Block {
  char* data = nullptr;  // pinned
  size_t size = 0;
  uint64_t first_record = 0;  // == seq * B
  uint64_t lines = 0;         // newlines among the bytes (4 * B in every block but the last)
  bool final = false;
  int slot = -1;
  uint64_t seq = 0;
};

class RecordBlocks {
 public:
  static constexpr size_t kPeek = 1u << 20;

  RecordBlocks(fqg_ctx* ctx, const char* path, int n_slots) : ctx_(ctx), path_(path), slots_((size_t)n_slots) {
    if (path_ == "-") gz_ = gzdopen(fileno(stdin), "rb");
    else {
      const int fd = open(path, O_RDONLY);
      struct stat sb;
      if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
          if ((pgz_ = open_pgzip(fd, (uint64_t)sb.st_size, path))) pgz_fd_ = fd;  // inflated on many cores (fq_pgzip.h)
        } else {
          plain_fd_ = fd;
          plain_size_ = (uint64_t)sb.st_size;
        }
      }
      if (plain_fd_ < 0 && pgz_fd_ < 0) {
        if (fd >= 0) close(fd);
        gz_ = gzopen(path, "r");
      }
    }
    if (!gz_ && plain_fd_ < 0 && pgz_fd_ < 0) {
      FQ_PRINT_ERROR("Unable to open %s", path);
      leave(kExitParams);
    }
    if (gz_) gzbuffer(gz_, 1 << 20);
    // the first bytes, read here: the caller probes the first record in them and sizes the blocks from their lines
    carry_.resize(kPeek);
    uint64_t nl = 0;
    std::vector<Seg> segs;
    const size_t got = read_some(carry_.data(), kPeek, &at_end_, &nl, 0, &segs);
    if (failed_) {
      FQ_PRINT_ERROR("%s.\n", fail_msg_.c_str());
      leave(kExitSys);
    }
    carry_.resize(got);
    carry_lines_ = nl;
  }
  ~RecordBlocks() {
    abort();
    if (producer_.joinable()) producer_.join();
    if (pinner_.joinable()) pinner_.join();
    if (getenv("FQGPU_TIMING"))
      fprintf(diag(), "fqgpu timing: block cutter of %s: %llu blocks; waiting for a free (pinned) slot %.3f s, reading + counting lines %.3f s, cutting + carrying %.3f s\n",
              path_.c_str(), (unsigned long long)t_blocks_, t_wait_, t_read_, t_cut_);
    if (gz_) gzclose(gz_);
    if (plain_fd_ >= 0) close(plain_fd_);
    pgzip_report(pgz_.get(), path_);
    pgz_.reset();
    if (pgz_fd_ >= 0) close(pgz_fd_);
    for (auto& s : slots_) slot_release(ctx_, s.buf);
  }
  RecordBlocks(const RecordBlocks&) = delete;
  RecordBlocks& operator=(const RecordBlocks&) = delete;

  const char* peek() const { return carry_.data(); }
  size_t peek_size() const { return carry_.size(); }
  uint64_t peek_lines() const { return carry_lines_; }
  const std::string& path() const { return path_; }

  // blocks of `records` records from now on (call once, before the first next())
  void start(uint64_t records) {
    per_block_ = std::max<uint64_t>(records, 1);
    // (bytes per line so far; a file without a newline in its first bytes is one long line)
    bytes_per_line_ = carry_lines_ ? (double)carry_.size() / (double)carry_lines_ : (double)std::max<size_t>(carry_.size(), 64);
    pin_cap_ = block_bytes_estimate();
    // (no block is longer than its file: a small plain file gets small slots; a gzip file may inflate to 24 times its size
    // and more - a block that outgrows its slot grows it, reserve())
    if (plain_fd_ >= 0) pin_cap_ = (size_t)std::min<uint64_t>(pin_cap_, plain_size_ + (64u << 10));
    producer_ = std::thread([this] { produce(); });
    pinner_ = std::thread([this] { pin_slots(); });
  }
  // next block in file order; false when the input is used up.  Thread-safe.
  bool next(Block* out) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !ready_.empty() || done_ || failed_ || quit_; });
    if (quit_) return false;
    if (failed_) {
      FQ_PRINT_ERROR("%s.\n", fail_msg_.c_str());
      leave(kExitSys);
    }
    if (ready_.empty()) return false;
    *out = ready_.front();
    ready_.pop_front();
    bytes_handed_out() += out->size;
    return true;
  }
  void release(const Block& b) {
    std::lock_guard<std::mutex> lk(mu_);
    slots_[(size_t)b.slot].busy = false;
    cv_.notify_all();
  }
  // stop handing out blocks (error paths: consumers stop with blocks held)
  void abort() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
  }

 private:
  struct Slot {
    char* buf = nullptr;
    size_t cap = 0;
    bool busy = false;
  };
  struct Seg {  // a run of the block's bytes whose newline count is known
    size_t begin, end;
    uint64_t lines;
  };

  void fail(const char* msg) {
    std::lock_guard<std::mutex> lk(mu_);
    fail_msg_ = msg;
    failed_ = true;
    cv_.notify_all();
  }
  static uint64_t count_lines(const char* a, const char* b) {
    uint64_t c = 0;
    for (const char* p = a; (p = (const char*)memchr(p, '\n', (size_t)(b - p))) != nullptr; ++p) ++c;
    return c;
  }
  // up to `want` bytes to dst (they will sit at offset `at` of their block); their newline counts part by part
  size_t read_some(char* dst, size_t want, bool* at_end, uint64_t* newlines, size_t at, std::vector<Seg>* segs) {
    size_t len = 0;
    if (plain_fd_ >= 0) {
      const uint64_t left = plain_size_ - plain_off_;
      len = (size_t)std::min<uint64_t>(want, left);
      const unsigned T = (unsigned)std::min<uint64_t>(host_read_threads(), std::max<uint64_t>(1, len >> 22));
      std::vector<uint64_t> cnt(T, 0);
      std::atomic<bool> bad{false};
      auto bounds = [&](unsigned t, size_t* a, size_t* b) {
        *a = len * t / T;
        *b = t + 1 == T ? len : len * (t + 1) / T;
      };
      auto part = [&](unsigned t) {
        size_t a, b;
        bounds(t, &a, &b);
        // (256 KiB at a time: the lines are counted while the bytes are still in this core's cache)
        size_t done = a;
        uint64_t c = 0;
        while (done < b) {
          const ssize_t got = pread(plain_fd_, dst + done, std::min<size_t>(b - done, 256u << 10), (off_t)(plain_off_ + done));
          if (got <= 0) {
            bad = true;
            return;
          }
          c += count_lines(dst + done, dst + done + (size_t)got);
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
      for (unsigned t = 0; t < T; ++t) {
        size_t a, b;
        bounds(t, &a, &b);
        if (b > a) segs->push_back(Seg{at + a, at + b, cnt[t]});
        *newlines += cnt[t];
      }
      plain_off_ += len;
      if (plain_off_ >= plain_size_) *at_end = true;
      return len;
    }
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
    if (!*at_end && !pgz_) {
      const int c = gzgetc(gz_);
      if (c < 0) *at_end = true;
      else gzungetc(c, gz_);
    }
    const uint64_t c = count_lines(dst, dst + len);
    if (len) segs->push_back(Seg{at, at + len, c});
    *newlines += c;
    return len;
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
  size_t block_bytes_estimate() const { return (size_t)((double)(4 * per_block_) * bytes_per_line_ * 1.06) + (1u << 20); }
  // The slots are pinned by a thread of their own, at the size a block is expected to have, while the cutter fills - and
  // fills again - the ones it has been given (pinning 128 MiB takes six times as long as reading them from tmpfs; see
  // fq_multi.h).  A block that turns out larger grows its slot where it is (reserve).
  void pin_slots() {
    // (every slot the caller asked for, while the cutter still runs: consumers may each hold one while another waits for
    // the next block with a lock of the caller's held - fewer slots than consumers + 2 is a deadlock of their making)
    const size_t want = slots_.size();
    for (size_t i = 0; i < want; ++i) {
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (quit_ || failed_ || done_) break;
      }
      const size_t cap = pin_cap_;
      char* buf = slot_alloc(ctx_, cap + 1);
      std::lock_guard<std::mutex> lk(mu_);
      if (!buf) {
        if (i == 0) {
          fail_msg_ = "unable to allocate pinned memory";
          failed_ = true;
        }
        cv_.notify_all();
        break;
      }
      slots_[i].buf = buf;
      slots_[i].cap = cap;
      cv_.notify_all();
    }
  }
  bool reserve(Slot& s, size_t keep, size_t want) {
    if (want <= s.cap) return true;
    const size_t cap = std::max(want, s.cap + s.cap / 2);
    char* nb = slot_alloc(ctx_, cap + 1);
    if (!nb) {
      fail("unable to allocate pinned memory");
      return false;
    }
    if (keep) memcpy(nb, s.buf, keep);
    slot_release(ctx_, s.buf);
    s.buf = nb;
    s.cap = cap;
    return true;
  }

  void produce() {
    const uint64_t need = 4 * per_block_;
    uint64_t seq = 0;
    for (;;) {
      if (at_end_ && carry_.empty() && seq > 0) break;
      const double t0 = t_clock();
      const int si = free_slot();
      if (si < 0) return;
      const double t1 = t_clock();
      t_wait_ += t1 - t0;
      double t_reading = 0;
      Slot& s = slots_[(size_t)si];
      std::vector<Seg> segs;
      size_t len = carry_.size();
      uint64_t lines = carry_lines_;
      size_t room = (size_t)((double)need * bytes_per_line_ * 1.06) + (1u << 20);
      if (plain_fd_ >= 0) room = (size_t)std::min<uint64_t>(room, plain_size_ + (64u << 10));  // (as the pinner sizes them)
      if (!reserve(s, 0, std::max<size_t>(room, len))) return;
      if (len) {
        memcpy(s.buf, carry_.data(), len);
        segs.push_back(Seg{0, len, lines});
      }
      carry_.clear();
      carry_lines_ = 0;
      while (lines < need && !at_end_) {
        // what the missing lines should take, a little more than that: the surplus is carried, a shortfall reads again
        size_t est = (size_t)((double)(need - lines) * bytes_per_line_ * 1.03) + (64u << 10);
        if (plain_fd_ >= 0) est = (size_t)std::max<uint64_t>(1, std::min<uint64_t>(est, plain_size_ - plain_off_));  // (what the file still has)
        if (!reserve(s, len, len + est)) return;
        uint64_t nl = 0;
        const double tr = t_clock();
        const size_t got = read_some(s.buf + len, est, &at_end_, &nl, len, &segs);
        t_reading += t_clock() - tr;
        if (failed_) return;
        len += got;
        lines += nl;
        total_bytes_ += got;
        total_lines_ += nl;
        if (total_lines_ > 1000) bytes_per_line_ = (double)total_bytes_ / (double)total_lines_;
        {
          std::lock_guard<std::mutex> lk(mu_);
          if (quit_) return;
        }
      }
      size_t cut = len;
      if (lines >= need) {
        // the byte behind the need-th newline
        uint64_t acc = 0;
        for (const Seg& g : segs) {
          if (acc + g.lines < need) {
            acc += g.lines;
            continue;
          }
          const char* p = s.buf + g.begin;
          const char* e = s.buf + g.end;
          for (uint64_t k = acc; k < need; ++k) p = (const char*)memchr(p, '\n', (size_t)(e - p)) + 1;
          cut = (size_t)(p - s.buf);
          break;
        }
        if (cut < len) {
          carry_.assign(s.buf + cut, s.buf + len);
          carry_lines_ = lines - need;
        }
        lines = need;
      }
      Block b;
      b.data = s.buf;
      b.size = cut;
      b.first_record = seq * per_block_;
      b.lines = lines;
      b.final = at_end_ && carry_.empty();
      b.slot = si;
      b.seq = seq++;
      {
        std::lock_guard<std::mutex> lk(mu_);
        ready_.push_back(b);
        cv_.notify_all();
      }
      t_read_ += t_reading;
      t_cut_ += t_clock() - t1 - t_reading;
      ++t_blocks_;
      if (b.final) break;
    }
    std::lock_guard<std::mutex> lk(mu_);
    done_ = true;
    cv_.notify_all();
  }

  fqg_ctx* ctx_;
  std::string path_;
  gzFile gz_ = nullptr;
  int pgz_fd_ = -1;  // a gzip file inflated on many cores (fq_pgzip.h)
  std::unique_ptr<ParallelGunzip> pgz_;
  int plain_fd_ = -1;
  uint64_t plain_size_ = 0, plain_off_ = 0;
  std::vector<Slot> slots_;
  std::deque<Block> ready_;
  std::vector<char> carry_;  // read, not handed out yet
  uint64_t carry_lines_ = 0;
  bool at_end_ = false;
  uint64_t per_block_ = 1;
  double bytes_per_line_ = 64;
  uint64_t total_bytes_ = 0, total_lines_ = 0;
  std::thread producer_, pinner_;
  static double t_clock() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  double t_wait_ = 0, t_read_ = 0, t_cut_ = 0;  // FQGPU_TIMING (the producer's; read after its join)
  uint64_t t_blocks_ = 0;
  size_t pin_cap_ = 0;  // what the pinner gives every slot (from the first bytes' line lengths; set before it starts)
  std::mutex mu_;
  std::condition_variable cv_;
  std::unique_ptr<ReaderPool> pool_;
  bool quit_ = false, failed_ = false, done_ = false;
  std::string fail_msg_;
};

}  // namespace fqhost
