// fq_input.h - host side of the drop-in programs: reading (optionally gzipped) FASTQ files into
// pinned staging buffers, piece by piece, with the incomplete tail of one piece carried into the
// next.  Decompression stays on the host (zlib), as in the reference (src/fastq.c:631-661).
//
// Reading runs AHEAD of the GPU: a producer thread fills a ring of pinned slots while the caller has
// the previous piece copied to the device and validated (what the reference does serially with four
// gzgets per record, src/fastq.c:245-261).  A plain (not gzipped) regular file is read with pread() by
// several threads at once - 50 Mreads/s of 150 bp reads are 17.5 GB/s, more than one core copies; gzip
// input and stdin are inflated by one thread (zlib), still ahead of the GPU.
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
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fqg.h"

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
inline bool& keep_slots_until_exit() {
  static bool v = false;
  return v;
}

// How the programs leave: with everything they said flushed, and WITHOUT exit()'s hooks.  The HIP runtime tears itself
// down in one of them, and it must not meet a thread of ours that is still inside a HIP call (a reader pinning its next
// slot while the main thread has found the file's first error): that is a crash after the error message, i.e. a wrong
// exit status.  Nothing is lost: outputs are closed by those who write them before they leave.
[[noreturn]] inline void leave(int code) {
  fflush(stdout);
  fflush(stderr);
  _exit(code);
}

inline unsigned host_read_threads() {
  if (const char* e = getenv("FQGPU_HOST_THREADS")) return (unsigned)std::max(1L, strtol(e, nullptr, 10));
  const unsigned hw = std::thread::hardware_concurrency();
  // (a dozen copy a tmpfs file faster than PCIe takes it; more of them only compete with the DMA for host memory:
  // 8 / 16 / 32 / 64 threads -> 1.06 / 1.16 / 1.24 / 1.46 s for the 100 M-read file of the bench)
  return std::max(1u, std::min(12u, hw ? hw : 1u));
}

// A few reader threads that stay around: every slot of a plain file is read by all of them at once, and starting
// thirty threads per 256 MiB slot cost a quarter of the time the slot's copy to the GPU takes.
class ReaderPool {
 public:
  explicit ReaderPool(unsigned n) : n_(std::max(1u, n)) {
    for (unsigned t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
  }
  ~ReaderPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  unsigned size() const { return n_; }
  // fn(t) for t in [0, parts), parts <= size(); the caller runs part 0 and returns when all are done
  template <class F>
  void run(unsigned parts, F&& fn) {
    if (parts <= 1) {
      fn(0u);
      return;
    }
    std::function<void(unsigned)> f = fn;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &f;
      parts_ = parts;
      left_ = parts - 1;
      ++gen_;
    }
    cv_.notify_all();
    fn(0u);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return left_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop(unsigned t) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(unsigned)>* f = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (quit_) return;
        if (t < parts_) f = fn_;
      }
      if (!f) continue;
      (*f)(t);
      std::lock_guard<std::mutex> lk(mu_);
      if (--left_ == 0) done_.notify_all();
    }
  }
  unsigned n_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned)>* fn_ = nullptr;
  unsigned parts_ = 0, left_ = 0;
  unsigned long gen_ = 0;
  bool quit_ = false;
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
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
          plain_fd_ = fd;
          plain_size_ = (uint64_t)sb.st_size;
        }
      }
      if (plain_fd_ < 0) {
        if (fd >= 0) close(fd);
        gz_ = gzopen(path, "r");
      }
    }
    if (!gz_ && plain_fd_ < 0) {
      FQ_PRINT_ERROR("Unable to open %s", path);
      leave(kExitParams);
    }
    if (gz_) gzbuffer(gz_, 1 << 20);
    if (plain_fd_ >= 0 && plain_size_ < cap_) cap_ = std::max<size_t>(plain_size_, 1);  // small file: one small slot
  }
  ~Input() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    if (producer_.joinable()) producer_.join();
    if (gz_) gzclose(gz_);
    if (plain_fd_ >= 0) close(plain_fd_);
    if (keep_slots_until_exit()) return;
    for (Slot& s : slots_)
      if (s.buf) fqg_host_free(ctx_, s.buf);
    if (whole_) fqg_host_free(ctx_, whole_);
    if (big_) fqg_host_free(ctx_, big_);
  }
  Input(const Input&) = delete;
  Input& operator=(const Input&) = delete;

  // Next piece: the carried tail of the previous one followed by fresh bytes.  Returns false once
  // the final piece has been handed out.  An empty file yields one empty, final piece.
  bool next(bool whole_file = false) {
    if (finished_) return false;
    if (whole_file) return next_whole();
    if (!producer_.joinable()) producer_ = std::thread([this] { produce(); });
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
      if (big_) fqg_host_free(ctx_, big_);
      big_ = nb;
      data_ = nb;
    } else {
      if (carry) memcpy(s.buf + s.head - carry, carry_src, carry);
      data_ = s.buf + s.head - carry;
    }
    len_ = carry + s.len;
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

  char* alloc(size_t n) {
    char* p = static_cast<char*>(fqg_host_alloc(ctx_, n));
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

  void produce() {
    // pinning a slot takes as long as filling it: the slots behind the first are allocated by a helper while the first
    // is being read (the file may well end inside the first)
    const size_t head = std::min(kHead, std::max<size_t>(cap_, 4096));  // (tiny files: tiny slots)
    std::thread helper;
    const bool more = plain_fd_ >= 0 && plain_size_ > cap_;  // (gz input, stdin: unknown - the slots are pinned as they are needed)
    if (more)
      helper = std::thread([this, head] {
        for (int i = 1; i < kSlots; ++i) {
          char* b = static_cast<char*>(fqg_host_alloc(ctx_, head + cap_ + 1));
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
        s.buf = alloc(head + cap_ + 1);
      }
      s.head = head;
      bool at_end = false;
      const size_t len = plain_fd_ >= 0 ? read_plain(s.buf + s.head, cap_, &at_end) : read_gz(s.buf + s.head, cap_, &at_end);
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
      if (len == cap) {
        char* nb = alloc(cap * 2 + 1);
        memcpy(nb, buf, len);
        fqg_host_free(ctx_, buf);
        buf = nb;
        cap *= 2;
      }
      if (plain_fd_ >= 0) len += read_plain(buf + len, cap - len, &at_end);
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
    if (whole_) fqg_host_free(ctx_, whole_);
    whole_ = buf;
    data_ = buf;
    len_ = len;
    eof_ = true;
    finished_ = true;
    return true;
  }

  fqg_ctx* ctx_;
  std::string path_;
  gzFile gz_ = nullptr;
  int plain_fd_ = -1;
  uint64_t plain_size_ = 0, plain_off_ = 0;
  size_t cap_;
  Slot slots_[kSlots];
  std::unique_ptr<ReaderPool> pool_;
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
};

}  // namespace fqhost
