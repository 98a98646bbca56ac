// fq_parallel.h - host-side (de)compression on all cores, the boundary work of SURVEY 8f-2.
// The reference compresses its FASTQ output and inflates its BAM input on one thread, which is 7x the
// cost of the transform itself (gzip level 4) and the whole cost of reading a BAM file.  Here:
//   GzipMembers  the text is cut into blocks, every block becomes a complete gzip member on its own
//                thread, members are written in order.  A multi-member file inflates to exactly the
//                bytes a single-member file would (RFC 1952 2.2; zlib's gzread and gzip -d read through
//                members), so what a consumer sees is unchanged.
//   bgzf_inflate_parallel  BGZF blocks are independent deflate streams with their inflated size in the
//                trailer: sizes first, then every block inflates to its own place.
// FQGPU_HOST_THREADS sets the number of threads (default: the hardware's, at most 256 - deflate at the reference's
// level runs at tens of MB/s per core, so 87 GB of re-tagged FASTQ are minutes on 32 cores).
#pragma once
#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "fq_fastdeflate.h"

namespace fqhost {

// the cores this process may run on (a container's share of the machine: more threads than that only take turns)
inline unsigned usable_cores() {
  unsigned n = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
  if (!n) n = std::thread::hardware_concurrency();
  if (!n) n = 1;
  // a container's CPU quota (cgroup v2 cpu.max "quota period", v1 cpu.cfs_quota_us / cpu.cfs_period_us): 256 runnable
  // threads on a quota of a few cores are throttled together
  double quota = 0, period = 0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
    fclose(f);
  } else {
    FILE* a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    FILE* b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (a && b && fscanf(a, "%lf", &quota) == 1 && fscanf(b, "%lf", &period) == 1) {
    } else quota = 0;
    if (a) fclose(a);
    if (b) fclose(b);
  }
  if (quota > 0 && period > 0) {
    const unsigned q = (unsigned)(quota / period + 0.5);
    if (q >= 1 && q < n) n = q;
  }
  return n;
}

inline unsigned host_threads() {
  const char* e = getenv("FQGPU_HOST_THREADS");
  long v = e ? atol(e) : 0;
  if (v < 1) v = (long)usable_cores();
  return (unsigned)std::min<long>(std::max<long>(v, 1), 256);
}

// fn(i) for i in [0, n) on up to host_threads() threads (dynamic: an atomic counter hands out items)
template <class F>
inline void parallel_items(size_t n, F fn) {
  const unsigned nt = (unsigned)std::min<size_t>(host_threads(), n);
  if (nt <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&] {
      for (size_t i; (i = next.fetch_add(1)) < n;) fn(i);
    });
  for (auto& t : th) t.join();
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

// gzip output as a sequence of members
class GzipMembers {
 public:
  // path "-": stdout (as the reference's gzdopen(fileno(stdout), "wb")).  false: cannot open.
  bool open(const char* path, int level) {
    level_ = level;
    // FQGPU_GZIP_LEVEL=1..9: another deflate level than the reference's (what a reader inflates is the same; level 1 is
    // about three times as fast as the reference's 4 and a fifth larger)
    if (const char* e = getenv("FQGPU_GZIP_LEVEL")) {
      const int v = atoi(e);
      if (v >= 1 && v <= 9) level_ = v;
    }
    // FQGPU_GZIP_FAST=1: the members from fq_fastdeflate.h instead of zlib's - three times as fast as the reference's
    // level 4 and a tenth larger (zlib level 1's size class)
    if (const char* e = getenv("FQGPU_GZIP_FAST")) fast_ = atoi(e) != 0;
    if (path[0] == '-' && path[1] == 0) {
      f_ = stdout;
      name_ = "<fd:1>";  // (what zlib's gzdopen calls it in its messages)
    } else {
      f_ = fopen(path, "wb");
      own_ = true;
      name_ = path;
    }
    return f_ != nullptr;
  }
  // what gzerror() would say about the write that failed (zlib's gz_error: "<path>: <text>", the text being
  // strerror(errno) for a failed write(2) and "out of memory" for a failed deflate)
  const std::string& error() const { return err_; }
  // Every member holds exactly 1 MiB of text (the last one the rest), whatever the sizes of the calls: the bytes
  // written are a function of the text alone.  A run that starts over on re-framed input (fq_respawn.h) skips as many
  // bytes of its stdout as the first run wrote - they must be the same bytes, though the two runs cut their input into
  // different pieces.  Text short of a member waits in pend_ (the first run never writes it).
  bool write(const char* text, size_t n) {
    const size_t block = 1u << 20;  // (a member per MiB: a piece of 128 MiB is work for every core)
    if (!pend_.empty()) {
      const size_t take = std::min(block - pend_.size(), n);
      pend_.append(text, take);
      text += take;
      n -= take;
      if (pend_.size() < block) return true;
      std::vector<uint8_t> m;
      if (!member(pend_.data(), pend_.size(), m)) {
        err_ = name_ + ": out of memory";
        return false;
      }
      pend_.clear();
      if (!put(m)) return false;
    }
    const size_t nb = n / block;
    if (nb) {
      std::vector<std::vector<uint8_t>> out(nb);
      std::atomic<bool> ok{true};
      parallel_items(nb, [&](size_t i) {
        if (!member(text + i * block, block, out[i])) ok = false;
      });
      if (!ok) {
        err_ = name_ + ": out of memory";
        return false;
      }
      for (auto& m : out)
        if (!put(m)) return false;
    }
    pend_.assign(text + nb * block, n - nb * block);
    return true;
  }
  bool close() {
    bool ok = true;
    if (!f_) return true;
    if (!pend_.empty() || !wrote_) {  // the rest - and an empty gzip stream is still a gzip stream
      std::vector<uint8_t> m;
      ok = member(pend_.data(), pend_.size(), m) && put(m);
      pend_.clear();
    }
    if (own_) ok = fclose(f_) == 0 && ok;
    else ok = fflush(f_) == 0 && ok;
    if (!ok && err_.empty()) err_ = name_ + ": " + strerror(errno);
    f_ = nullptr;
    return ok;
  }

 private:
  bool put(const std::vector<uint8_t>& m) {
    if (fwrite(m.data(), 1, m.size(), f_) != m.size()) {
      err_ = name_ + ": " + strerror(errno);
      return false;
    }
    wrote_ = true;
    return true;
  }
  bool member(const char* p, size_t n, std::vector<uint8_t>& out) const {
    if (fast_) return fdef::gzip_member_fast(p, n, out);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level_, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&zs, (uLong)n) + 32);
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(p));
    zs.avail_in = (uInt)n;
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return rc == Z_STREAM_END;
  }
  FILE* f_ = nullptr;
  std::string name_, err_, pend_;
  bool own_ = false, wrote_ = false, fast_ = false;
  int level_ = 4;
};

// the whole file, BGZF members inflated back to back (SAM/BAM specification, section 4.1)
inline bool bgzf_inflate_parallel(const std::vector<uint8_t>& raw, std::vector<uint8_t>& out) {
  struct Block {
    size_t at, size, xlen, out_at, isize;
  };
  std::vector<Block> blocks;
  size_t p = 0, total = 0;
  while (p + 18 <= raw.size()) {
    if (raw[p] != 0x1f || raw[p + 1] != 0x8b || raw[p + 2] != 8 || !(raw[p + 3] & 4)) return false;
    const size_t xlen = raw[p + 10] | (raw[p + 11] << 8);
    size_t q = p + 12, bsize = 0;
    bool found = false;
    while (q + 4 <= p + 12 + xlen && q + 4 <= raw.size()) {
      const size_t slen = raw[q + 2] | (raw[q + 3] << 8);
      if (raw[q] == 66 && raw[q + 1] == 67 && slen == 2 && q + 6 <= raw.size()) {
        bsize = (raw[q + 4] | (raw[q + 5] << 8)) + 1;
        found = true;
      }
      q += 4 + slen;
    }
    if (!found || p + bsize > raw.size() || bsize < 12 + xlen + 8) return false;
    const uint8_t* isz = &raw[p + bsize - 4];
    const size_t isize = (size_t)isz[0] | ((size_t)isz[1] << 8) | ((size_t)isz[2] << 16) | ((size_t)isz[3] << 24);
    blocks.push_back({p, bsize, xlen, total, isize});
    total += isize;
    p += bsize;
  }
  out.resize(total);
  std::atomic<bool> ok{true};
  // blocks are at most 64 KiB: hand them out in runs
  const size_t run = 64, n_runs = (blocks.size() + run - 1) / run;
  parallel_items(n_runs, [&](size_t r) {
    for (size_t i = r * run; i < std::min(blocks.size(), (r + 1) * run); ++i) {
      const Block& b = blocks[i];
      if (b.isize == 0) continue;  // (the end-of-file block; zlib refuses a null output pointer when nothing else was inflated)
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) {
        ok = false;
        return;
      }
      zs.next_in = const_cast<Bytef*>(&raw[b.at + 12 + b.xlen]);
      zs.avail_in = (uInt)(b.size - 12 - b.xlen - 8);
      zs.next_out = out.data() + b.out_at;
      zs.avail_out = (uInt)b.isize;
      const int rc = inflate(&zs, Z_FINISH);
      const bool good = rc == Z_STREAM_END && zs.total_out == b.isize;
      inflateEnd(&zs);
      if (!good) ok = false;
    }
  });
  return ok;
}

// BGZF output (SAM/BAM specification 4.1, what libbam's bgzf_write / bgzf_close produce): blocks of at most 0xff00
// input bytes, each a gzip member with the BC extra field, every block deflated on its own thread; the empty block
// that marks the end of the file last.  What a reader inflates is `data`; the block boundaries are not part of it.
inline bool bgzf_deflate_parallel(const std::vector<std::pair<const uint8_t*, size_t>>& pieces, int level, std::vector<uint8_t>& out) {
  constexpr size_t kBlock = 0xff00;
  size_t n = 0;
  for (auto& pc : pieces) n += pc.second;
  const size_t nb = (n + kBlock - 1) / kBlock;
  std::vector<std::vector<uint8_t>> blocks(nb);
  std::atomic<bool> ok{true};
  // byte k of the concatenation of the pieces
  auto gather = [&](size_t from, size_t len, uint8_t* dst) {
    size_t skip = from;
    for (auto& pc : pieces) {
      if (skip >= pc.second) {
        skip -= pc.second;
        continue;
      }
      const size_t take = std::min(len, pc.second - skip);
      memcpy(dst, pc.first + skip, take);
      dst += take;
      len -= take;
      skip = 0;
      if (!len) break;
    }
  };
  const size_t run = 16, n_runs = (nb + run - 1) / run;
  parallel_items(n_runs, [&](size_t r) {
    std::vector<uint8_t> in(kBlock);
    for (size_t b = r * run; b < std::min(nb, (r + 1) * run); ++b) {
      const size_t len = std::min(kBlock, n - b * kBlock);
      gather(b * kBlock, len, in.data());
      std::vector<uint8_t>& o = blocks[b];
      o.resize(18 + compressBound((uLong)len) + 8);
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
        ok = false;
        return;
      }
      zs.next_in = in.data();
      zs.avail_in = (uInt)len;
      zs.next_out = o.data() + 18;
      zs.avail_out = (uInt)(o.size() - 26);
      const int rc = deflate(&zs, Z_FINISH);
      const size_t clen = zs.total_out;
      deflateEnd(&zs);
      const size_t bsize = 18 + clen + 8;
      if (rc != Z_STREAM_END || bsize > 65536) {
        ok = false;
        return;
      }
      static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
      memcpy(o.data(), head, 16);
      o[16] = (uint8_t)((bsize - 1) & 0xff);
      o[17] = (uint8_t)((bsize - 1) >> 8);
      const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in.data(), (uInt)len);
      uint8_t* t = o.data() + 18 + clen;
      for (int k = 0; k < 4; ++k) t[k] = (uint8_t)(crc >> (8 * k));
      for (int k = 0; k < 4; ++k) t[4 + k] = (uint8_t)((uint32_t)len >> (8 * k));
      o.resize(bsize);
    }
  });
  if (!ok) return false;
  size_t total = 28;
  for (auto& b : blocks) total += b.size();
  out.clear();
  out.reserve(total);
  for (auto& b : blocks) out.insert(out.end(), b.begin(), b.end());
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  out.insert(out.end(), eof, eof + 28);
  return true;
}

}  // namespace fqhost
