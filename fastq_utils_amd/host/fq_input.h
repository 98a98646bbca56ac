// fq_input.h - host side of the drop-in programs: reading (optionally gzipped) FASTQ files into
// pinned staging buffers, piece by piece, with the incomplete tail of one piece carried into the
// next.  Decompression stays on the host (zlib), as in the reference (src/fastq.c:631-661).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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

class Input {
 public:
  Input(fqg_ctx* ctx, const char* path, size_t piece_bytes) : ctx_(ctx), path_(path), cap_(piece_bytes) {
    // fastq_open, src/fastq.c:631-661
    if (path_ == "-") gz_ = gzdopen(fileno(stdin), "rb");
    else gz_ = gzopen(path, "r");
    if (!gz_) {
      FQ_PRINT_ERROR("Unable to open %s", path);
      exit(kExitParams);
    }
    gzbuffer(gz_, 1 << 20);
    buf_ = alloc(cap_);
  }
  ~Input() {
    if (gz_) gzclose(gz_);
    if (buf_) fqg_host_free(ctx_, buf_);
  }
  Input(const Input&) = delete;
  Input& operator=(const Input&) = delete;

  // Next piece: the carried tail of the previous one followed by fresh bytes.  Returns false once
  // the final piece has been handed out.  An empty file yields one empty, final piece.
  bool next(bool whole_file = false) {
    if (finished_) return false;
    len_ = carry_;
    carry_ = 0;
    while (!eof_) {
      if (len_ == cap_) {
        if (!whole_file) break;
        grow(cap_ * 2);
      }
      const size_t want = cap_ - len_;
      const int got = gzread(gz_, buf_ + len_, (unsigned)(want > (1u << 30) ? (1u << 30) : want));
      if (got < 0) {
        int en = 0;
        FQ_PRINT_ERROR("%s.\n", gzerror(gz_, &en));
        exit(kExitSys);
      }
      if (got == 0) eof_ = true;
      len_ += (size_t)got;
    }
    if (!eof_) {  // a file that ends exactly where the buffer does
      const int c = gzgetc(gz_);
      if (c < 0) eof_ = true;
      else gzungetc(c, gz_);
    }
    if (eof_) finished_ = true;
    return true;
  }
  // keep bytes [consumed, size) for the next piece (only meaningful for non-final pieces)
  void carry_from(size_t consumed) {
    carry_ = len_ - consumed;
    if (carry_ == cap_) grow(cap_ * 2);  // a single record larger than a whole piece
    if (carry_) memmove(buf_, buf_ + consumed, carry_);
  }
  void stop() { finished_ = true; }
  const char* data() const { return buf_; }
  size_t size() const { return len_; }
  bool final() const { return eof_; }
  const std::string& path() const { return path_; }

 private:
  char* alloc(size_t n) {
    char* p = static_cast<char*>(fqg_host_alloc(ctx_, n));
    if (!p) {
      FQ_PRINT_ERROR("unable to allocate %zu bytes of pinned memory", n);
      exit(kExitSys);
    }
    return p;
  }
  void grow(size_t ncap) {
    char* nb = alloc(ncap);
    memcpy(nb, buf_, len_);
    fqg_host_free(ctx_, buf_);
    buf_ = nb;
    cap_ = ncap;
  }
  fqg_ctx* ctx_;
  std::string path_;
  gzFile gz_ = nullptr;
  char* buf_ = nullptr;
  size_t cap_, len_ = 0, carry_ = 0;
  bool eof_ = false, finished_ = false;
};

}  // namespace fqhost
