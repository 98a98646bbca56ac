// fq_filter_run.h - the record loop shared by the drop-in fastq_filter_n and fastq_trim_poly_at:
// (gz) input piece by piece -> frame on the GPU -> fqg_records_filter -> text back to the host ->
// the caller's sink.  Reading follows fastq_read_entry (reference src/fastq.c:245-261): a record
// that stops after one to three lines is a truncated file (exit 1), reported with the file's line
// counter, which starts at 0 and has advanced by 4 per complete record.
#pragma once
#include <functional>
#include <vector>

#include "fq_input.h"

namespace fqhost {

inline size_t piece_bytes_env() {
  const char* e = getenv("FQGPU_CHUNK_MB");
  size_t mb = e ? strtoull(e, nullptr, 10) : 512;
  if (mb < 1) mb = 1;
  return mb << 20;
}

struct FilterTotals {
  unsigned long processed = 0, trimmed = 0, discarded = 0;
};

// sink(text, bytes): the kept records of one piece; progress(before, after): records read so far
inline FilterTotals run_filter(fqg_ctx* ctx, const char* path, const fqg_filter_params& fp,
                               const std::function<void(const char*, size_t)>& sink,
                               const std::function<void(unsigned long, unsigned long)>& progress) {
  auto lib = [&](int rc, const char* what) {
    if (rc != 0) {
      FQ_PRINT_ERROR("GPU library failure in %s (%d): %s", what, rc, fqg_last_error(ctx));
      fqhost::leave(kExitSys);
    }
  };
  Input in(ctx, path, piece_bytes_env());
  fqg_file_state st;
  memset(&st, 0, sizeof(st));
  FilterTotals t;
  std::vector<char> host;
  int tail_lines = 0;
  while (in.next()) {
    fqg_validate_result r;
    lib(fqg_validate(ctx, nullptr, in.data(), in.size(), FQG_MEM_HOST, in.final() ? 1 : 0, &st,
                     FQG_VALIDATE_FRAME_ONLY | in.vflags(), &r),
        "fqg_validate");
    if (r.code == FQG_E_LINE_TOO_LONG) {
      // The reference reads such a line in pieces (src/fastq.c:249-253) and copies the pieces.  Everything in front of
      // this piece of input went the reference's way; the program runs itself again, as a child, on input that is cut
      // where gzgets cuts it (fq_respawn.h, fq_reframe.h: inflated input and stdin are cut while they are read and never
      // come here), where every piece is a line - a C string to the kernels, as it is to the reference.
      if (reframe_supported() && !reframing() && strcmp(path, "-") != 0) respawn_reframed();
      FQ_PRINT_ERROR("Error in file %s: record %lu has a line longer than the reference's line buffers (%d / %d bytes)", path,
                     t.processed + (unsigned long)r.record + 1, FQG_MAX_LABEL_LENGTH - 1, FQG_MAX_READ_LENGTH - 1);
      fflush(stdout);
      fqhost::leave(kExitSys);
    }
    tail_lines = r.tail_lines;
    if (r.n_records) {
      fqg_frame* frame = nullptr;
      lib(fqg_frame_retain(ctx, &frame), "fqg_frame_retain");
      fqg_filter_result fr;
      lib(fqg_records_filter(ctx, frame, 0, r.n_records, &fp, &fr), "fqg_records_filter");
      if (fr.out_bytes) {
        if (host.size() < fr.out_bytes) host.resize(fr.out_bytes);
        lib(fqg_records_filter_output(ctx, host.data(), fr.out_bytes), "fqg_records_filter_output");
        sink(host.data(), fr.out_bytes);
      }
      fqg_frame_release(frame);
      const unsigned long before = t.processed;
      t.processed += r.n_records;
      t.trimmed += fr.n_trimmed;
      t.discarded += fr.n_discarded;
      progress(before, t.processed);
    }
    // a record whose first line starts with a NUL byte: "no entry", the loop ends (src/fastq.c:250); one with another
    // line that starts with NUL: an empty string - the file is truncated there (tail_lines says so)
    if (r.stopped || r.code == FQG_E_TRUNCATED) break;
    if (!in.final()) in.carry_from(r.consumed);
  }
  if (tail_lines > 0) {
    FQ_PRINT_ERROR("Error in file %s: line %lu: file truncated", path, 4ul * t.processed);
    fflush(stdout);
    fqhost::leave(1);
  }
  return t;
}

}  // namespace fqhost
