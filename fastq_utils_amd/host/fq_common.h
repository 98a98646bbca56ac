// fq_common.h - what the drop-in programs that read FASTQ through the name index share: the library handle, the
// wording of the reference's messages (src/fastq.c), the once-per-file probes and the indexing loop
// (fastq_index_readnames, src/fastq.c:396-439) as bulk calls.  Included by fastq_info.cpp and fastq_filterpair.cpp.
#pragma once
#include <regex.h>

#include <algorithm>
#include <string>
#include <vector>

#include "fq_input.h"

using namespace fqhost;

namespace {

fqg_ctx* g_ctx = nullptr;

[[noreturn]] void die_lib(const char* what, int rc) {
  FQ_PRINT_ERROR("GPU library failure in %s (%d): %s", what, rc, g_ctx ? fqg_last_error(g_ctx) : "no context");
  fqhost::leave(kExitSys);
}
#define LIB(call)                   \
  do {                              \
    int rc__ = (call);              \
    if (rc__ != 0) die_lib(#call, rc__); \
  } while (0)

size_t piece_bytes() {
  const char* e = getenv("FQGPU_CHUNK_MB");
  size_t mb = e ? strtoull(e, nullptr, 10) : 128;  // (3 pinned slots of this size: fq_input.h; pinning them is part of the start-up)
  if (mb < 1) mb = 1;
  return mb << 20;
}

// ---- the four lines of one record, as the reference's buffers would hold them ---------------
struct RecordText {
  std::string l[4];  // with '\n' when present; c_str() cuts at an embedded NUL like the C strings do
};

// record `r` of a piece that starts at a record boundary
RecordText locate_record(const char* buf, size_t n, uint64_t r) {
  RecordText t;
  const char* p = buf;
  const char* end = buf + n;
  uint64_t line = 0;
  while (p < end && line < 4 * r) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    if (!nl) return t;
    p = nl + 1;
    ++line;
  }
  for (int k = 0; k < 4 && p < end; ++k) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    const char* stop = nl ? nl + 1 : end;
    t.l[k].assign(p, stop);
    p = stop;
  }
  return t;
}

// fastq_get_readname's result for a header line (src/fastq.c:442-516), for messages
std::string canonical_name(const std::string& hdr_line, const fqg_file_state& st) {
  std::string rn = hdr_line.c_str() + (hdr_line.empty() ? 0 : 1);  // strncpy(rn, &hdr[1], ...)
  long len;
  switch (st.readname_format) {
    case FQG_NAME_DEFAULT:
      len = (long)rn.size();
      if (st.is_pe) len--;
      if (len >= 1) rn.resize((size_t)len - 1);
      break;
    case FQG_NAME_INTEGER:
      len = (long)rn.size();
      if (len >= 1) rn.resize((size_t)len - 1);
      break;
    case FQG_NAME_CASAVA18: {
      size_t sp = rn.find(' ');
      if (sp == std::string::npos) sp = rn.size();
      rn.resize(sp);
      if (sp >= 2 && rn[sp - 2] == '/') rn.resize(sp - 2);
      break;
    }
  }
  return rn;
}

// fastq_qualRange2enc, src/fastq.c:274-297
const char* qual_range_to_enc(unsigned long min_qual, unsigned long max_qual) {
  static const char* names[] = {"33", "64", "solexa", "33 *", "sanger"};
  int enc;
  const unsigned int mn = (unsigned int)min_qual, mx = (unsigned int)max_qual;
  if (mn >= 33 && mn < 59 && mx >= 90) enc = 4;
  else if (mn >= 33 && mx <= 73) enc = 0;
  else if (mn < 59) enc = 0;
  else if (mn >= 64 && mx > 74) enc = 1;
  else if (mn >= 59 && mx > 74) enc = 2;
  else enc = 3;
  if (mx > FQG_MAX_PHRED_QUAL) return nullptr;
  if (enc != 4 && mx > mn + 60) return nullptr;
  return names[enc];
}

struct Probe {
  fqg_file_state st{};
  bool done = false;
  std::string format_line;  // text printed with the format decision
};

void probe_piece(Probe& pr, const char* buf, size_t n, int is_pe) {
  if (pr.done) return;
  pr.st.is_pe = is_pe;
  pr.st.readname_format = FQG_NAME_UNDEF;
  pr.st.space = FQG_SPACE_UNDEF;
  if (n == 0) return;
  if (fqg_probe_first_record(buf, n, is_pe, &pr.st) != 0) return;
  pr.done = true;
  // which of the two formats with value 2 it is decides the text (src/fastq.c:465-474)
  if (pr.st.readname_format == FQG_NAME_CASAVA18) pr.format_line = "CASAVA=1.8\n";
  else if (pr.st.readname_format == FQG_NAME_INTEGER) {
    const RecordText t = locate_record(buf, std::min<size_t>(n, 4096), 0);
    const std::string s = t.l[0].size() > 1 ? std::string(t.l[0].c_str() + 1) : std::string();
    regex_t rx;
    bool all_digits = false;
    if (regcomp(&rx, "^[0-9]+[\n\r]?$", REG_EXTENDED) == 0) {  // src/fastq.c:694
      all_digits = regexec(&rx, s.c_str(), 0, nullptr, 0) == 0;
      regfree(&rx);
    }
    pr.format_line = all_digits ? "Read name provided as an integer\n" : "Read name provided with no suffix\n";
  }
}

// set by a several-device pass that printed the line and then hands the file to a one-device loop (a NUL byte at a
// record start): that loop's first print_probe is the same line again
inline bool& probe_line_is_out() {
  static bool v = false;
  return v;
}
void print_probe(const Probe& pr) {
  if (probe_line_is_out()) {
    probe_line_is_out() = false;
    return;
  }
  fputs(pr.format_line.c_str(), stderr);
  if (pr.st.space == FQG_SPACE_COLOUR) fputs("Color space\n", stderr);
}

// PRINT_READS_PROCESSED (src/fastq.h:82) for the counts first..last
void ticker(uint64_t first, uint64_t last, uint64_t every, uint64_t scale = 1) {
  for (uint64_t c = (first + every - 1) / every * every; c <= last; c += every) {
    if (c == 0) continue;
    fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", (unsigned long)(c * scale));
    fflush(stderr);
  }
}

bool is_early_code(int code) {
  return code == FQG_E_TRUNCATED || code == FQG_E_HDR1_AT || code == FQG_E_HDR1_SHORT || code == FQG_E_SEQ_CHAR ||
         code == FQG_E_SEQ_UT || code == FQG_E_LEN_SMALL || code == FQG_E_HDR2_PLUS || code == FQG_E_LINE_TOO_LONG;
}

// One validation finding as text (src/fastq.c:300-392).  `cline` is FASTQ_FILE.cline at the time.
void print_validation_error(const char* fname, unsigned long cline, const fqg_validate_result& r,
                            const RecordText& t) {
  switch (r.code) {
    case FQG_E_HDR1_AT:
      FQ_PRINT_ERROR("Error in file %s: line %lu: sequence identifier should start with an @ - %s", fname, cline,
                     t.l[0].c_str());
      break;
    case FQG_E_HDR1_SHORT:
      FQ_PRINT_ERROR("Error in file %s: line %lu: sequence identifier should be longer than 1", fname, cline);
      break;
    case FQG_E_SEQ_CHAR:
      FQ_PRINT_ERROR(
          "Error in file %s: line %lu: invalid character '%c' (hex. code:'%x'), expected ACGTUacgtu0123nN.", fname,
          cline + 1, (char)r.aux0, (int)(char)r.aux0);
      break;
    case FQG_E_SEQ_UT:
      FQ_PRINT_ERROR("Error in file %s: line %lu: read contains both U and T bases", fname, cline - 2);
      break;
    case FQG_E_LEN_SMALL:
      FQ_PRINT_ERROR("Error in file %s: line %lu: read length too small - %lu", fname, cline + 1,
                     (unsigned long)r.aux0);
      break;
    case FQG_E_HDR2_PLUS:
      FQ_PRINT_ERROR(
          "Error in file %s: line %lu:  header2 wrong. The line should contain only '+' followed by a newline or "
          "read name (header1).",
          fname, cline + 2);
      break;
    case FQG_E_HDR2_DIFF:
      FQ_PRINT_ERROR("Error in file %s: line %lu:  header2 differs from header1\nheader 1 \"%s\"\nheader 2 \"%s\"",
                     fname, cline, t.l[0].c_str(), t.l[2].c_str());
      break;
    case FQG_E_QLEN:
      FQ_PRINT_ERROR("Error in file %s: line %lu: sequence and quality don't have the same length %lu!=%lu", fname,
                     cline, (unsigned long)r.aux0, (unsigned long)r.aux1);
      break;
    case FQG_E_QLEN_CS:
      FQ_PRINT_ERROR("Error in file %s: line %lu: sequence and quality length don't match %lu!=%lu", fname, cline,
                     (unsigned long)r.aux0, (unsigned long)r.aux1);
      break;
    default:
      FQ_PRINT_ERROR("Error in file %s: line %lu: unexpected outcome %d", fname, cline, r.code);
  }
}

[[noreturn]] void fail_truncated(const char* fname, unsigned long cline) {
  FQ_PRINT_ERROR("Error in file %s: line %lu: file truncated", fname, cline);  // src/fastq.c:255
  fqhost::leave(1);
}
// A line beyond the reference's gzgets limits (src/fastq.c:249-253).  Everything up to that record went the reference's
// way; from there on the reference reads the file in pieces of its own.  The program runs itself again, as a child, on
// input that is cut the same way while it is read (fq_respawn.h, fq_reframe.h), and leaves with the child's status.
// Only a stream that several devices were asked to share cannot be read twice.
[[noreturn]] void fail_too_long(const char* fname, uint64_t rec) {
  if (fqhost::reframe_supported() && strcmp(fname, "-") != 0 && !fqhost::reframing()) fqhost::respawn_reframed();
  FQ_PRINT_ERROR(
      "Error in file %s: record %lu has a line longer than the reference's line buffers (%d / %d bytes); the reference "
      "reads such a line in pieces, this program refuses it (fastq_info reproduces the pieces, on files and on a stream "
      "that one device reads)",
      fname, (unsigned long)(rec + 1), FQG_MAX_LABEL_LENGTH - 1, FQG_MAX_READ_LENGTH - 1);
  fqhost::leave(kExitSys);
}
[[noreturn]] void fail_wrong_header(const char* fname, unsigned long cline, const std::string& hdr) {
  FQ_PRINT_ERROR("Error in file %s: line %lu: wrong header %s", fname, cline, hdr.c_str());  // src/fastq.c:449
  fqhost::leave(kExitFormat);
}

struct Stats {
  unsigned long num_reads1 = 0;
  fqg_acc* acc1 = nullptr;
  fqg_acc* acc2 = nullptr;  // non-null when the reference would pass fd2 to median_rl()
};

// ---- default: fastq_index_readnames (src/fastq.c:396-439) ---------------------------------
struct IndexedFile {
  fqg_index* index = nullptr;
  fqg_file_state st{};
  uint64_t n_records = 0;
  uint64_t entries = 0, index_mem = 0;
  bool lookups = true;  // a second file will be looked up in it (false: the index is only the uniqueness test)
};

// (`in`: the opened file; whole: the file as one image - one retained frame - instead of pieces)
void run_index_input(Input& in, const char* path, int is_pe, Stats& S, IndexedFile& F, bool whole);
void run_index_file(const char* path, int is_pe, Stats& S, IndexedFile& F) {
  Input in(g_ctx, path, piece_bytes());
  run_index_input(in, path, is_pe, S, F, false);
}
void run_index_input(Input& in, const char* path, int is_pe, Stats& S, IndexedFile& F, bool whole) {
  Probe pr;
  uint64_t base = 0;
  bool info_pending = true;
  F.index_mem = 8;
  while (whole ? in.next(true) : in.next()) {
    probe_piece(pr, in.data(), in.size(), is_pe);
    fqg_validate_result r;
    LIB(fqg_validate(g_ctx, S.acc1, in.data(), in.size(), FQG_MEM_HOST, in.final() ? 1 : 0, &pr.st,
                     FQG_VALIDATE_COUNT_TWICE | (F.lookups ? FQG_VALIDATE_NAMES : FQG_VALIDATE_NAME_DIGESTS) | in.vflags(), &r));
    if (!F.index) {
      // sized from the first piece: a plain file holds about (its bytes / this piece's mean record) names - the table
      // then never has to be rebuilt at twice the size
      uint64_t expect = 1 << 20;
      if (r.n_records && r.consumed && in.plain_bytes() > in.size())
        expect = (uint64_t)((double)in.plain_bytes() / ((double)r.consumed / (double)r.n_records) * 1.05) + 1024;
      LIB(fqg_index_create(g_ctx, expect, &F.index));
      if (!F.lookups) LIB(fqg_index_expect_lookups(F.index, 0));
    }
    fqg_index_result ir{};
    if (r.n_records > 0) LIB(fqg_index_insert_unique(g_ctx, F.index, &pr.st, &ir));
    // which finding does the serial loop hit first?  per record: read (truncation), name
    // (wrong header), duplicate, validation
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
    if (ir.code == FQG_E_DUP_NAME) offer(ir.record, 2);
    if (info_pending && base == 0 && r.n_records > 0) {
      if (!(best_rec == 0 && best_stage <= 1)) print_probe(pr);
      info_pending = false;
    }
    if (best_rec != ~0ull) {
      const uint64_t R = base + best_rec;
      ticker(base + 1, R, 100000);
      const RecordText t = locate_record(in.data(), in.size(), best_rec);
      if (best_stage == 0) {
        if (r.code == FQG_E_LINE_TOO_LONG) fail_too_long(path, R);
        fail_truncated(path, 4 * R);
      }
      if (best_stage == 1) fail_wrong_header(path, 4 * (R + 1), t.l[0]);
      if (best_stage == 2) {
        FQ_PRINT_ERROR("Error in file %s: line %lu: duplicated sequence %s", path, (unsigned long)(4 * (R + 1)),
                       canonical_name(t.l[0], pr.st).c_str());
        fqhost::leave(kExitFormat);
      }
      print_validation_error(path, 4 * (R + 1), r, t);
      fqhost::leave(kExitFormat);
    }
    ticker(base + 1, base + r.n_records, 100000);
    base += r.n_records;
    if (r.n_records > 0) {
      F.entries = ir.n_entries;
      F.index_mem = ir.index_mem;
    }
    if (r.stopped) break;
    if (!in.final()) in.carry_from(r.consumed);
  }
  if (!F.index) LIB(fqg_index_create(g_ctx, 1024, &F.index));  // an empty file: an empty index
  F.st = pr.st;
  F.n_records = base;
}


}  // namespace
