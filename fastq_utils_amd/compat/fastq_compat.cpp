// libfastq_gpu.so - the reference's per-record C API (src/fastq.h:133-158, src/hash.h:64-78) on top of
// the bulk GPU library (include/fqg.h).  See include/fastq_gpu_compat.h.
//
// How a serial, per-record API is served by bulk kernels:
//   fastq_read_entry      first call on a file: the whole file is read through the caller's gzFile and
//                         framed on the GPU (line index -> one descriptor per record); the gzFile is
//                         then swapped for a stand-in with one byte per record, consumed one per call,
//                         so that the callers' own `while(!gzeof(fd->fd))` loops keep working.
//                         Every call copies the four lines of the next record into the FASTQ_ENTRY.
//   fastq_validate_entry  first call for a (file, file state) pair: ONE bulk validation
//                         (fqg_validate) gives the first failing record and its code.  A call for an
//                         earlier record returns 0; the call for that record prints the reference's
//                         message and returns 1.
//   fastq_index_readnames one bulk validation + fqg_index_insert_unique; findings are ordered and
//                         worded as the reference's loop (src/fastq.c:414-436) would.
//   fastq_index_lookup_header / fastq_index_delete   the pairing loops of fastq_info (:333-350) and of
//                         fastq_filterpair (src/fastq_filterpair.c:108-216): the first lookup of a file's
//                         records in an index runs ONE fqg_index_probe_delete of that file against it (which
//                         entry every record finds and takes, with the serial loop's answers); a file's
//                         lookups in its OWN index (filterpair's last loop) answer from fqg_index_alive.
//   fastq_rewind / fastq_quick_copy_entry / fastq_seek_copy_read   positions are record starts of the
//                         decompressed file (INDEX_ENTRY.entry_start): they move the record cursor.
// The host keeps only bookkeeping on values it already has (counters, copies, text).
#include <errno.h>
#include <fcntl.h>
#include <regex.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fastq_gpu_compat.h"
#include "../../include/fqg.h"
#include <sys/stat.h>

#include "../host/fq_pgzip.h"
#include "../host/fq_reframe.h"

namespace {

const char kVersion[] = "0.25.3";
enum { kExitParams = 1, kExitSys = 2, kExitFormat = 3 };

#define PRINT_ERROR(...)          \
  do {                            \
    fprintf(stderr, "\nERROR: "); \
    fprintf(stderr, __VA_ARGS__); \
    fprintf(stderr, "\n");        \
  } while (0)

fqg_ctx* g_ctx = nullptr;

fqg_ctx* gpu() {
  if (!g_ctx) {
    const int rc = fqg_open(0, &g_ctx);
    if (rc != 0) {
      PRINT_ERROR("no usable MI355X device (fqg_open: %d); this library has no CPU path", rc);
      exit(kExitSys);
    }
  }
  return g_ctx;
}

[[noreturn]] void die_lib(const char* what, int rc) {
  PRINT_ERROR("GPU library failure in %s (%d): %s", what, rc, g_ctx ? fqg_last_error(g_ctx) : "no context");
  exit(kExitSys);
}
#define LIB(call)                        \
  do {                                   \
    int rc__ = (call);                   \
    if (rc__ != 0) die_lib(#call, rc__); \
  } while (0)

struct Verdict {  // one bulk validation of a file under one file state
  fqg_file_state st;
  fqg_validate_result r;
  fqg_file_stats stats;
};

struct FileCtx {
  FASTQ_FILE* fd = nullptr;
  bool loaded = false;
  // the file's bytes - cut at the reference's gzgets limits when a line is beyond them (host/fq_reframe.h: a piece
  // that gzgets returns without its newline is followed by "\0\n"); `cuts` = where those two bytes lie in `image`.
  // Positions the caller sees (FASTQ_ENTRY.offset, INDEX_ENTRY.entry_start, the offsets it seeks to) are the FILE's.
  std::vector<char> image;
  std::vector<size_t> cuts;
  uint32_t vflags = 0;  // FQG_VALIDATE_REFRAMED once the image is cut
  std::vector<fqg_record> rec;
  uint64_t n_records = 0, next = 0;
  int tail_lines = 0;   // lines of an incomplete last record
  std::vector<Verdict> verdicts;
};

uint64_t to_file_offset(const FileCtx* f, uint64_t at) {  // image position (a record start) -> position in the file
  return at - 2 * (uint64_t)(std::lower_bound(f->cuts.begin(), f->cuts.end(), (size_t)at) - f->cuts.begin());
}
uint64_t to_image_offset(const FileCtx* f, uint64_t file_at) {  // ... and back: behind every cut made at or before it
  uint64_t k = 0;
  while (k < f->cuts.size() && f->cuts[k] - 2 * k <= file_at) ++k;  // (cuts are few: files with such lines are rare)
  return file_at + 2 * k;
}

std::unordered_map<FASTQ_FILE*, FileCtx*> g_files;
struct Origin {
  FileCtx* f;
  uint64_t record;
};
std::unordered_map<FASTQ_ENTRY*, Origin> g_origin;
Origin g_last_read{nullptr, 0};

struct IndexCtx {
  fqg_index* ix = nullptr;
  FileCtx* owner = nullptr;                                  // the file whose names it holds
  std::unordered_map<FileCtx*, std::vector<uint64_t>> asked; // per asking file: which entry each record took
  std::vector<uint8_t> alive;                                // entries nobody has taken (own-file lookups)
  bool alive_valid = false;
};
std::unordered_map<hashtable, IndexCtx*> g_indexes;

FileCtx* ctx_of(FASTQ_FILE* fd) {
  auto it = g_files.find(fd);
  if (it != g_files.end()) return it->second;
  FileCtx* f = new FileCtx();
  f->fd = fd;
  g_files[fd] = f;
  return f;
}

// a gzFile that reports end-of-file after exactly n successful one-byte reads
gzFile counting_stream(uint64_t n) {
  char name[] = "/tmp/fqgpu_stream_XXXXXX";
  const int h = mkstemp(name);
  if (h < 0) {
    PRINT_ERROR("unable to create a temporary file");
    exit(kExitSys);
  }
  unlink(name);
  std::vector<char> z(1 << 16, 'r');
  for (uint64_t left = n; left;) {
    const size_t k = (size_t)(left < z.size() ? left : z.size());
    if (write(h, z.data(), k) != (ssize_t)k) {
      PRINT_ERROR("unable to write a temporary file");
      exit(kExitSys);
    }
    left -= k;
  }
  lseek(h, 0, SEEK_SET);
  return gzdopen(h, "rb");
}

void load(FileCtx* f) {
  if (f->loaded) return;
  f->loaded = true;
  FASTQ_FILE* fd = f->fd;
  // A gzip file of some size is inflated on every core the process may use (host/fq_pgzip.h, as in the drop-in programs;
  // FQGPU_NO_PARALLEL_INFLATE / FQGPU_PGZIP_MIN / FQGPU_PGZIP_CHUNK as there); anything else through the gzFile that
  // fastq_new opened.  An inflate error ends the data where it is met - gzgets gives the reference NULL there.
  bool inflated = false;
  if (!(fd->filename[0] == '-' && fd->filename[1] == '\0') && !getenv("FQGPU_NO_PARALLEL_INFLATE") && fqhost::host_threads() > 1) {
    const int h = open(fd->filename, O_RDONLY);
    struct stat sb;
    unsigned char magic[2] = {0, 0};
    const char* e_min = getenv("FQGPU_PGZIP_MIN");
    const uint64_t min_bytes = e_min ? (uint64_t)std::max(0L, atol(e_min)) : (1u << 20);
    if (h >= 0 && fstat(h, &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size >= min_bytes && pread(h, magic, 2, 0) == 2 &&
        magic[0] == 0x1f && magic[1] == 0x8b) {
      const unsigned T = std::min(fqhost::host_threads(), 64u);
      size_t chunk = std::max<size_t>(512u << 10, std::min<size_t>(4u << 20, (128u << 20) / T));
      chunk = std::min<size_t>(chunk, std::max<size_t>((size_t)sb.st_size / T, 128u << 10));
      if (const char* e = getenv("FQGPU_PGZIP_CHUNK")) chunk = (size_t)std::max(4096L, atol(e));
      fqhost::ParallelGunzip pg(h, (uint64_t)sb.st_size, fd->filename, T, chunk);
      std::vector<char> piece(32u << 20);
      for (bool at_end = false; !at_end && !pg.failed();) {
        const size_t n = pg.read(piece.data(), piece.size(), &at_end);
        f->image.insert(f->image.end(), piece.data(), piece.data() + n);
      }
      inflated = true;
    }
    if (h >= 0) close(h);
  }
  if (!inflated) {
    char buf[1 << 16];
    int got;
    while ((got = gzread(fd->fd, buf, sizeof(buf))) > 0) f->image.insert(f->image.end(), buf, buf + got);
  }
  gzclose(fd->fd);
  {  // the reference's gzgets calls (src/fastq.c:249-253), as cuts
    fqhost::Reframer rf;
    std::string cut_image;
    bool clean = true;
    rf.run(f->image.data(), f->image.size(), true, cut_image, &clean, &f->cuts);
    if (!clean) {
      f->image.assign(cut_image.begin(), cut_image.end());
      f->vflags = FQG_VALIDATE_REFRAMED;
    } else f->cuts.clear();
  }
  fqg_file_state st;
  memset(&st, 0, sizeof(st));
  st.is_pe = fd->is_pe;
  st.readname_format = FQG_NAME_UNDEF;
  st.space = FQG_SPACE_UNDEF;
  fqg_validate_result r;
  LIB(fqg_validate(gpu(), nullptr, f->image.data(), f->image.size(), FQG_MEM_HOST, 1, &st,
                   FQG_VALIDATE_FRAME_ONLY | FQG_VALIDATE_NO_STATS | f->vflags, &r));
  f->n_records = r.n_records;
  f->tail_lines = r.tail_lines;
  f->rec.resize(r.n_records ? r.n_records : 1);
  if (r.n_records) LIB(fqg_frame_records(gpu(), 0, r.n_records, f->rec.data(), FQG_MEM_HOST));
  // one more byte when an incomplete record follows: the reference reads its first line before it fails
  fd->fd = counting_stream(f->n_records + (f->tail_lines ? 1 : 0));
}

const Verdict& verdict_for(FileCtx* f, const fqg_file_state& st) {
  for (const Verdict& v : f->verdicts)
    if (v.st.is_pe == st.is_pe && v.st.readname_format == st.readname_format && v.st.space == st.space) return v;
  Verdict v;
  v.st = st;
  fqg_acc* acc = nullptr;
  LIB(fqg_acc_create(gpu(), &acc));
  LIB(fqg_validate(gpu(), acc, f->image.data(), f->image.size(), FQG_MEM_HOST, 1, &st, f->vflags, &v.r));
  LIB(fqg_acc_read(acc, &v.stats));
  fqg_acc_destroy(acc);
  f->verdicts.push_back(v);
  return f->verdicts.back();
}

bool rx_match(const char* pattern, int flags, const char* s) {
  regex_t rx;
  if (regcomp(&rx, pattern, flags) != 0) return false;
  const bool m = regexec(&rx, s, 0, nullptr, 0) == 0;
  regfree(&rx);
  return m;
}

}  // namespace

extern "C" {

unsigned long index_mem = 0;
static char enc0[] = "33", enc1[] = "64", enc2[] = "solexa", enc3[] = "33 *", enc4[] = "sanger";
char* encodings[] = {enc0, enc1, enc2, enc3, enc4};

void fastq_print_version(void) { fprintf(stderr, "fastq_utils %s\n", kVersion); }

FASTQ_ENTRY* fastq_new_entry(void) {
  FASTQ_ENTRY* e = (FASTQ_ENTRY*)malloc(sizeof(FASTQ_ENTRY));
  if (!e) {
    PRINT_ERROR("unable to allocate %ld bytes of memory", (long)sizeof(FASTQ_ENTRY));
    exit(kExitSys);
  }
  e->hdr1[0] = e->hdr2[0] = e->seq[0] = e->qual[0] = '\0';
  e->read_len = 0;
  e->offset = 0;
  return e;
}

gzFile fastq_open(const char* filename, const char* mode) {  // src/fastq.c:631-664
  gzFile fd1;
  if (filename[0] == '-' && filename[1] == '\0') {
    fd1 = mode[0] == 'r' ? gzdopen(fileno(stdin), "rb") : gzdopen(fileno(stdout), mode);
    if (fd1 == nullptr) {
      PRINT_ERROR("Unable to gzdopen %s", mode[0] == 'r' ? "stdin" : "stdout");
      exit(kExitParams);
    }
  } else {
    fd1 = gzopen(filename, mode);
    if (fd1 == nullptr) {
      PRINT_ERROR("Unable to open %s", filename);
      exit(kExitParams);
    }
  }
  gzbuffer(fd1, 128000);
  return fd1;
}

FASTQ_FILE* fastq_new(const char* filename, const int fix_dot, const char* mode) {  // src/fastq.c:163-188
  FASTQ_FILE* n = (FASTQ_FILE*)calloc(1, sizeof(FASTQ_FILE));
  if (!n) {
    PRINT_ERROR("Error while processing file %s: unable to allocate %ld bytes of memory", filename,
                (long)sizeof(FASTQ_FILE));
    exit(kExitSys);
  }
  n->min_rl = FQC_MAX_READ_LENGTH;
  n->min_qual = FQG_MAX_PHRED_QUAL;
  n->fix_dot = fix_dot;
  n->readname_format = FQG_NAME_UNDEF;
  n->is_casava_18 = -1;
  n->space = FQG_SPACE_UNDEF;
  strncpy(n->filename, filename, FQC_MAX_FILENAME_LENGTH - 1);
  n->fd = fastq_open(filename, mode);
  return n;
}

void fastq_destroy(FASTQ_FILE* fd) {
  if (fd->fd && gzclose(fd->fd) != Z_OK) {
    PRINT_ERROR("unable to close file descriptor");
    exit(kExitSys);
  }
}

void fastq_is_pe(FASTQ_FILE* fd) { fd->is_pe = 1; }
unsigned long get_elength(FASTQ_ENTRY* m) { return m->read_len - 2; }

void fastq_new_entry_stats(FASTQ_FILE* fd, FASTQ_ENTRY* entry) {  // src/fastq.c:97-110
  const unsigned long slen = entry->read_len;
  if (slen < fd->min_rl) fd->min_rl = slen;
  if (slen > fd->max_rl) fd->max_rl = slen;
  ++fd->num_rds;
  fd->last_rl = slen;
  if (slen < (unsigned long)FQC_MAX_READ_LENGTH) fd->rdlen_ctr[slen]++;
}

int fastq_read_entry(FASTQ_FILE* fd, FASTQ_ENTRY* e) {  // src/fastq.c:245-261
  FileCtx* f = ctx_of(fd);
  load(f);
  e->offset = (long long)to_file_offset(f, f->next < f->n_records ? f->rec[f->next].offset : (uint64_t)f->image.size());
  if (gzeof(fd->fd)) return 0;
  if (gzgetc(fd->fd) < 0) {
    e->hdr1[0] = '\0';
    return 0;
  }
  if (f->next >= f->n_records) {  // the incomplete last record
    PRINT_ERROR("Error in file %s: line %lu: file truncated", fd->filename, fd->cline);
    exit(1);
  }
  const fqg_record& d = f->rec[f->next];
  // a line of the image into one of the caller's buffers: a piece that ends in the two bytes of a cut is the piece
  // alone - what gzgets left in the buffer (limit - 1 bytes and the NUL)
  const char* p = f->image.data() + d.offset;
  auto line_to = [&](char* dst, size_t cap, uint32_t len) {
    size_t n = len;
    if (f->vflags && n >= 2 && p[n - 2] == '\0' && p[n - 1] == '\n') n -= 2;
    if (n > cap - 1) {  // (cannot be: the image was cut at the limits)
      PRINT_ERROR("Error in file %s: record %lu has a line longer than the reference's line buffers", fd->filename,
                  (unsigned long)(f->next + 1));
      exit(kExitSys);
    }
    memcpy(dst, p, n);
    dst[n] = '\0';
    p += len;
  };
  line_to(e->hdr1, sizeof(e->hdr1), d.hdr1_len);
  line_to(e->seq, sizeof(e->seq), d.seq_len);
  line_to(e->hdr2, sizeof(e->hdr2), d.hdr2_len);
  line_to(e->qual, sizeof(e->qual), d.qual_len);
  fd->cline += 4;
  e->read_len = d.read_len;
  g_origin[e] = Origin{f, f->next};
  g_last_read = Origin{f, f->next};
  ++f->next;
  return 1;
}

int fastq_read_next_entry(FASTQ_FILE* fd, FASTQ_ENTRY* e) {
  const int r = fastq_read_entry(fd, e);
  if (r <= 0) return r;
  fastq_new_entry_stats(fd, e);
  return 1;
}

char* fastq_get_readname(FASTQ_FILE* fd, FASTQ_ENTRY* e, char* rn, unsigned long* len_p, int is_header1) {
  // src/fastq.c:442-516
  unsigned long len = 0;
  char* hdr = is_header1 ? e->hdr1 : e->hdr2;
  if (is_header1 == 1 && hdr[0] != '@') {
    PRINT_ERROR("Error in file %s: line %lu: wrong header %s", fd->filename, fd->cline, hdr);
    exit(kExitFormat);
  }
  strncpy(rn, &hdr[1], FQC_MAX_LABEL_LENGTH - 1);
  if (fd->readname_format == FQG_NAME_UNDEF) {
    const int fmt = fqg_probe_readname_format(rn);
    fd->is_casava_18 = fmt == FQG_NAME_CASAVA18;
    fd->readname_format = fmt;
    if (fmt == FQG_NAME_CASAVA18) fprintf(stderr, "CASAVA=1.8\n");
    else if (fmt == FQG_NAME_INTEGER)  // INTEGERNAME and NOP share the value; the text tells them apart
      fprintf(stderr, rx_match("^[0-9]+[\n\r]?$", REG_EXTENDED, rn) ? "Read name provided as an integer\n"
                                                                      : "Read name provided with no suffix\n");
  }
  if (fd->space == FQG_SPACE_UNDEF) {
    fd->space = fqg_probe_space(e->seq);
    if (fd->space == FQG_SPACE_COLOUR) fprintf(stderr, "Color space\n");
  }
  switch (fd->readname_format) {
    case FQG_NAME_DEFAULT:
      len = strlen(rn);
      if (fd->is_pe) len--;
      rn[len - 1] = '\0';
      break;
    case FQG_NAME_INTEGER:
      len = strlen(rn);
      rn[len - 1] = '\0';
      break;
    case FQG_NAME_CASAVA18:
      len = 0;
      while (rn[len] != ' ' && rn[len] != '\0') ++len;
      rn[len] = '\0';
      if (len >= 2 && rn[len - 2] == '/') {
        rn[len - 2] = '\0';
        len = len - 2;
      }
      break;
  }
  *len_p = len;
  return rn;
}

// message of one validation finding (src/fastq.c:300-392); cline = fd->cline at the time
static void print_finding(FASTQ_FILE* fd, FASTQ_ENTRY* e, const fqg_validate_result& r) {
  const char* fn = fd->filename;
  const unsigned long cl = fd->cline;
  switch (r.code) {
    case FQG_E_HDR1_AT:
      PRINT_ERROR("Error in file %s: line %lu: sequence identifier should start with an @ - %s", fn, cl, e->hdr1);
      break;
    case FQG_E_HDR1_SHORT:
      PRINT_ERROR("Error in file %s: line %lu: sequence identifier should be longer than 1", fn, cl);
      break;
    case FQG_E_SEQ_CHAR:
      PRINT_ERROR("Error in file %s: line %lu: invalid character '%c' (hex. code:'%x'), expected ACGTUacgtu0123nN.", fn,
                  cl + 1, (char)r.aux0, (int)(char)r.aux0);
      break;
    case FQG_E_SEQ_UT:
      PRINT_ERROR("Error in file %s: line %lu: read contains both U and T bases", fn, cl - 2);
      break;
    case FQG_E_LEN_SMALL:
      PRINT_ERROR("Error in file %s: line %lu: read length too small - %lu", fn, cl + 1, (unsigned long)r.aux0);
      break;
    case FQG_E_HDR2_PLUS:
      PRINT_ERROR(
          "Error in file %s: line %lu:  header2 wrong. The line should contain only '+' followed by a newline or read "
          "name (header1).",
          fn, cl + 2);
      break;
    case FQG_E_HDR2_DIFF:
      PRINT_ERROR("Error in file %s: line %lu:  header2 differs from header1\nheader 1 \"%s\"\nheader 2 \"%s\"", fn, cl,
                  e->hdr1, e->hdr2);
      break;
    case FQG_E_QLEN:
      PRINT_ERROR("Error in file %s: line %lu: sequence and quality don't have the same length %lu!=%lu", fn, cl,
                  (unsigned long)r.aux0, (unsigned long)r.aux1);
      break;
    case FQG_E_QLEN_CS:
      PRINT_ERROR("Error in file %s: line %lu: sequence and quality length don't match %lu!=%lu", fn, cl,
                  (unsigned long)r.aux0, (unsigned long)r.aux1);
      break;
    default:
      PRINT_ERROR("Error in file %s: line %lu: unexpected outcome %d", fn, cl, r.code);
  }
}

static bool is_before_stats(int code) {  // findings raised before fastq_new_entry_stats runs (:306-341)
  return code == FQG_E_HDR1_AT || code == FQG_E_HDR1_SHORT || code == FQG_E_SEQ_CHAR || code == FQG_E_SEQ_UT;
}
static bool is_before_names(int code) {  // ... before the header-2 comparison calls fastq_get_readname (:363)
  return is_before_stats(code) || code == FQG_E_LEN_SMALL || code == FQG_E_HDR2_PLUS;
}

int fastq_validate_entry(FASTQ_FILE* fd, FASTQ_ENTRY* e) {
  auto it = g_origin.find(e);
  if (it == g_origin.end()) {
    PRINT_ERROR("fastq_validate_entry: the entry was not filled by fastq_read_entry of this library");
    exit(kExitSys);
  }
  FileCtx* f = it->second.f;
  const uint64_t r = it->second.record;
  // the file state the reference would use: the one of `fd`, decided by its first fastq_get_readname call
  fqg_file_state st;
  memset(&st, 0, sizeof(st));
  st.is_pe = fd->is_pe;
  st.readname_format = fd->readname_format != FQG_NAME_UNDEF ? fd->readname_format : fqg_probe_readname_format(e->hdr1 + 1);
  st.space = fd->space != FQG_SPACE_UNDEF ? fd->space : fqg_probe_space(e->seq);
  const Verdict& v = verdict_for(f, st);
  const bool failing = v.r.code != FQG_OK && v.r.code != FQG_E_TRUNCATED && v.r.record == r;
  if (!(failing && is_before_stats(v.r.code))) fastq_new_entry_stats(fd, e);
  if (!(failing && is_before_names(v.r.code))) {
    // the comparison of the two headers is where the reference fixes the file's name format and space
    char rn[FQC_MAX_LABEL_LENGTH];
    unsigned long len;
    if (fd->readname_format == FQG_NAME_UNDEF || fd->space == FQG_SPACE_UNDEF) fastq_get_readname(fd, e, rn, &len, 1);
  }
  if (failing) {
    print_finding(fd, e, v.r);
    return 1;
  }
  // quality range: the bulk statistics of the file hold it (ranges only ever widen, and the callers
  // read them after the last record; a file with a finding never gets that far)
  if (v.r.code == FQG_OK && v.stats.num_rds) {
    if (v.stats.min_qual < fd->min_qual) fd->min_qual = v.stats.min_qual;
    if (v.stats.max_qual > fd->max_qual) fd->max_qual = v.stats.max_qual;
  }
  return 0;
}

void fastq_index_readnames(FASTQ_FILE* fd1, hashtable index, long long start_offset, int replace_dots) {
  // src/fastq.c:396-439 as ONE bulk pass: validation with every record counted twice (:415, :432) and
  // the GPU read-name index; the first finding in (record, stage) order wins, as in the serial loop
  fd1->fix_dot = replace_dots;
  if (fd1->fd == nullptr) {
    PRINT_ERROR("Unable to open %s", fd1->filename);
    exit(kExitParams);
  }
  if (start_offset > 0) {
    PRINT_ERROR(" Not implemented");
    exit(kExitSys);
  }
  FileCtx* f = ctx_of(fd1);
  load(f);
  fqg_ctx* c = gpu();
  fqg_file_state st;
  memset(&st, 0, sizeof(st));
  st.readname_format = FQG_NAME_UNDEF;
  st.space = FQG_SPACE_UNDEF;
  if (!f->image.empty()) fqg_probe_first_record(f->image.data(), f->image.size(), fd1->is_pe, &st);
  st.is_pe = fd1->is_pe;
  fqg_acc* acc = nullptr;
  LIB(fqg_acc_create(c, &acc));
  fqg_validate_result r;
  LIB(fqg_validate(c, acc, f->image.data(), f->image.size(), FQG_MEM_HOST, 1, &st, FQG_VALIDATE_COUNT_TWICE | FQG_VALIDATE_NAMES | f->vflags, &r));
  fqg_index* ix = nullptr;
  LIB(fqg_index_create(c, r.n_records, &ix));
  fqg_index_result ir;
  memset(&ir, 0, sizeof(ir));
  if (r.n_records) LIB(fqg_index_insert_unique(c, ix, &st, &ir));
  // order of events for record k: read (truncation), name (wrong header, format lines), duplicate, validation
  const uint64_t none = ~0ull;
  const uint64_t rec_trunc = r.code == FQG_E_TRUNCATED ? r.record : none;
  const uint64_t rec_valid = (r.code != FQG_OK && r.code != FQG_E_TRUNCATED) ? r.record : none;
  const uint64_t rec_wrong = ir.code == FQG_E_WRONG_HEADER ? ir.record : none;
  const uint64_t rec_dup = ir.code == FQG_E_DUP_NAME ? ir.record : none;
  uint64_t first = std::min(std::min(rec_trunc, rec_valid), std::min(rec_wrong, rec_dup));
  auto entry_text = [&](uint64_t k, FASTQ_ENTRY* e) {
    const fqg_record& d = f->rec[k];
    const char* p = f->image.data() + d.offset;
    snprintf(e->hdr1, sizeof(e->hdr1), "%.*s", (int)d.hdr1_len, p);
    snprintf(e->seq, 4096, "%.*s", (int)std::min<uint32_t>(d.seq_len, 4000), p + d.hdr1_len);
    snprintf(e->hdr2, sizeof(e->hdr2), "%.*s", (int)d.hdr2_len, p + d.hdr1_len + d.seq_len);
  };
  FASTQ_ENTRY* e = fastq_new_entry();
  if (f->n_records > 0 && !(first == 0 && rec_wrong == 0)) {
    // the first record's name fixes the format (printed once); not reached if its header is wrong
    entry_text(0, e);
    char rn[FQC_MAX_LABEL_LENGTH];
    unsigned long len;
    fastq_get_readname(fd1, e, rn, &len, 1);
  }
  if (first != none) {
    // the ticker the reference printed before it stopped (PRINT_READS_PROCESSED, src/fastq.h:82)
    for (uint64_t k = 100000; k <= first; k += 100000) fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", (unsigned long)k);
    if (first == rec_trunc && rec_trunc <= std::min(rec_wrong, std::min(rec_dup, rec_valid))) {
      PRINT_ERROR("Error in file %s: line %lu: file truncated", fd1->filename, (unsigned long)(4 * first));
      exit(1);
    }
    fd1->cline = 4 * (first + 1);
    entry_text(first, e);
    if (first == rec_wrong) {
      PRINT_ERROR("Error in file %s: line %lu: wrong header %s", fd1->filename, fd1->cline, e->hdr1);
      exit(kExitFormat);
    }
    if (first == rec_dup) {
      char rn[FQC_MAX_LABEL_LENGTH];
      unsigned long len;
      fastq_get_readname(fd1, e, rn, &len, 1);
      PRINT_ERROR("Error in file %s: line %lu: duplicated sequence %s", fd1->filename, fd1->cline, rn);
      exit(kExitFormat);
    }
    print_finding(fd1, e, r);
    exit(kExitFormat);
  }
  for (uint64_t k = 100000; k <= f->n_records; k += 100000) fprintf(stderr, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", (unsigned long)k);
  free(e);
  // hand the statistics to the FASTQ_FILE the caller reads (src/fastq_info.c:316-319, median_rl :39-55)
  fqg_file_stats s;
  LIB(fqg_acc_read(acc, &s));
  fd1->num_rds = s.num_rds;
  if (s.num_rds) {
    fd1->min_rl = s.min_rl;
    fd1->max_rl = s.max_rl;
    fd1->min_qual = s.min_qual;
    fd1->max_qual = s.max_qual;
  }
  size_t nb = 0;
  LIB(fqg_acc_hist_nonzero(acc, nullptr, nullptr, 0, &nb));
  std::vector<uint64_t> lens(nb ? nb : 1), cnts(nb ? nb : 1);
  LIB(fqg_acc_hist_nonzero(acc, lens.data(), cnts.data(), nb, &nb));
  for (size_t k = 0; k < nb; ++k)
    if (lens[k] < (uint64_t)FQC_MAX_READ_LENGTH) fd1->rdlen_ctr[lens[k]] = cnts[k];
  fqg_acc_destroy(acc);
  fd1->cline = 4 * f->n_records;
  f->next = f->n_records;
  while (gzgetc(fd1->fd) >= 0) {
  }
  index->n_entries = ir.n_entries;
  index_mem += ir.index_mem >= 8 ? ir.index_mem - 8 : 0;  // (the caller adds sizeof(hashtable) itself)
  IndexCtx* ic = new IndexCtx();
  ic->ix = ix;
  ic->owner = f;
  g_indexes[index] = ic;
}

INDEX_ENTRY* fastq_index_lookup_header(hashtable sn_index, char* hdr) {
  static INDEX_ENTRY found;
  auto it = g_indexes.find(sn_index);
  if (it == g_indexes.end()) return nullptr;
  IndexCtx* ic = it->second;
  FileCtx* f = g_last_read.f;  // the name is the one of the record read last (every caller's loop does that)
  if (!f) return nullptr;
  const uint64_t k = g_last_read.record;
  found.hdr = hdr;
  if (f == ic->owner) {
    // a file asking its own index: the entry is there unless another file's record took it
    if (!ic->alive_valid) {
      ic->alive.assign(f->n_records ? f->n_records : 1, 0);
      if (f->n_records) LIB(fqg_index_alive(gpu(), ic->ix, ic->alive.data(), ic->alive.size()));
      ic->alive_valid = true;
    }
    if (k >= f->n_records || !ic->alive[k]) return nullptr;
    found.entry_start = (off_t)to_file_offset(f, f->rec[k].offset);
    return &found;
  }
  auto asked = ic->asked.find(f);
  if (asked == ic->asked.end()) {
    // the whole loop of this file against the index at once (src/fastq_info.c:333-350, src/fastq_filterpair.c:
    // 108-170): which entry every record finds and takes
    fqg_file_state st;
    memset(&st, 0, sizeof(st));
    st.is_pe = f->fd->is_pe;
    st.readname_format = f->fd->readname_format;
    st.space = f->fd->space;
    fqg_validate_result r;
    LIB(fqg_validate(gpu(), nullptr, f->image.data(), f->image.size(), FQG_MEM_HOST, 1, &st,
                     FQG_VALIDATE_FRAME_ONLY | FQG_VALIDATE_NO_STATS | FQG_VALIDATE_NAMES | f->vflags, &r));
    std::vector<uint64_t> m(r.n_records ? r.n_records : 1, FQG_NO_MATCH);
    fqg_index_result ir;
    if (r.n_records) LIB(fqg_index_probe_delete(gpu(), ic->ix, &st, m.data(), &ir));
    asked = ic->asked.emplace(f, std::move(m)).first;
    ic->alive_valid = false;
  }
  if (k >= asked->second.size() || asked->second[k] >= FQG_MATCH_WRONG_HEADER) return nullptr;
  found.entry_start = (off_t)to_file_offset(ic->owner, ic->owner->rec[asked->second[k]].offset);
  return &found;
}

void fastq_index_delete(char* rname, hashtable index) {
  (void)rname;
  if (index->n_entries) --index->n_entries;
}

// ---- positions (src/fastq.c:77-80, 124-157, 191-199) -----------------------------------------------------------
static uint64_t cursor_offset(const FileCtx* f) {  // (the FILE's offset of the record the cursor is at)
  return to_file_offset(f, f->next < f->n_records ? f->rec[f->next].offset : (uint64_t)f->image.size());
}
// the record that starts at `offset`, n_records for the end of the file, ~0 for anything else
static uint64_t record_at(const FileCtx* f, uint64_t file_offset) {
  const uint64_t offset = to_image_offset(f, file_offset);
  if (offset >= f->image.size()) return offset == f->image.size() ? f->n_records : ~0ull;
  uint64_t lo = 0, hi = f->n_records;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (f->rec[mid].offset < offset) lo = mid + 1;
    else hi = mid;
  }
  return (lo < f->n_records && f->rec[lo].offset == offset) ? lo : ~0ull;
}
static void move_cursor(FileCtx* f, uint64_t k) {
  f->next = k;
  gzseek(f->fd->fd, (z_off_t)k, SEEK_SET);  // the stand-in stream holds one byte per record
}

void fastq_rewind(FASTQ_FILE* fd) {
  fd->cline = 1;
  FileCtx* f = ctx_of(fd);
  if (f->loaded) move_cursor(f, 0);
  else gzrewind(fd->fd);
}

static unsigned long ctr_seek = 0, ctr_noseek = 0;
static void seek_to(FileCtx* f, long offset) {
  const uint64_t k = offset < 0 ? ~0ull : record_at(f, (uint64_t)offset);
  if (k == ~0ull) {  // not a record start: the reference would read lines from the middle of a record
    PRINT_ERROR("Error in file %s: line %lu: gzseek failed", f->fd->filename, f->fd->cline);
    exit(kExitSys);
  }
  move_cursor(f, k);
}

void fastq_quick_copy_entry(long offset, FASTQ_FILE* from, FASTQ_FILE* to) {
  FileCtx* f = ctx_of(from);
  load(f);
  if ((long)cursor_offset(f) != offset) {
    seek_to(f, offset);
    ++ctr_seek;
  } else ++ctr_noseek;
  fprintf(stderr, "%lu / %lu\n", ctr_seek, ctr_noseek);
  if (f->next >= f->n_records) {  // nothing (complete) to read there
    PRINT_ERROR("Error in file %s: line %lu: file truncated", from->filename, from->cline);
    exit(kExitFormat);
  }
  const fqg_record& d = f->rec[f->next];
  const char* p = f->image.data() + d.offset;
  const uint32_t len[4] = {d.hdr1_len, d.seq_len, d.hdr2_len, d.qual_len};
  std::string line;
  for (int i = 0; i < 4; ++i) {
    line.assign(p, len[i]);
    GZ_WRITE(to->fd, const_cast<char*>(line.c_str()));
    p += len[i];
  }
  move_cursor(f, f->next + 1);
  from->cur_offset = (long long)cursor_offset(f);
}

void fastq_seek_copy_read(long offset, FASTQ_FILE* from, FASTQ_FILE* to) {
  static FASTQ_ENTRY* tmp = nullptr;
  FileCtx* f = ctx_of(from);
  load(f);
  seek_to(f, offset);
  if (!tmp) tmp = fastq_new_entry();
  fastq_read_entry(from, tmp);
  fastq_write_entry(to, tmp);
}

char* fastq_qualRange2enc(unsigned int min_qual, unsigned int max_qual) {  // src/fastq.c:274-297
  int enc;
  if (min_qual >= 33 && min_qual < 59 && max_qual >= 90) enc = 4;
  else if (min_qual >= 33 && max_qual <= 73) enc = 0;
  else if (min_qual < 59) enc = 0;
  else if (min_qual >= 64 && max_qual > 74) enc = 1;
  else if (min_qual >= 59 && max_qual > 74) enc = 2;
  else enc = 3;
  if (max_qual > FQG_MAX_PHRED_QUAL) return nullptr;
  if (enc != 4 && max_qual > min_qual + 60) return nullptr;
  return encodings[enc];
}

void GZ_WRITE(gzFile fd, char* s) {  // src/fastq.c:211-235
  int n = gzputs(fd, s);
  if (n > 0) return;
  if (*s == '\0') return;
  const char* errmsg = gzerror(fd, &n);
  PRINT_ERROR("%s.\n", errmsg);
  exit(kExitSys);
}

void fastq_write_entry(FASTQ_FILE* fd, FASTQ_ENTRY* e) {
  GZ_WRITE(fd->fd, e->hdr1);
  GZ_WRITE(fd->fd, e->seq);
  GZ_WRITE(fd->fd, e->hdr2);
  GZ_WRITE(fd->fd, e->qual);
}

void fastq_write_entry2stdout(FASTQ_ENTRY* e) {
  fprintf(stdout, "%s", e->hdr1);
  fprintf(stdout, "%s", e->seq);
  fprintf(stdout, "%s", e->hdr2);
  fprintf(stdout, "%s", e->qual);
}

// ---- hash.h: a chained table keyed by unsigned long long (the container API; the read-name work of
// fastq.c does not go through it here) -------------------------------------------------------------
hashtable new_hashtable(fq_ulong hashsize) {  // src/hash.c:106-123
  hashtable t = (hashtable)malloc(sizeof(struct hashtable_s));
  if (!t) return nullptr;
  t->buckets = (hashnode**)calloc((size_t)hashsize, sizeof(hashnode*));
  t->buckets_last = (hashnode**)calloc((size_t)hashsize, sizeof(hashnode*));
  if (!t->buckets || !t->buckets_last) return nullptr;
  t->size = hashsize;
  t->last_bucket = 0;
  t->n_entries = 0;
  t->last_node = nullptr;
  return t;
}

int insere(hashtable t, fq_ulong key, void* obj) {  // append at the tail of the chain, duplicates allowed
  hashnode* n = (hashnode*)malloc(sizeof(hashnode));
  if (!n) return -1;
  n->value = key;
  n->obj = obj;
  n->next = nullptr;
  const fq_ulong b = key % t->size;
  if (t->buckets[b] == nullptr) t->buckets[b] = n;
  else t->buckets_last[b]->next = n;
  t->buckets_last[b] = n;
  t->n_entries++;
  return 0;
}

void* get_object(hashtable t, fq_ulong key) {
  hashnode* b = t->buckets[key % t->size];
  while (b != nullptr) {
    if (b->value == key) {
      t->last_node = b;
      return b->obj;
    }
    b = b->next;
  }
  return nullptr;
}

void* get_next_object(hashtable t, fq_ulong key) {
  if (t->last_node == nullptr) return nullptr;
  hashnode* b = t->last_node->next;
  while (b != nullptr) {
    if (b->value == key) {
      t->last_node = b;
      return b->obj;
    }
    b = b->next;
  }
  return nullptr;
}

void* fqc_delete(hashtable t, fq_ulong key, void* obj) __asm__("delete");
void* fqc_delete(hashtable t, fq_ulong key, void* obj) {
  const fq_ulong bi = key % t->size;
  hashnode *b = t->buckets[bi], *prev = nullptr;
  while (b != nullptr) {
    if (b->value == key && b->obj == obj) {
      if (prev) prev->next = b->next;
      else t->buckets[bi] = b->next;
      if (t->buckets_last[bi] == b) t->buckets_last[bi] = prev;
      free(b);
      t->n_entries--;
      return obj;
    }
    prev = b;
    b = b->next;
  }
  return nullptr;
}

void reset_hashtable(hashtable t) {
  for (fq_ulong i = 0; i < t->size; ++i) {
    hashnode* b = t->buckets[i];
    while (b) {
      hashnode* nx = b->next;
      free(b);
      b = nx;
    }
    t->buckets[i] = t->buckets_last[i] = nullptr;
  }
  t->n_entries = 0;
}

void free_hashtable(hashtable t) {
  if (!t) return;
  reset_hashtable(t);
  free(t->buckets);
  free(t->buckets_last);
  free(t);
}

void init_hash_traversal(hashtable t) {
  t->last_bucket = 0;
  t->last_node = nullptr;
}

void* next_hashnode(hashtable t) {  // (the reference's traversal starts at bucket 1: src/hash.c:223-251)
  if (t->last_node != nullptr && t->last_node->next != nullptr) {
    t->last_node = t->last_node->next;
    return t->last_node;
  }
  while (t->last_bucket + 1 < t->size) {
    ++t->last_bucket;
    if (t->buckets[t->last_bucket] != nullptr) {
      t->last_node = t->buckets[t->last_bucket];
      return t->last_node;
    }
  }
  return nullptr;
}

void* next_hash_object(hashtable t) {
  hashnode* n = (hashnode*)next_hashnode(t);
  return n ? n->obj : nullptr;
}

void hashtable_stats(hashtable t) {  // src/hash.c:125-150: empty buckets, chains longer than one
  fq_ulong zbuckets = 0, collisions = 0, max_col = 0;
  for (fq_ulong i = 0; i < t->size; ++i) {
    fq_ulong ctr = 0;
    for (const hashnode* b = t->buckets[i]; b; b = b->next) ++ctr;
    if (!ctr) ++zbuckets;
    else if (ctr > 1) {
      collisions += ctr;
      if (ctr > max_col) max_col = ctr;
    }
  }
  fprintf(stderr, "size: %llu\n", t->size);
  fprintf(stderr, "max. col: %llu\n", max_col);
  fprintf(stderr, "zbuckets: %llu\n", zbuckets);
  fprintf(stderr, "%% zbuckets: %.2f\n", zbuckets * 1.0 / t->size);
  fprintf(stderr, "collisions: %llu\n", collisions);
  fprintf(stderr, "avg. collisions: %.2f\n", collisions * 1.0 / (t->size - zbuckets));
}

}  // extern "C"
