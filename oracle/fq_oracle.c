/*
 * fq_oracle.c - TEST INFRASTRUCTURE, not product code.
 *
 * A sequential CPU restatement of the reference's fastq_info path (nunofonseca/fastq_utils
 * 0.25.3), working on whole decompressed file images held in memory.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; nothing under
 * fastq_utils_amd/ or bin/ links or calls it.
 *
 * Parity is pinned: tests/test_oracle.py runs every golden vector under tests/golden/ (the
 * reference's own fixtures, with outputs captured from the reference binary built by
 * oracle/Makefile into oracle/_ref/) through this file and requires identical exit status,
 * stdout and stderr text.
 *
 * Each function names the reference lines it restates (paths relative to the reference tree).
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <regex.h>

#include "../include/fqg_codes.h"
#include "fq_oracle.h"

/* ------------------------------------------------------------------------------------------
 * run context: captured stdout/stderr, exit() emulation, first structured outcome
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  char *p;
  size_t n, cap;
} sink_t;

typedef struct {
  sink_t out, err;
  jmp_buf bail;
  int status;
  fqo_outcome first; /* structured copy of the first record-level error */
  unsigned long index_mem;
  void *owned[64]; /* heap blocks released when the run ends, whichever way it ends */
  int n_owned;
  struct nameset_s *idx;
  void *map[2]; /* fastq_filterpair's indexes */
  sink_t files[3]; /* ... and its three outputs (in the heap block: read again after the longjmp) */
} run_t;

static void *own(run_t *r, void *p) {
  if (r->n_owned < 64) r->owned[r->n_owned++] = p;
  return p;
}

static void sink_put(sink_t *s, const char *fmt, va_list ap) {
  va_list ap2;
  va_copy(ap2, ap);
  int need = vsnprintf(NULL, 0, fmt, ap2);
  va_end(ap2);
  if (need < 0) return;
  if (s->n + (size_t)need + 1 > s->cap) {
    size_t cap = s->cap ? s->cap * 2 : 4096;
    while (cap < s->n + (size_t)need + 1) cap *= 2;
    s->p = (char *)realloc(s->p, cap);
    s->cap = cap;
  }
  vsnprintf(s->p + s->n, (size_t)need + 1, fmt, ap);
  s->n += (size_t)need;
}
static void eprintf(run_t *r, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  sink_put(&r->err, fmt, ap);
  va_end(ap);
}
static void oprintf(run_t *r, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  sink_put(&r->out, fmt, ap);
  va_end(ap);
}
static void leave(run_t *r, int status) {
  r->status = status;
  longjmp(r->bail, 1);
}
static void note(run_t *r, int code, uint64_t record, uint64_t line, uint64_t a0, uint64_t a1) {
  if (r->first.code != FQG_OK) return;
  r->first.code = code;
  r->first.record = record;
  r->first.line = line;
  r->first.aux0 = a0;
  r->first.aux1 = a1;
}
/* src/fastq.h:68-70 PRINT_INFO / PRINT_ERROR */
#define ERR_OPEN(r) eprintf(r, "\nERROR: ")
#define ERR_CLOSE(r) eprintf(r, "\n")

/* ------------------------------------------------------------------------------------------
 * line input: zlib gzgets()/gzeof() semantics over a memory image
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const unsigned char *buf;
  size_t n, pos;
  int past; /* zlib's state->past: a read found the input exhausted */
} memgz;

/* gzgets(fd, s, max) as used by GZ_READ, src/fastq.c:202-209: on NULL the buffer gets "" */
static void mg_line(memgz *g, char *s, long max) {
  long left = max - 1;
  char *w = s;
  while (left > 0) {
    if (g->pos >= g->n) {
      g->past = 1;
      break;
    }
    unsigned char c = g->buf[g->pos++];
    *w++ = (char)c;
    --left;
    if (c == '\n') break;
  }
  *w = '\0';
}

/* ------------------------------------------------------------------------------------------
 * record and file objects (src/fastq.h:97-131)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  char hdr1[FQG_MAX_LABEL_LENGTH];
  char hdr2[FQG_MAX_LABEL_LENGTH];
  char *seq, *qual; /* FQG_MAX_READ_LENGTH each */
  unsigned long read_len;
  long long offset;
} entry_t;

typedef struct {
  memgz gz;
  const char *filename;
  unsigned long cline;
  unsigned long max_rl, last_rl, min_rl, min_qual, max_qual, num_rds;
  unsigned long *rdlen_ctr;
  int is_pe, readname_format, is_casava_18, space;
} file_t;

static entry_t *entry_new(run_t *r) {
  entry_t *e = (entry_t *)own(r, calloc(1, sizeof(entry_t)));
  e->seq = (char *)own(r, malloc(FQG_MAX_READ_LENGTH));
  e->qual = (char *)own(r, malloc(FQG_MAX_READ_LENGTH));
  return e;
}
/* src/fastq.c:163-188 fastq_new (cline starts at 0: the struct comes from zero pages) */
static file_t *file_new(run_t *r, const unsigned char *buf, size_t n, const char *name) {
  file_t *f = (file_t *)own(r, calloc(1, sizeof(file_t)));
  f->gz.buf = buf;
  f->gz.n = n;
  f->filename = name;
  f->min_rl = FQG_MAX_READ_LENGTH;
  f->min_qual = FQG_MAX_PHRED_QUAL;
  f->readname_format = FQG_NAME_UNDEF;
  f->is_casava_18 = -1;
  f->space = FQG_SPACE_UNDEF;
  f->rdlen_ctr = (unsigned long *)own(r, calloc(FQG_MAX_READ_LENGTH, sizeof(unsigned long)));
  return f;
}

/* src/fastq.c:97-110 */
static void entry_stats(file_t *f, const entry_t *e) {
  unsigned long slen = e->read_len;
  if (slen < f->min_rl) f->min_rl = slen;
  if (slen > f->max_rl) f->max_rl = slen;
  ++f->num_rds;
  f->last_rl = slen;
  f->rdlen_ctr[slen]++;
}

/* src/fastq.c:245-261 */
static int read_entry(run_t *r, file_t *f, entry_t *e) {
  e->offset = (long long)f->gz.pos;
  if (f->gz.past) return 0;
  mg_line(&f->gz, e->hdr1, FQG_MAX_LABEL_LENGTH);
  if (e->hdr1[0] == '\0') return 0;
  mg_line(&f->gz, e->seq, FQG_MAX_READ_LENGTH);
  mg_line(&f->gz, e->hdr2, FQG_MAX_LABEL_LENGTH);
  mg_line(&f->gz, e->qual, FQG_MAX_READ_LENGTH);
  if (e->seq[0] == '\0' || e->hdr2[0] == '\0' || e->qual[0] == '\0') {
    ERR_OPEN(r);
    eprintf(r, "Error in file %s: line %lu: file truncated", f->filename, f->cline);
    ERR_CLOSE(r);
    note(r, FQG_E_TRUNCATED, f->cline / 4, f->cline, 0, 0);
    leave(r, 1);
  }
  f->cline += 4;
  e->read_len = strlen(e->seq);
  return 1;
}
/* src/fastq.c:237-243 */
static int read_next_entry(run_t *r, file_t *f, entry_t *e) {
  int k = read_entry(r, f, e);
  if (k <= 0) return k;
  entry_stats(f, e);
  return 1;
}

/* the four once-per-file probes, src/fastq.c:666-754 (same POSIX patterns) */
static int rx_match(const char *pat, int flags, const char *s) {
  regex_t rx;
  if (regcomp(&rx, pat, flags)) return 0;
  int hit = regexec(&rx, s, 0, NULL, 0) == 0;
  regfree(&rx);
  return hit;
}

/* src/fastq.c:442-516 */
static char *get_readname(run_t *r, file_t *f, entry_t *e, char *rn, unsigned long *len_p,
                          int is_header1) {
  const char *hdr = is_header1 ? e->hdr1 : e->hdr2;
  unsigned long len = 0;
  if (is_header1 && hdr[0] != '@') {
    ERR_OPEN(r);
    eprintf(r, "Error in file %s: line %lu: wrong header %s", f->filename, f->cline, hdr);
    ERR_CLOSE(r);
    note(r, FQG_E_WRONG_HEADER, f->cline / 4 - 1, f->cline, 0, 0);
    leave(r, 3);
  }
  /* one spare byte in front so that the reference's rn[len-1] / rn[len-2] accesses with
   * len<2 stay inside the buffer (they are out-of-bounds stack accesses there) */
  strncpy(rn, hdr + 1, FQG_MAX_LABEL_LENGTH - 1);
  if (f->readname_format == FQG_NAME_UNDEF) {
    f->is_casava_18 = rx_match("[A-Z0-9:]* [1234]:[YN]:[0-9]*.*", 0, rn);
    if (f->is_casava_18) {
      eprintf(r, "CASAVA=1.8\n");
      f->readname_format = FQG_NAME_CASAVA18;
    } else if (rx_match("^[0-9]+[\n\r]?$", REG_EXTENDED, rn)) {
      eprintf(r, "Read name provided as an integer\n");
      f->readname_format = FQG_NAME_INTEGER;
    } else if (!rx_match("[# \t/:][0-9abAB][\n\r]?$", REG_EXTENDED, rn)) {
      eprintf(r, "Read name provided with no suffix\n");
      f->readname_format = FQG_NAME_NOP;
    } else {
      f->readname_format = FQG_NAME_DEFAULT;
    }
  }
  if (f->space == FQG_SPACE_UNDEF) {
    f->space = rx_match("^[GT]?[0123n\\.NtT]+\n?$", REG_EXTENDED, e->seq) ? FQG_SPACE_COLOUR
                                                                         : FQG_SPACE_SEQ;
    if (f->space == FQG_SPACE_COLOUR) eprintf(r, "Color space\n");
  }
  switch (f->readname_format) {
    case FQG_NAME_DEFAULT:
      len = strlen(rn);
      if (f->is_pe) len--;
      rn[(long)len - 1] = '\0';
      break;
    case FQG_NAME_INTEGER:
      len = strlen(rn);
      rn[(long)len - 1] = '\0';
      break;
    case FQG_NAME_CASAVA18:
      len = 0;
      while (rn[len] != ' ' && rn[len] != '\0') ++len;
      rn[len] = '\0';
      if (rn[(long)len - 2] == '/') {
        rn[(long)len - 2] = '\0';
        len = len - 2;
      }
      break;
  }
  *len_p = len;
  return rn;
}

/* src/fastq.c:543-566 */
static int same_headers(const char *a, const char *b) {
  unsigned i = 0;
  if (b[0] == '\n' || b[0] == '\r' || b[0] == '\0') return 1;
  while (a[i] != '\0' && b[i] != '\0' && a[i] == b[i]) ++i;
  unsigned j = i;
  for (; a[i] != '\0'; ++i)
    if (a[i] != '\r' && a[i] != '\n') return 0;
  for (; b[j] != '\0'; ++j)
    if (b[j] != '\r' && b[j] != '\n') return 0;
  return 1;
}

static int is_base(char c) {
  return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'U' || c == 'a' || c == 'c' ||
         c == 'g' || c == 't' || c == 'u' || c == '0' || c == '1' || c == '2' || c == '3' ||
         c == 'n' || c == 'N' || c == '.';
}

/* src/fastq.c:300-392.  `rec` is only used for the structured outcome. */
static int validate_entry(run_t *r, file_t *f, entry_t *e, uint64_t rec) {
  /* two guard bytes in front of each name buffer, see get_readname() */
  char name1[FQG_MAX_LABEL_LENGTH + 8], name2[FQG_MAX_LABEL_LENGTH + 8];
  memset(name1, 0, 8);
  memset(name2, 0, 8);
  if (e->hdr1[0] != '@') {
    ERR_OPEN(r);
    eprintf(r, "Error in file %s: line %lu: sequence identifier should start with an @ - %s",
            f->filename, f->cline, e->hdr1);
    ERR_CLOSE(r);
    note(r, FQG_E_HDR1_AT, rec, f->cline, 0, 0);
    return 1;
  }
  if (e->hdr1[1] == '\0' || e->hdr1[1] == '\n' || e->hdr1[1] == '\r') {
    ERR_OPEN(r);
    eprintf(r, "Error in file %s: line %lu: sequence identifier should be longer than 1",
            f->filename, f->cline);
    ERR_CLOSE(r);
    note(r, FQG_E_HDR1_SHORT, rec, f->cline, 0, 0);
    return 1;
  }
  unsigned long slen = 0;
  int seen_t = 0, seen_u = 0;
  for (;; ++slen) {
    char c = e->seq[slen];
    if (c == '\0' || c == '\n' || c == '\r') break;
    if (!is_base(c)) {
      ERR_OPEN(r);
      eprintf(r,
              "Error in file %s: line %lu: invalid character '%c' (hex. code:'%x'), expected "
              "ACGTUacgtu0123nN.",
              f->filename, f->cline + 1, c, c);
      ERR_CLOSE(r);
      note(r, FQG_E_SEQ_CHAR, rec, f->cline + 1, (unsigned char)c, 0);
      return 1;
    }
    if (c == 'U' || c == 'u') seen_u = 1;
    else if (c == 'T' || c == 't') seen_t = 1;
    if (seen_u && seen_t) {
      ERR_OPEN(r);
      eprintf(r, "Error in file %s: line %lu: read contains both U and T bases", f->filename,
              f->cline - 2);
      ERR_CLOSE(r);
      note(r, FQG_E_SEQ_UT, rec, f->cline - 2, 0, 0);
      return 1;
    }
  }
  entry_stats(f, e);
  if (slen < 1) {
    ERR_OPEN(r);
    eprintf(r, "Error in file %s: line %lu: read length too small - %lu", f->filename,
            f->cline + 1, slen);
    ERR_CLOSE(r);
    note(r, FQG_E_LEN_SMALL, rec, f->cline + 1, slen, 0);
    return 1;
  }
  if (e->hdr2[0] != '+') {
    ERR_OPEN(r);
    eprintf(r,
            "Error in file %s: line %lu:  header2 wrong. The line should contain only '+' "
            "followed by a newline or read name (header1).",
            f->filename, f->cline + 2);
    ERR_CLOSE(r);
    note(r, FQG_E_HDR2_PLUS, rec, f->cline + 2, 0, 0);
    return 1;
  }
  unsigned long len;
  {
    char *rn1 = get_readname(r, f, e, name1 + 8, &len, 1);
    char *rn2 = get_readname(r, f, e, name2 + 8, &len, 0);
    if (!same_headers(rn1, rn2)) {
      ERR_OPEN(r);
      eprintf(r,
              "Error in file %s: line %lu:  header2 differs from header1\nheader 1 \"%s\"\nheader "
              "2 \"%s\"",
              f->filename, f->cline, e->hdr1, e->hdr2);
      ERR_CLOSE(r);
      note(r, FQG_E_HDR2_DIFF, rec, f->cline, 0, 0);
      return 1;
    }
  }
  unsigned long qlen = 0;
  for (;; ++qlen) {
    char c = e->qual[qlen];
    if (c == '\0' || c == '\n' || c == '\r') break;
    unsigned int x = (unsigned int)c; /* sign-extends bytes >= 0x80, as the reference does */
    if (x < f->min_qual) f->min_qual = x;
    if (x > f->max_qual) f->max_qual = x;
  }
  if (f->space == FQG_SPACE_SEQ && qlen != slen) {
    ERR_OPEN(r);
    eprintf(r,
            "Error in file %s: line %lu: sequence and quality don't have the same length %lu!=%lu",
            f->filename, f->cline, slen, qlen);
    ERR_CLOSE(r);
    note(r, FQG_E_QLEN, rec, f->cline, slen, qlen);
    return 1;
  }
  if (f->space == FQG_SPACE_COLOUR && (qlen == slen - 1 || qlen == slen)) return 0;
  if (f->space == FQG_SPACE_COLOUR) {
    ERR_OPEN(r);
    eprintf(r, "Error in file %s: line %lu: sequence and quality length don't match %lu!=%lu",
            f->filename, f->cline, slen, qlen);
    ERR_CLOSE(r);
    note(r, FQG_E_QLEN_CS, rec, f->cline, slen, qlen);
    return 1;
  }
  return 0;
}

/* src/fastq.h:82 PRINT_READS_PROCESSED */
static void progress(run_t *r, unsigned long c, unsigned long every) {
  if (c % every == 0) eprintf(r, "\b\b\b\b\b\b\b\b\b\b\b\b\b\b\b%lu", c);
}

/* ------------------------------------------------------------------------------------------
 * exact name set.  The reference keeps names in hash.c (src/hash.c:38-184) keyed by an sdbm
 * hash and confirms with strcmp (src/fastq.c:577-587); observable behaviour is that of a set
 * of strings with insert / member / delete, which is all this restates.
 * ---------------------------------------------------------------------------------------- */
typedef struct nameset_s {
  char **slot;
  size_t cap, live, used;
} nameset;
#define TOMB ((char *)1)
static uint64_t fnv(const char *s) {
  uint64_t h = 1469598103934665603ull;
  for (; *s; ++s) h = (h ^ (unsigned char)*s) * 1099511628211ull;
  return h;
}
static void ns_init(nameset *t, size_t cap) {
  t->cap = cap;
  t->live = t->used = 0;
  t->slot = (char **)calloc(cap, sizeof(char *));
}
static void ns_free(nameset *t) {
  for (size_t i = 0; i < t->cap; ++i)
    if (t->slot[i] && t->slot[i] != TOMB) free(t->slot[i]);
  free(t->slot);
}
static void ns_put_raw(nameset *t, char *s) {
  size_t i = fnv(s) & (t->cap - 1);
  while (t->slot[i] && t->slot[i] != TOMB) i = (i + 1) & (t->cap - 1);
  if (!t->slot[i]) t->used++;
  t->slot[i] = s;
  t->live++;
}
static void ns_grow(nameset *t) {
  nameset big;
  ns_init(&big, t->cap * 2);
  for (size_t i = 0; i < t->cap; ++i)
    if (t->slot[i] && t->slot[i] != TOMB) ns_put_raw(&big, t->slot[i]);
  free(t->slot);
  *t = big;
}
static long ns_find(const nameset *t, const char *s) {
  size_t i = fnv(s) & (t->cap - 1);
  while (t->slot[i]) {
    if (t->slot[i] != TOMB && !strcmp(t->slot[i], s)) return (long)i;
    i = (i + 1) & (t->cap - 1);
  }
  return -1;
}
static void ns_add(nameset *t, const char *s) {
  if ((t->used + 1) * 2 > t->cap) ns_grow(t);
  ns_put_raw(t, strdup(s));
}
static void ns_del(nameset *t, long i) {
  free(t->slot[i]);
  t->slot[i] = TOMB;
  t->live--;
}

/* src/fastq.c:396-439 fastq_index_readnames, with new_indexentry's accounting (:590-611):
 * sizeof(INDEX_ENTRY)=16, sizeof(hashnode)=24 on LP64 */
static void index_readnames(run_t *r, file_t *f, nameset *idx) {
  entry_t *e = entry_new(r);
  char namebuf[FQG_MAX_LABEL_LENGTH + 8];
  memset(namebuf, 0, 8);
  unsigned long len;
  while (!f->gz.past) {
    if (read_next_entry(r, f, e) == 0) break;
    uint64_t rec = f->cline / 4 - 1;
    char *name = get_readname(r, f, e, namebuf + 8, &len, 1);
    if (ns_find(idx, name) >= 0) {
      ERR_OPEN(r);
      eprintf(r, "Error in file %s: line %lu: duplicated sequence %s", f->filename, f->cline,
              name);
      ERR_CLOSE(r);
      note(r, FQG_E_DUP_NAME, rec, f->cline, 0, 0);
      leave(r, 3);
    }
    {
      /* new_indexentry copies `len` bytes and terminates: the stored name is name[0..len) cut
       * at its first NUL */
      char stored[FQG_MAX_LABEL_LENGTH + 1];
      strncpy(stored, name, len);
      stored[len] = '\0';
      ns_add(idx, stored);
      r->index_mem += 16 + len + 1 + 24;
    }
    if (validate_entry(r, f, e, rec) != 0) {
      leave(r, 3);
    }
    progress(r, f->cline / 4, 100000);
  }
}

/* src/fastq_info.c:39-55 */
static unsigned int median_rl(const file_t *f1, const file_t *f2) {
  unsigned long long ctr = 0;
  unsigned int crl = 1;
  unsigned long nreads = f1->num_rds;
  if (f1->num_rds == 1 && f2 == NULL) return (unsigned int)f1->min_rl;
  if (f2 != NULL) nreads += f2->num_rds;
  while (crl < FQG_MAX_READ_LENGTH) {
    ctr += f1->rdlen_ctr[crl];
    if (f2 != NULL) ctr += f2->rdlen_ctr[crl];
    if (f1->num_rds > 1 && ctr > nreads / 2) break;
    ++crl;
  }
  return crl;
}

/* src/fastq.c:274-297 */
const char *fqo_qual_range_to_enc(unsigned int min_qual, unsigned int max_qual) {
  static const char *names[] = {"33", "64", "solexa", "33 *", "sanger"};
  int enc;
  if (min_qual >= 33 && min_qual < 59 && max_qual >= 90) enc = 4;
  else if (min_qual >= 33 && max_qual <= 73) enc = 0;
  else if (min_qual < 59) enc = 0;
  else if (min_qual >= 64 && max_qual > 74) enc = 1;
  else if (min_qual >= 59 && max_qual > 74) enc = 2;
  else enc = 3;
  if (max_qual > FQG_MAX_PHRED_QUAL) return NULL;
  if (enc != 4 && max_qual > min_qual + 60) return NULL;
  return names[enc];
}

/* src/fastq_info.c:57-106 */
static file_t *run_interleaved(run_t *r, const unsigned char *b, size_t n, const char *name) {
  eprintf(r, "Paired-end interleaved\n");
  file_t *f = file_new(r, b, n, name);
  f->is_pe = 1;
  entry_t *m1 = entry_new(r), *m2 = entry_new(r);
  char nb1[FQG_MAX_LABEL_LENGTH + 8], nb2[FQG_MAX_LABEL_LENGTH + 8];
  memset(nb1, 0, 8);
  memset(nb2, 0, 8);
  unsigned long len = 0;
  while (!f->gz.past) {
    if (read_entry(r, f, m1) == 0) break;
    if (read_entry(r, f, m2) == 0) {
      ERR_OPEN(r);
      eprintf(r, "Error in file %s: line %lu: file truncated?", name, f->cline);
      ERR_CLOSE(r);
      note(r, FQG_E_TRUNCATED, f->cline / 4, f->cline, 1, 0);
      leave(r, 3);
    }
    uint64_t rec2 = f->cline / 4 - 1;
    char *n1 = get_readname(r, f, m1, nb1 + 8, &len, 1);
    char *n2 = get_readname(r, f, m2, nb2 + 8, &len, 1);
    if (strcmp(n1, n2)) {
      ERR_OPEN(r);
      eprintf(r, "Error in file %s: line %lu: unpaired read - %s", name, f->cline, n1);
      ERR_CLOSE(r);
      note(r, FQG_E_UNPAIRED, rec2 - 1, f->cline, 0, 0);
      leave(r, 3);
    }
    if (validate_entry(r, f, m1, rec2 - 1)) leave(r, 3);
    if (validate_entry(r, f, m2, rec2)) leave(r, 3);
    progress(r, f->cline / 4, 100000);
  }
  oprintf(r, "\n");
  return f;
}

/* src/fastq_info.c:108-152 */
static file_t *run_paired_sorted(run_t *r, const unsigned char *b1, size_t n1, const char *name1,
                                 const unsigned char *b2, size_t n2, const char *name2,
                                 file_t **f2_out) {
  file_t *f1 = file_new(r, b1, n1, name1), *f2 = file_new(r, b2, n2, name2);
  *f2_out = f2;
  f1->is_pe = f2->is_pe = 1;
  entry_t *m1 = entry_new(r), *m2 = entry_new(r);
  char nb1[FQG_MAX_LABEL_LENGTH + 8], nb2[FQG_MAX_LABEL_LENGTH + 8];
  memset(nb1, 0, 8);
  memset(nb2, 0, 8);
  unsigned long l1, l2;
  while (!f1->gz.past) {
    if (read_entry(r, f1, m1) == 0) break;
    if (validate_entry(r, f1, m1, f1->cline / 4 - 1)) leave(r, 3);
    if (read_entry(r, f2, m2) == 0) break;
    if (validate_entry(r, f2, m2, f2->cline / 4 - 1)) {
      r->first.file = 2;
      leave(r, 3);
    }
    get_readname(r, f1, m1, nb1 + 8, &l1, 1);
    get_readname(r, f2, m2, nb2 + 8, &l2, 1);
    if (strcmp(nb1 + 8, nb2 + 8)) {
      ERR_OPEN(r);
      eprintf(r, "Readnames do not match across files (read #%ld)", (long)(f1->cline / 4 + 1));
      ERR_CLOSE(r);
      note(r, FQG_E_NAME_MISMATCH, f1->cline / 4 - 1, f1->cline, 0, 0);
      leave(r, 3);
    }
    progress(r, f1->cline / 2, 100000);
  }
  if (read_entry(r, f1, m1) != 0) {
    ERR_OPEN(r);
    eprintf(r, "Premature end of file2");
    ERR_CLOSE(r);
    leave(r, 3);
  }
  if (read_entry(r, f2, m2) != 0) {
    ERR_OPEN(r);
    eprintf(r, "Premature end of file1");
    ERR_CLOSE(r);
    leave(r, 3);
  }
  oprintf(r, "\n");
  return f1;
}

/* src/fastq_info.c:155-176 */
static file_t *run_single_noindex(run_t *r, const unsigned char *b, size_t n, const char *name) {
  file_t *f = file_new(r, b, n, name);
  f->is_pe = 1;
  entry_t *m = entry_new(r);
  while (!f->gz.past) {
    if (read_entry(r, f, m) == 0) break;
    if (validate_entry(r, f, m, f->cline / 4 - 1)) leave(r, 3);
    progress(r, f->cline / 4, 100000);
  }
  oprintf(r, "\n");
  return f;
}

/* src/fastq_info.c:190-396 main(), after option parsing */
static void run_fastq_info(run_t *r, const fqo_job *job, fqo_summary *sum) {
  const int paired = job->arg2_kind != FQO_ARG2_NONE;
  const int interleaved = job->arg2_kind == FQO_ARG2_PE;
  const int sorted = (job->flags & FQO_FLAG_S) != 0, empty_ok = (job->flags & FQO_FLAG_E) != 0,
            noenc_ok = (job->flags & FQO_FLAG_Q) != 0, skip_names = (job->flags & FQO_FLAG_R) != 0;
  unsigned long num_reads1 = 0, num_reads2 = 0;
  file_t *f1 = NULL, *f2 = NULL;
  nameset idx;

  eprintf(r, "fastq_utils %s\n", "0.25.3");
  if (interleaved) {
    f1 = run_interleaved(r, job->buf1, job->n1, job->name1);
    num_reads1 = f1->num_rds;
  } else if (paired && sorted && skip_names) {
    eprintf(r, "-s option used: assuming that reads have the same ordering in both files\n");
    file_t *tmp = NULL;
    f1 = run_paired_sorted(r, job->buf1, job->n1, job->name1, job->buf2, job->n2, job->name2, &tmp);
    num_reads1 = f1->num_rds;
  } else if (!paired && skip_names) {
    eprintf(r, "Skipping check for duplicated read names\n");
    f1 = run_single_noindex(r, job->buf1, job->n1, job->name1);
    num_reads1 = f1->num_rds;
  } else {
    f1 = file_new(r, job->buf1, job->n1, job->name1);
    if (paired) f1->is_pe = 1;
    eprintf(r, "DEFAULT_HASHSIZE=%lu\n", 39000001ul);
    ns_init(&idx, 1024);
    r->idx = &idx;
    r->index_mem += 8; /* sizeof(hashtable): a pointer */
    eprintf(r, "Scanning and indexing all reads from %s\n", f1->filename);
    index_readnames(r, f1, &idx);
    eprintf(r, "Scanning complete.\n");
    num_reads1 = idx.live;
    eprintf(r, "\n");
    eprintf(r, "Reads processed: %llu\n", (unsigned long long)idx.live);
    eprintf(r, "Memory used in indexing: ~%ld MB\n", (long)(r->index_mem / 1024 / 1024));
  }
  if (num_reads1 == 0) {
    if (empty_ok) {
      oprintf(r, "Number of reads: %lu\n", 0L);
      oprintf(r, "Quality encoding range: %lu %lu\n", 0L, 0L);
      oprintf(r, "Quality encoding: %s\n", "");
      oprintf(r, "Read length: %lu %lu %u\n", 0L, 0L, 0);
      leave(r, 0);
    }
    ERR_OPEN(r);
    eprintf(r, "No reads found in %s.", job->name1);
    ERR_CLOSE(r);
    leave(r, 3);
  }
  unsigned long min_rl = f1->min_rl, max_rl = f1->max_rl, min_qual = f1->min_qual,
                max_qual = f1->max_qual;
  if (paired && !interleaved && !sorted) {
    eprintf(r, "File %s processed\n", job->name1);
    eprintf(r, "Next file %s\n", job->name2);
    f2 = file_new(r, job->buf2, job->n2, job->name2);
    f2->is_pe = 1;
    entry_t *m2 = entry_new(r);
    char nb[FQG_MAX_LABEL_LENGTH + 8];
    memset(nb, 0, 8);
    unsigned long len;
    while (!f2->gz.past) {
      if (read_entry(r, f2, m2) == 0) break;
      r->first.file = 2;
      char *name = get_readname(r, f2, m2, nb + 8, &len, 1);
      long at = ns_find(&idx, name);
      if (at < 0) {
        ERR_OPEN(r);
        eprintf(r, "Error in file %s: line %lu: unpaired read - %s", job->name2, f2->cline, name);
        ERR_CLOSE(r);
        note(r, FQG_E_UNPAIRED, f2->cline / 4 - 1, f2->cline, 0, 0);
        leave(r, 3);
      }
      ns_del(&idx, at);
      /* file-2 records are validated against file 1's state (src/fastq_info.c:345) */
      if (validate_entry(r, f1, m2, f2->cline / 4 - 1)) leave(r, 3);
      progress(r, f2->cline / 4, 100000);
    }
    r->first.file = 0;
    oprintf(r, "\n");
    if (idx.live > 0) {
      ERR_OPEN(r);
      eprintf(r, "Error in file %s: found %llu unpaired reads", job->name1,
              (unsigned long long)idx.live);
      ERR_CLOSE(r);
      leave(r, 3);
    }
    if (f2->min_rl < min_rl) min_rl = f2->min_rl;
    if (f2->max_rl > max_rl) max_rl = f2->max_rl;
    if (f2->min_qual < min_qual) min_qual = f2->min_qual;
    if (f2->max_qual > max_qual) max_qual = f2->max_qual;
  }
  /* structured copy of the statistics, taken before the encoding verdict can end the run */
  unsigned int med = median_rl(f1, f2);
  if (sum) {
    sum->num_reads = num_reads1;
    sum->min_rl = min_rl;
    sum->max_rl = max_rl;
    sum->median_rl = med;
    sum->min_qual = min_qual;
    sum->max_qual = max_qual;
    sum->num_rds_counted = f1->num_rds;
  }
  eprintf(r, "------------------------------------\n");
  if (num_reads2 > 0) eprintf(r, "Number of reads: %lu %lu\n", num_reads1, num_reads2);
  else eprintf(r, "Number of reads: %lu\n", num_reads1);
  const char *enc = fqo_qual_range_to_enc((unsigned int)min_qual, (unsigned int)max_qual);
  if (enc == NULL && !noenc_ok) {
    ERR_OPEN(r);
    if (max_qual > FQG_MAX_PHRED_QUAL)
      eprintf(r, "Unable to determine quality encoding - unknown range [%lu,>%u]", min_qual,
              FQG_MAX_PHRED_QUAL);
    else
      eprintf(r, "Unable to determine quality encoding - unknown range [%lu,%lu]", min_qual,
              max_qual);
    ERR_CLOSE(r);
    leave(r, 3);
  }
  eprintf(r, "Quality encoding range: %lu %lu\n", min_qual, max_qual);
  if (enc == NULL) eprintf(r, "Quality encoding: NA\n");
  else eprintf(r, "Quality encoding: %s\n", enc);
  eprintf(r, "Read length: %lu %lu %u\n", min_rl - 1, max_rl - 1, med - 1);
  eprintf(r, "OK\n");
  leave(r, 0);
}

/* ------------------------------------------------------------------------------------------
 * fastq_filterpair (src/fastq_filterpair.c:38-228)
 * ---------------------------------------------------------------------------------------- */
/* name -> entry_start (INDEX_ENTRY, src/fastq.h:91-95): the index of fastq_index_readnames with the offsets the
 * pairing loop hands to fastq_quick_copy_entry */
typedef struct {
  char **name;
  long long *start;
  size_t cap, live;
} namemap;
static void nm_init(namemap *m, size_t cap) {
  m->cap = cap;
  m->live = 0;
  m->name = (char **)calloc(cap, sizeof(char *));
  m->start = (long long *)calloc(cap, sizeof(long long));
}
static void nm_free(namemap *m) {
  for (size_t i = 0; i < m->cap; ++i)
    if (m->name[i] && m->name[i] != TOMB) free(m->name[i]);
  free(m->name);
  free(m->start);
}
static long nm_find(const namemap *m, const char *s) {
  size_t i = fnv(s) & (m->cap - 1);
  while (m->name[i]) {
    if (m->name[i] != TOMB && !strcmp(m->name[i], s)) return (long)i;
    i = (i + 1) & (m->cap - 1);
  }
  return -1;
}
static void nm_add(namemap *m, const char *s, long long start) {
  size_t i = fnv(s) & (m->cap - 1);
  while (m->name[i]) i = (i + 1) & (m->cap - 1);
  m->name[i] = strdup(s);
  m->start[i] = start;
  m->live++;
}
static void nm_del(namemap *m, long i) {
  free(m->name[i]);
  m->name[i] = TOMB;
  m->live--;
}
static size_t count_newlines(const unsigned char *b, size_t n) {
  size_t c = 0;
  for (size_t i = 0; i < n; ++i) c += b[i] == '\n';
  return c;
}
/* fastq_index_readnames (src/fastq.c:396-439) into a namemap */
static void index_readnames_map(run_t *r, file_t *f, namemap *idx) {
  entry_t *e = entry_new(r);
  char namebuf[FQG_MAX_LABEL_LENGTH + 8];
  unsigned long len;
  while (!f->gz.past) {
    if (read_next_entry(r, f, e) == 0) break;
    uint64_t rec = f->cline / 4 - 1;
    char *name = get_readname(r, f, e, namebuf + 8, &len, 1);
    if (nm_find(idx, name) >= 0) {
      ERR_OPEN(r);
      eprintf(r, "Error in file %s: line %lu: duplicated sequence %s", f->filename, f->cline, name);
      ERR_CLOSE(r);
      note(r, FQG_E_DUP_NAME, rec, f->cline, 0, 0);
      leave(r, 3);
    }
    char stored[FQG_MAX_LABEL_LENGTH + 1];
    strncpy(stored, name, len);
    stored[len] = '\0';
    nm_add(idx, stored, e->offset);
    r->index_mem += 16 + len + 1 + 24;
    if (validate_entry(r, f, e, rec) != 0) leave(r, 3);
    progress(r, f->cline / 4, 100000);
  }
}
static void put_entry(sink_t *o, const entry_t *e) { /* fastq_write_entry, src/fastq.c:265-272 */
  const char *l[4] = {e->hdr1, e->seq, e->hdr2, e->qual};
  for (int i = 0; i < 4; ++i) {
    size_t n = strlen(l[i]);
    if (o->n + n + 1 > o->cap) {
      size_t cap = o->cap ? o->cap * 2 : 65536;
      while (cap < o->n + n + 1) cap *= 2;
      o->p = (char *)realloc(o->p, cap);
      o->cap = cap;
    }
    memcpy(o->p + o->n, l[i], n);
    o->n += n;
  }
}
static void mg_rewind(file_t *f) { /* fastq_rewind, src/fastq.c:77-80 */
  f->cline = 1;
  f->gz.pos = 0;
  f->gz.past = 0;
}
static void run_filterpair(run_t *r, const fqo_job *job, sink_t out[3]) {
  int argc = 6 + ((job->flags & FQO_FLAG_S) ? 1 : 0);
  eprintf(r, "%d", argc);
  file_t *fd1 = file_new(r, job->buf1, job->n1, job->name1);
  fd1->is_pe = 1;
  file_t *fd2 = file_new(r, job->buf2, job->n2, job->name2);
  fd2->is_pe = 1;
  int sorted = (job->flags & FQO_FLAG_S) != 0;
  eprintf(r, "HASHSIZE=%u\n", 100000001u);
  if (sorted) eprintf(r, "Assuming sorted fastq files\n");
  namemap *idx = (namemap *)own(r, calloc(1, sizeof(namemap))), *idx2 = NULL;
  size_t cap = 1024;
  while (cap < count_newlines(job->buf1, job->n1)) cap <<= 1;
  nm_init(idx, cap);
  r->map[0] = idx;
  r->index_mem += 8;
  eprintf(r, "Scanning and indexing all reads from %s\n", fd1->filename);
  index_readnames_map(r, fd1, idx);
  eprintf(r, "Scanning complete.\n");
  eprintf(r, "Reads indexed: %llu\n", (unsigned long long)idx->live);
  eprintf(r, "Memory used in indexing: %ld MB\n", (long)(r->index_mem / 1024 / 1024));
  unsigned long up2 = 0, paired = 0;
  entry_t *m1 = entry_new(r), *m2 = entry_new(r), *tmp = entry_new(r);
  char namebuf[FQG_MAX_LABEL_LENGTH + 8];
  unsigned long len;
  if (sorted) {
    idx2 = (namemap *)own(r, calloc(1, sizeof(namemap)));
    cap = 1024;
    while (cap < count_newlines(job->buf2, job->n2)) cap <<= 1;
    nm_init(idx2, cap);
    r->map[1] = idx2;
    r->index_mem += 8;
    eprintf(r, "Scanning and indexing all reads from %s\n", fd2->filename);
    index_readnames_map(r, fd2, idx2);
    eprintf(r, "Scanning complete.\n");
    eprintf(r, "Reads indexed: %llu\n", (unsigned long long)idx2->live);
    eprintf(r, "Memory used in indexing: %ld MB\n", (long)(r->index_mem / 1024 / 1024));
    mg_rewind(fd1);
    mg_rewind(fd2);
    file_t *fs[2] = {fd1, fd2};
    namemap *other[2] = {idx2, idx};
    for (int s = 0; s < 2; ++s) {
      file_t *f = fs[s];
      eprintf(r, "Filtering %s...\n", f->filename);
      while (!f->gz.past) {
        if (read_next_entry(r, f, m2) == 0) break;
        char *name = get_readname(r, f, m2, namebuf + 8, &len, 1);
        long at = nm_find(other[s], name);
        if (at < 0) {
          ++up2;
          put_entry(&out[2], m2);
        } else {
          if (s == 0) ++paired;
          put_entry(&out[s], m2);
          nm_del(other[s], at);
        }
        progress(r, f->cline / 4, 10000);
      }
    }
  } else {
    mg_rewind(fd1);
    eprintf(r, "Processing %s\n", fd2->filename);
    unsigned long ctr_seek = 0, ctr_noseek = 0;
    while (!fd2->gz.past) {
      if (read_next_entry(r, fd2, m2) == 0) break;
      char *name = get_readname(r, fd2, m2, namebuf + 8, &len, 1);
      long at = nm_find(idx, name);
      if (at < 0) {
        ++up2;
        put_entry(&out[2], m2);
      } else {
        ++paired;
        put_entry(&out[1], m2);
        /* fastq_quick_copy_entry, src/fastq.c:125-157 */
        long long off = idx->start[at];
        if ((long long)fd1->gz.pos != off) {
          fd1->gz.pos = (size_t)off;
          fd1->gz.past = 0;
          ++ctr_seek;
        } else ++ctr_noseek;
        eprintf(r, "%lu / %lu\n", ctr_seek, ctr_noseek);
        if (fd1->gz.past) {
          ERR_OPEN(r);
          eprintf(r, "Error in file %s: line %lu: premature eof", fd1->filename, fd1->cline);
          ERR_CLOSE(r);
          leave(r, 3);
        }
        mg_line(&fd1->gz, tmp->hdr1, FQG_MAX_LABEL_LENGTH);
        int bad = tmp->hdr1[0] == '\0';
        if (!bad) {
          mg_line(&fd1->gz, tmp->seq, FQG_MAX_READ_LENGTH);
          mg_line(&fd1->gz, tmp->hdr2, FQG_MAX_LABEL_LENGTH);
          mg_line(&fd1->gz, tmp->qual, FQG_MAX_READ_LENGTH);
          bad = tmp->seq[0] == '\0' || tmp->hdr2[0] == '\0' || tmp->qual[0] == '\0';
        }
        if (bad) {
          ERR_OPEN(r);
          eprintf(r, "Error in file %s: line %lu: file truncated", fd1->filename, fd1->cline);
          ERR_CLOSE(r);
          leave(r, 3);
        }
        put_entry(&out[0], tmp);
        nm_del(idx, at);
      }
      progress(r, fd2->cline / 4, 10000);
    }
    eprintf(r, "\n");
    eprintf(r, "Recording %llu unpaired reads from %s\n", (unsigned long long)idx->live, job->name1);
    unsigned long remaining = idx->live;
    while (!fd1->gz.past && remaining) {
      if (read_next_entry(r, fd1, m1) == 0) break;
      char *name = get_readname(r, fd1, m1, namebuf + 8, &len, 1);
      if (nm_find(idx, name) >= 0) {
        put_entry(&out[2], m1);
        remaining--;
      }
      progress(r, fd1->cline / 4, 100000);
    }
    eprintf(r, "Unpaired from %s: %llu\n", job->name1, (unsigned long long)idx->live);
    eprintf(r, "Unpaired from %s: %ld\n", job->name2, (long)up2);
  }
  eprintf(r, "\n");
  eprintf(r, "Paired: %ld\n", (long)paired);
  if (paired == 0) {
    eprintf(r, "!!!WARNING!!! 0 paired reads! are the headers ok?\n");
    leave(r, 3);
  }
  leave(r, 0);
}

/* job: buf1/buf2 = the two inputs (decompressed), flags & FQO_FLAG_S = the 7th argument "sorted".  The three
 * outputs (paired1, paired2, unpaired: what the reference gzips) come back uncompressed in out[0..2]. */
int fqo_fastq_filterpair(const fqo_job *job, fqo_result *res, char *out[3], size_t out_len[3]) {
  run_t *r = (run_t *)calloc(1, sizeof(run_t));
  sink_t *o = r->files;
  memset(res, 0, sizeof(*res));
  eprintf(r, "fastq_utils %s\n", "0.25.3");
  if (setjmp(r->bail) == 0) run_filterpair(r, job, r->files);
  o = r->files;
  for (int k = 0; k < 2; ++k)
    if (r->map[k]) nm_free((namemap *)r->map[k]);
  for (int i = 0; i < r->n_owned; ++i) free(r->owned[i]);
  res->exit_status = r->status;
  res->first = r->first;
  res->out = r->out.p ? r->out.p : strdup("");
  res->out_len = r->out.n;
  res->err = r->err.p ? r->err.p : strdup("");
  res->err_len = r->err.n;
  for (int k = 0; k < 3; ++k) {
    out[k] = o[k].p ? o[k].p : strdup("");
    out_len[k] = o[k].n;
  }
  free(r);
  return res->exit_status;
}

int fqo_fastq_info(const fqo_job *job, fqo_result *res) {
  run_t *r = (run_t *)calloc(1, sizeof(run_t));
  memset(res, 0, sizeof(*res));
  if (setjmp(r->bail) == 0) run_fastq_info(r, job, &res->summary);
  if (r->idx) ns_free(r->idx);
  for (int i = 0; i < r->n_owned; ++i) free(r->owned[i]);
  res->exit_status = r->status;
  res->first = r->first;
  res->out = r->out.p ? r->out.p : strdup("");
  res->out_len = r->out.n;
  res->err = r->err.p ? r->err.p : strdup("");
  res->err_len = r->err.n;
  free(r);
  return res->exit_status;
}

void fqo_result_free(fqo_result *res) {
  free(res->out);
  free(res->err);
  res->out = res->err = NULL;
}
