// fqg_barcode_kernels.hip - the per-read transform of fastq_pre_barcodes on the GPU
// (reference src/fastq_pre_barcodes.c:594-727): lock-step records of up to five files, read
// names must agree, UMI / cell / sample substrings are cut out (with an optional minimum base
// quality), the header gets the STAGS_..._ETAGS_ prefix, reads are optionally sliced, and the
// result is emitted as FASTQ text (one or two outputs) or as SAM lines.
//
// Three launches per batch of iterations; plan and emit work on TILES of T consecutive iterations,
// one wavefront per tile, whose records (contiguous in every input image) are first copied into LDS
// with 16-byte loads:
//   k_bc_plan_tile  one lane per iteration: status (keep / discarded) and the exact number of output bytes of
//                   every output - from the line index and the bytes of the barcode-carrying files alone
//   scan            64-bit exclusive prefix of the byte counts -> where each iteration writes
//   k_bc_emit_tile  the name checks (on the headers of every file, which it stages anyway), then one lane per
//                   output line (SAM) or record (FASTQ): the lane writes its text into the tile's LDS output
//                   area, which then goes to the output image with 16-byte stores; three tiles under way
// T is chosen by the host from the mean record sizes so that a tile fits its LDS areas; a tile that
// does not fit (long reads) is flagged by the plan and handled by the slower direct paths (records read
// from the images, one wavefront per iteration in k_bc_emit_direct).
// Inputs are framed images (line index from fqg_validate); nothing is re-parsed on the host.
#include "fqg_device.h"

namespace fqg {

constexpr int kBcFiles = 6;  // index 1..5 as in the reference (READ1, READ2, INDEX1..3)
enum : uint8_t { kBcKeep = 0, kBcDiscardShort = 1, kBcDiscardQual = 2, kBcFinding = 3 };

// tile geometry of one call (host: bc_tile_for)
struct BcTile {
  uint32_t T;        // iterations per tile (<= 64 / lines per iteration)
  uint32_t in_cap;   // LDS bytes for the staged records
  uint32_t out_cap;  // LDS bytes for the output text (emit only)
  uint32_t plan_m;   // the plan kernel works on plan_m tiles at once (one lane per iteration: T * plan_m <= 64)
  uint32_t plan_cap; // LDS bytes for what the plan stages of T * plan_m iterations (bc_staged<PLAN>)
};

struct BcFile {
  FrameView fv;
  uint64_t first;  // record of iteration 0
  uint32_t step;   // records per iteration (2 for the two interleaved references)
  uint32_t add;    // 1 for the second interleaved reference
  int32_t present;
  int32_t fmt;     // read-name format of this file
  int32_t has_nul; // the image holds NUL bytes: lines are C strings (bc_clip_nul)
  int32_t pad_;
};

struct BcParams {
  BcFile f[kBcFiles];
  int32_t n_inputs;
  int32_t umi_read, cell_read, sample_read;
  int32_t phred, min_qual;
  int32_t out_sam, tenx;
  int32_t has_nul;  // some input's image holds NUL bytes: every tile takes the record-by-record kernels (bc_clip_nul)
  int32_t ablate;  // measurement only (FQGPU_BC_ABL, SAM emit): 1 = no line is written, 2 = no flush, 4 = no name check, 8 = no landing of the spans
  int32_t emit[3];
  int64_t umi_off, umi_size, cell_off, cell_size, sample_off, sample_size;
  int64_t read_off[3], read_size[3];
  uint64_t first_read_number;  // processed_reads before this batch
};

struct BcCall {
  unsigned long long first_finding;  // min over iterations of (iteration << 8 | code << 3 | file)
  unsigned long long first_discard;  // first discarded iteration (interleaved input re-syncs there)
  unsigned long long discarded, short_warnings;
  unsigned long long big;  // tiles that do not fit LDS
};

struct BcLine {
  const uint8_t* p;
  uint32_t len;  // bytes before '\n'
  uint32_t nl;
};

// Which input files exist.  MASK = 0: ask the parameters at run time; otherwise bit x of MASK says
// it at compile time, and the code and registers of absent files disappear from the kernel.
template <int MASK>
__device__ __forceinline__ bool bc_has(const BcParams& P, int x) {
  return MASK ? ((MASK >> x) & 1) != 0 : P.f[x].present != 0;
}

// A line as the reference holds it: gzgets puts it into a buffer and everything after that is a C string function
// (strlen, gzputs, printf("%s"), strncpy, the scans) - the line ENDS at its first NUL byte, and a line that ends there
// has no '\n'.  This is also what a line cut at the gzgets limits is (host/fq_reframe.h: the piece is followed by
// "\0\n").  Only images that hold a NUL byte come here (BcFile::has_nul); the line is in global memory.
__device__ __forceinline__ void bc_clip_nul(BcLine& l) {
  const uint32_t n = l.len;
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    __builtin_memcpy(&w, l.p + i, 8);
    const uint64_t z = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;  // 0x80 in the lowest zero byte (and maybe above it)
    if (z) {
      l.len = i + ((uint32_t)__builtin_ctzll(z) >> 3);
      l.nl = 0;
      return;
    }
  }
  for (; i < n; ++i)
    if (l.p[i] == 0) {
      l.len = i;
      l.nl = 0;
      return;
    }
}

__device__ __forceinline__ void bc_lines(const BcFile& f, uint64_t k, BcLine ln[4]) {
  const uint64_t r = f.first + k * f.step + f.add;
  uint64_t prev = r == 0 ? ~0ull : f.fv.line_end[4 * r - 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t e = f.fv.line_end[4 * r + i];
    ln[i].p = f.fv.img + prev + 1;
    ln[i].len = (uint32_t)(e - prev - 1);
    ln[i].nl = e < f.fv.nbytes ? 1u : 0u;
    prev = e;
  }
  if (f.has_nul) {  // (four calls, not a loop over ln[i]: an index that is no constant puts the lines in scratch memory)
    bc_clip_nul(ln[0]);
    bc_clip_nul(ln[1]);
    bc_clip_nul(ln[2]);
    bc_clip_nul(ln[3]);
  }
}

// 8 bytes at any alignment (LDS: one ds_read_b64; gfx950 has unaligned DS access)
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
// 0x80 in every byte of x that equals c
__device__ __forceinline__ uint64_t bytes_eq(uint64_t x, uint8_t c) {
  const uint64_t y = x ^ (0x0101010101010101ull * c);
  const uint64_t t = ((y & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | y;
  return ~(t | 0x7F7F7F7F7F7F7F7Full);
}

// canonical read name with is_pe set (every input of fastq_pre_barcodes has it, :570).
// WIDE: the line is in LDS with slack behind it - scan 8 bytes per step.
template <bool WIDE>
__device__ __forceinline__ uint32_t bc_name_len(const BcLine& h, int fmt) {
  const uint32_t L = h.len + h.nl - 1;  // strlen(&hdr[1]); images with NUL bytes are refused earlier
  if (fmt == FQG_NAME_CASAVA18) {
    uint32_t sp = L;
    if (WIDE) {
      for (uint32_t i = 0; i < L; i += 8) {
        const uint64_t m = bytes_eq(ld8(h.p + 1 + i), (uint8_t)' ');
        if (m) {
          const uint32_t at = i + ((uint32_t)__builtin_ctzll(m) >> 3);
          if (at < L) sp = at;
          break;
        }
      }
    } else {
      for (uint32_t i = 0; i < L; ++i)
        if (h.p[1 + i] == ' ') {
          sp = i;
          break;
        }
    }
    if (sp >= 2 && h.p[1 + sp - 2] == '/') sp -= 2;
    return sp;
  }
  long l = (long)L;
  if (fmt == FQG_NAME_DEFAULT) l--;
  return l >= 1 ? (uint32_t)(l - 1) : L;
}
__device__ __forceinline__ bool bc_same_bytes_wide(const uint8_t* a, const uint8_t* b, uint32_t n) {
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8)
    if (ld8(a + i) != ld8(b + i)) return false;
  if (i < n) return ((ld8(a + i) ^ ld8(b + i)) & ((1ull << (8 * (n - i))) - 1ull)) == 0;
  return true;
}

// slice_read's effect on one line (src/fastq_pre_barcodes.c:168-189): the result is
// s[from .. from+n) followed by '\n' when add_nl.  s = the line as a C string of length L.
struct Cut {
  uint32_t from, n, add_nl;
};
__device__ __forceinline__ Cut bc_cut(uint32_t L, long off, long size) {
  if (size == 0) return Cut{0, 0, 1};
  if (off > 0 && size == -1) return Cut{0, 0, 0};  // seq[-1]='\n'; seq[0]='\0'
  uint32_t from = 0, Lt = L;
  if (off > 0) {
    from = (uint32_t)((unsigned long)off < L ? off : L);
    Lt = L - from;
  }
  if (size < 0) return Cut{from, Lt, 0};
  if ((unsigned long)size < Lt) return Cut{from, (uint32_t)size, 1};
  if ((unsigned long)size == Lt) return Cut{from, Lt, 1};
  return Cut{from, Lt, 0};
}

__device__ __forceinline__ bool bc_slices(const BcParams& P, int x) {
  if (P.read_off[x] == -1) return false;
  if (P.read_off[x] == 0 && P.read_size[x] == -1) return false;
  return true;
}

__device__ __forceinline__ uint32_t dec_digits(unsigned long v) {
  if (!(v >> 32)) {  // compare chain: no division
    const uint32_t x = (uint32_t)v;
    return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) +
           (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
  }
  uint32_t d = 1;
  while (v >= 10) {
    v /= 10;
    ++d;
  }
  return d;
}

struct BcTags {
  uint32_t n[3];   // umi, cell, sample lengths (0: absent)
  uint32_t qn[3];  // quality characters that go with them: the reference copies them with strncpy from the quality
                   // STRING (src/fastq_pre_barcodes.c:250-252), which a malformed record may end before offset + size
  const uint8_t* s[3];
  const uint8_t* q[3];
};
// characters of a quality line (with its '\n', as gzgets leaves it) inside [off, off + size)
__device__ __forceinline__ uint32_t bc_qual_chars(const BcLine& q, long off, long size) {
  const long Lq = (long)(q.len + q.nl);
  const long n = Lq - off;
  return (uint32_t)(n < 0 ? 0 : (n < size ? n : size));
}

// get_barcode for one tag (src/fastq_pre_barcodes.c:218-259).  0 ok, 1 short, 2 low quality
template <bool WIDE>
__device__ __forceinline__ int bc_get(const BcLine (&ln)[4], long off, long size, int phred, int min_qual, uint32_t* n,
                                      uint32_t* qn, const uint8_t** s, const uint8_t** q) {
  *n = 0;
  *qn = 0;
  if (off == -1 || size == 0) return 0;
  const unsigned long rl1 = (unsigned long)(ln[1].len + ln[1].nl) - 1ul;
  if ((unsigned long)off > rl1 || (unsigned long)(off + size) > rl1) return 1;
  if (min_qual > 0) {
    const long Lq = (long)(ln[3].len + ln[3].nl);
    if (WIDE && off + size <= Lq) {  // 8 quality characters per LDS access
      for (long x = off; x < off + size; x += 8) {
        const uint64_t v = ld8(ln[3].p + x);
        const int left = (int)(off + size - x);
        bool low = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) low |= j < left && (int)(signed char)(v >> (8 * j)) - phred < min_qual;
        if (low) return 2;
      }
    } else {
      for (long x = off; x < off + size; ++x) {
        const int c = (int)(signed char)(x < Lq ? ln[3].p[x] : 0);
        if (c - phred < min_qual) return 2;
      }
    }
  }
  *n = (uint32_t)size;
  *qn = bc_qual_chars(ln[3], off, size);
  *s = ln[1].p + off;
  *q = ln[3].p + off;
  return 0;
}

// Do two header lines (in LDS, 8 bytes of slack behind them) surely have the same canonical name?  One pass over the
// words of both until they differ - no search for the end of either name:
//   * no difference up to and including the '\n' of the shorter line: the lines are the same line, and so are their
//     names when both files have the same name format;
//   * CASAVA 1.8 (the name ends at the first blank, src/fastq.c:502-511): the byte in front of the first difference is
//     a blank - the first blank of both lines lies in what they share ("... 1:N:0" against "... 2:N:0": the usual case);
//   * DEFAULT with is_pe (the name is the line without its last character, :489-495): lines of one length that differ
//     in that last character first.
// Anything else - and an unterminated last line - is decided by the exact comparison (bc_names_differ).
// The first 32 bytes of both lines are fetched at once when the lines are that long (BcHead: READ1's are fetched once
// for all files): the kernel that makes this test waits for every LDS round trip, so four of them cost what one does.
struct BcHead {
  uint64_t w[4];  // bytes 0..31 of the line
  bool wide;      // ... when it has them
};
__device__ __forceinline__ BcHead bc_head(const BcLine& a) {
  BcHead h;
  h.wide = a.len >= 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) h.w[j] = h.wide ? ld8(a.p + 8 * j) : 0ull;
  return h;
}
__device__ __forceinline__ bool bc_names_agree_fast(const BcLine& a, const BcHead& ha, int fa, const BcLine& b, int fb) {
  if (fa != fb || !(a.nl & b.nl)) return false;
  const uint32_t lmin = a.len < b.len ? a.len : b.len;  // position of the shorter line's '\n'
  uint32_t i = 1, at = 0;
  uint64_t d = 0;
  uint32_t before = 0x100u;  // the byte in front of the first difference, when it is known already
  if (ha.wide && b.len >= 32) {
    uint64_t x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = ha.w[j] ^ ld8(b.p + 8 * j);
    x[0] &= ~0xFFull;  // (byte 0: the '@' of both, checked by the caller)
    if (x[0] | x[1] | x[2] | x[3]) {
      const int j = x[0] ? 0 : (x[1] ? 1 : (x[2] ? 2 : 3));
      d = x[0] ? x[0] : (x[1] ? x[1] : (x[2] ? x[2] : x[3]));
      at = 8u * (uint32_t)j + ((uint32_t)__builtin_ctzll(d) >> 3);  // >= 1, < 32 <= lmin
      const uint32_t q = at - 1;
      const uint64_t wq = (q >> 3) == 0 ? ha.w[0] : ((q >> 3) == 1 ? ha.w[1] : ((q >> 3) == 2 ? ha.w[2] : ha.w[3]));
      before = (uint32_t)(wq >> (8u * (q & 7u))) & 0xFFu;
    } else {
      i = 32;
    }
  }
  if (!d) {
    for (; i <= lmin; i += 8) {
      d = ld8(a.p + i) ^ ld8(b.p + i);
      if (d) break;
    }
    if (!d) return true;  // every byte up to that '\n' is the same (lmin = 0: two empty names)
    at = i + ((uint32_t)__builtin_ctzll(d) >> 3);  // the first byte that differs
    if (at > lmin) return true;
  }
  if (fa == FQG_NAME_CASAVA18) return at >= 2 && (before < 0x100u ? before : (uint32_t)a.p[at - 1]) == ' ';
  if (fa == FQG_NAME_DEFAULT) return a.len == b.len && at + 1 == a.len && at >= 2;
  return false;
}

// canonical names: file x against READ1, exactly (src/fastq_pre_barcodes.c:606-635)
template <bool WIDE>
__device__ __forceinline__ bool bc_names_differ(const BcLine& h1, int f1, const BcLine& hx, int fx) {
  const uint32_t n1 = bc_name_len<WIDE>(h1, f1), nx = bc_name_len<WIDE>(hx, fx);
  bool same = nx == n1;
  if (WIDE) same = same && bc_same_bytes_wide(h1.p + 1, hx.p + 1, n1);
  else
    for (uint32_t i = 0; same && i < n1; ++i) same = h1.p[1 + i] == hx.p[1 + i];
  return !same;
}

// The name checks of one iteration: '@' first in every header (fastq_get_readname, src/fastq.c:448), then every file's
// canonical name against READ1's.  0, or (code << 3) | file of the first check that fails.  These checks come before
// anything else the reference does with an iteration (discards included), but nothing the plan computes depends on
// them: the EMIT kernels make them, on the header bytes they stage anyway (the plan then needs the bytes of the
// barcode-carrying files only).
template <bool WIDE, int MASK = 0>
__device__ __forceinline__ uint32_t bc_check_names(const BcParams& P, const BcLine (&L)[kBcFiles][4]) {
  if (P.n_inputs <= 1) return 0;
  uint32_t bad = 0;
#pragma unroll
  for (int x = kBcFiles - 1; x >= 1; --x)  // (all first bytes are fetched before any is looked at; the first file wins)
    if (bc_has<MASK>(P, x) && L[x][0].p[0] != '@') bad = (FQG_E_WRONG_HEADER << 3) | x;
  if (bad) return bad;
  BcHead h1;
  if (WIDE) h1 = bc_head(L[1][0]);
#pragma unroll
  for (int x = 2; x < kBcFiles; ++x)
    if (bc_has<MASK>(P, x)) {
      if (WIDE && bc_names_agree_fast(L[1][0], h1, P.f[1].fmt, L[x][0], P.f[x].fmt)) continue;
      if (bc_names_differ<WIDE>(L[1][0], P.f[1].fmt, L[x][0], P.f[x].fmt)) return (FQG_E_NAME_MISMATCH << 3) | x;
    }
  return 0;
}

// status + tags of one iteration from the lines of its records (in an image or in LDS): extract_info for each file in
// order - umi, sample, cell (src/fastq_pre_barcodes.c:262-285).  Only the lines of the barcode-carrying files are
// looked at, and their BYTES only when a minimum quality is asked for (bc_plan_staged).
template <bool WIDE, int MASK = 0>
__device__ __forceinline__ uint8_t bc_decide_lines(const BcParams& P, const BcLine (&L)[kBcFiles][4], BcTags* tags) {
  tags->n[0] = tags->n[1] = tags->n[2] = 0;
  tags->qn[0] = tags->qn[1] = tags->qn[2] = 0;
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x)
    if (bc_has<MASK>(P, x)) {
      int rc = 0;
      if (P.umi_read == x) rc = bc_get<WIDE>(L[x], P.umi_off, P.umi_size, P.phred, P.min_qual, &tags->n[0], &tags->qn[0], &tags->s[0], &tags->q[0]);
      if (!rc && P.sample_read == x)
        rc = bc_get<WIDE>(L[x], P.sample_off, P.sample_size, P.phred, P.min_qual, &tags->n[2], &tags->qn[2], &tags->s[2], &tags->q[2]);
      if (!rc && P.cell_read == x)
        rc = bc_get<WIDE>(L[x], P.cell_off, P.cell_size, P.phred, P.min_qual, &tags->n[1], &tags->qn[1], &tags->s[1], &tags->q[1]);
      if (rc) return rc == 1 ? kBcDiscardShort : kBcDiscardQual;
    }
  return kBcKeep;
}

// tags of an iteration that bc_decide_lines has already found to be kept: geometry only, none of
// the byte-by-byte checks
template <int MASK = 0>
__device__ __forceinline__ void bc_tags_of_kept(const BcParams& P, const BcLine (&lines)[kBcFiles][4], BcTags* tags) {
  tags->n[0] = tags->n[1] = tags->n[2] = 0;
  tags->qn[0] = tags->qn[1] = tags->qn[2] = 0;
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x)
    if (bc_has<MASK>(P, x) && (P.umi_read == x || P.sample_read == x || P.cell_read == x)) {
      const BcLine(&ln)[4] = lines[x];
      auto take = [&](int slot, long off, long size) {
        if (off == -1 || size == 0) return;
        tags->n[slot] = (uint32_t)size;
        tags->qn[slot] = bc_qual_chars(ln[3], off, size);
        tags->s[slot] = ln[1].p + off;
        tags->q[slot] = ln[3].p + off;
      };
      if (P.umi_read == x) take(0, P.umi_off, P.umi_size);
      if (P.sample_read == x) take(2, P.sample_off, P.sample_size);
      if (P.cell_read == x) take(1, P.cell_off, P.cell_size);
    }
}

// byte count of the FASTQ record written for a file (src/fastq_pre_barcodes.c:713-718);
// sliced / off / size: bc_slices(P, x), P.read_off[x], P.read_size[x]
__device__ __forceinline__ uint32_t bc_fastq_len(bool sliced, long off, long size, const BcLine (&ln)[4], const BcTags& t) {
  const bool tagged = (t.n[0] | t.n[1] | t.n[2]) != 0;
  uint32_t n = ln[0].len + ln[0].nl + (tagged ? 31u + t.n[0] + t.n[1] + t.n[2] : 0u);
  if (tagged && n >= (uint32_t)FQG_MAX_LABEL_LENGTH) {  // the tagged header runs over hdr1[] into hdr2[]: bc_emit_fastq
    n = n == (uint32_t)FQG_MAX_LABEL_LENGTH ? n : (uint32_t)FQG_MAX_LABEL_LENGTH + 2u;
    n += n == (uint32_t)FQG_MAX_LABEL_LENGTH ? 0u : 2u;
  } else
    n += (tagged || sliced) ? 2u : ln[2].len + ln[2].nl;
  if (sliced) {
    const Cut cs = bc_cut(ln[1].len + ln[1].nl, off, size);
    const Cut cq = bc_cut(ln[3].len + ln[3].nl, off, size);
    n += cs.n + cs.add_nl + cq.n + cq.add_nl;
  } else {
    n += ln[1].len + ln[1].nl + ln[3].len + ln[3].nl;
  }
  return n;
}

// what the SAM line prints for a mate: sequence / quality without their last character
// (src/fastq_pre_barcodes.c:666-700)
struct SamGeom {
  Cut cs, cq;
  uint32_t seq_n, qual_n;  // characters printed
  uint32_t shown;          // the number in column 9
  uint32_t name_n;         // characters of the on:Z: value
};
__device__ __forceinline__ SamGeom bc_sam_geom(bool sliced, long off, long size, bool mate1, const BcLine (&ln)[4]) {
  SamGeom g;
  const uint32_t Ls = ln[1].len + ln[1].nl, Lq = ln[3].len + ln[3].nl;
  if (sliced) {
    g.cs = bc_cut(Ls, off, size);
    g.cq = bc_cut(Lq, off, size);
  } else {
    g.cs = Cut{0, Ls, 0};
    g.cq = Cut{0, Lq, 0};
  }
  const uint32_t ls = g.cs.n + g.cs.add_nl, lq = g.cq.n + g.cq.add_nl;  // strlen after slicing
  g.seq_n = ls ? ls - 1 : 0;
  g.qual_n = lq ? lq - 1 : 0;
  g.shown = mate1 ? ls - 1u : ls;  // unsigned arithmetic as in the reference (len-1 for mate 1, len for mate 2)
  // format_read_name: up to the first '\n' of the header, without the '@'
  g.name_n = ln[0].len ? ln[0].len - 1 : 0;
  return g;
}

// one SAM line: number = the read number printed in column 1
__device__ __forceinline__ uint32_t bc_sam_line_len(unsigned long number, unsigned flag, const SamGeom& g, const BcTags& t) {
  uint32_t n = dec_digits(number) + 1 + dec_digits(flag);
  n += 15;  // "\t*\t0\t255\t*\t*\t0\t"
  n += dec_digits(g.shown) + 1 + g.seq_n + 1 + g.qual_n + 6 + g.name_n + 6 + g.qual_n;
  if (t.n[0]) n += 12 + t.n[0] + t.qn[0];
  if (t.n[1]) n += 12 + t.n[1] + t.qn[1];
  if (t.n[2]) n += 12 + t.n[2] + t.qn[2];
  return n + 1;
}
__device__ __forceinline__ unsigned bc_sam_flag(bool se, bool mate1) {
  return se ? 4u : (mate1 ? 77u : 141u);  // BAM_FUNMAP | FMUNMAP | FPAIRED | FREAD1/2
}

// byte counts of the outputs of a kept iteration
__device__ __forceinline__ void bc_out_lens(const BcParams& P, uint64_t k, const BcLine (&L)[kBcFiles][4], const BcTags& t,
                                            uint32_t* a, uint32_t* b, uint32_t* c) {
  *a = *b = *c = 0;
  if (P.out_sam) {
    const bool se = !P.f[2].present;
    *a = bc_sam_line_len(P.first_read_number + k + 1, bc_sam_flag(se, true),
                         bc_sam_geom(bc_slices(P, 1), P.read_off[1], P.read_size[1], true, L[1]), t);
    if (!se)
      *a += bc_sam_line_len(P.first_read_number + k + 1, bc_sam_flag(se, false),
                            bc_sam_geom(bc_slices(P, 2), P.read_off[2], P.read_size[2], false, L[2]), t);
  } else {
    if (P.emit[1]) *b = bc_fastq_len(bc_slices(P, 1), P.read_off[1], P.read_size[1], L[1], t);
    if (P.emit[2]) *c = bc_fastq_len(bc_slices(P, 2), P.read_off[2], P.read_size[2], L[2], t);
  }
}

// bc_decide_lines and bc_out_lens for the kernels for any set of files: the lines of a file are asked for (lines_of(x, ln))
// only inside the branch that looks at them - files that carry no barcode are never asked for
template <bool WIDE, int MASK, class LinesOf>
__device__ __forceinline__ uint8_t bc_decide_with(const BcParams& P, LinesOf lines_of, BcTags* tags) {
  tags->n[0] = tags->n[1] = tags->n[2] = 0;
  tags->qn[0] = tags->qn[1] = tags->qn[2] = 0;
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x)
    if (bc_has<MASK>(P, x) && (P.umi_read == x || P.sample_read == x || P.cell_read == x)) {
      BcLine ln[4];
      lines_of(x, ln);
      int rc = 0;
      if (P.umi_read == x) rc = bc_get<WIDE>(ln, P.umi_off, P.umi_size, P.phred, P.min_qual, &tags->n[0], &tags->qn[0], &tags->s[0], &tags->q[0]);
      if (!rc && P.sample_read == x)
        rc = bc_get<WIDE>(ln, P.sample_off, P.sample_size, P.phred, P.min_qual, &tags->n[2], &tags->qn[2], &tags->s[2], &tags->q[2]);
      if (!rc && P.cell_read == x)
        rc = bc_get<WIDE>(ln, P.cell_off, P.cell_size, P.phred, P.min_qual, &tags->n[1], &tags->qn[1], &tags->s[1], &tags->q[1]);
      if (rc) return rc == 1 ? kBcDiscardShort : kBcDiscardQual;
    }
  return kBcKeep;
}
template <int MASK, class LinesOf>
__device__ __forceinline__ void bc_out_lens_with(const BcParams& P, uint64_t k, LinesOf lines_of, const BcTags& t, uint32_t* a,
                                                 uint32_t* b, uint32_t* c) {
  *a = *b = *c = 0;
  const bool se = !bc_has<MASK>(P, 2);
  if (P.out_sam || P.emit[1]) {
    BcLine ln[4];
    lines_of(1, ln);
    if (P.out_sam)
      *a = bc_sam_line_len(P.first_read_number + k + 1, bc_sam_flag(se, true),
                           bc_sam_geom(bc_slices(P, 1), P.read_off[1], P.read_size[1], true, ln), t);
    else *b = bc_fastq_len(bc_slices(P, 1), P.read_off[1], P.read_size[1], ln, t);
  }
  if (!se && (P.out_sam || P.emit[2])) {
    BcLine ln[4];
    lines_of(2, ln);
    if (P.out_sam)
      *a += bc_sam_line_len(P.first_read_number + k + 1, bc_sam_flag(se, false),
                            bc_sam_geom(bc_slices(P, 2), P.read_off[2], P.read_size[2], false, ln), t);
    else *c = bc_fastq_len(bc_slices(P, 2), P.read_off[2], P.read_size[2], ln, t);
  }
}

// ---- wavefront helpers --------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_max32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ unsigned long long wave_min64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(v, d, 64);
    v = o < v ? o : v;
  }
  return v;
}

// ---- tiles: the records of T consecutive iterations of every file, copied into LDS ----------------
// Two dependent round trips to memory per tile: the line index of the tile's records (what a lane
// needs to know its iteration's lines, and lane 0 / the last lane to know the tile's byte span),
// then the spans themselves.  The emit kernels and the record filters keep three tiles under way (index of the
// tile after next and spans of the next tile in flight while the current one is written); the plan kernel of
// fastq_pre_barcodes requests the next tile's index with the current tile's spans.  Units are aligned on the
// image ADDRESS, so a unit always holds at least one byte of the image and never leaves its page.
// (Nothing is computed from the loaded values where they are requested - no select, no addition: an instruction that
// reads them there is a wait for the memory round trip there.  The requests of a tile are made one tile ahead.)
struct BcGeo {
  uint64_t praw, e[4];  // line ends of the lane's record, and of the line before it (record 0: its own first line end, unused)
  uint32_t r0;          // the lane's record is record 0 of its file
  __device__ __forceinline__ uint64_t start() const { return r0 ? 0ull : praw + 1; }  // first byte of the record
};
struct TileGeo {
  BcGeo f[kBcFiles];
  // emit: the iteration's place in each output is off + sum (added where it is used: an addition inside the function
  // that requests the loads would make the wavefront wait for them there)
  unsigned long long off[3], sum[3];
  uint32_t olen[3];             // emit: ... and the bytes it has there (what the plan computed)
  uint8_t st;                   // emit: status
  uint8_t big;                  // emit: tile flag
};
__device__ __forceinline__ void bc_geo_load(const BcFile& f, uint64_t k, BcGeo& g) {
  const uint64_t r = f.first + k * f.step + f.add;
  const uint64_t* __restrict__ le = f.fv.line_end + 4 * r;
  g.r0 = r == 0 ? 1u : 0u;
  g.praw = le[r == 0 ? 0 : -1];
#pragma unroll
  for (int i = 0; i < 4; ++i) g.e[i] = le[i];
}
// which files a kernel stages.  The emit kernels (and the record filters): every input - they print the records and
// compare the names.  PLAN: only the files a barcode is cut from, and those only when a minimum quality is asked for
// (get_barcode reads quality characters then); everything else the plan needs - line lengths - is in the line index.
template <bool PLAN, int MASK>
__device__ __forceinline__ bool bc_staged(const BcParams& P, int x) {
  if (!bc_has<MASK>(P, x)) return false;
  if (PLAN && (P.min_qual <= 0 || (P.umi_read != x && P.cell_read != x && P.sample_read != x))) return false;
  return true;
}

// Where the spans of a tile lie: the tile's records of every staged file are one byte span, cut into 16-byte units
// aligned on the image ADDRESS; the units of all files are numbered through.
struct SpanPlan {
  const uint8_t* gbase[kBcFiles];  // address of the file's unit 0 minus 16 * its first unit number
  uint32_t first_unit[kBcFiles], skew[kBcFiles];
  uint64_t s0[kBcFiles];
  uint32_t units;
  bool fit;  // the spans fit in_cap
};
template <bool PLAN, int MASK>
__device__ __forceinline__ void bc_span_plan(const BcParams& P, const TileGeo& tg, int last_lane, uint32_t in_cap, SpanPlan& sp) {
  sp.units = 0;
  sp.fit = true;
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x) {
    sp.first_unit[x] = 0xFFFFFFFFu;  // never selected
    sp.gbase[x] = P.f[1].fv.img;
    sp.skew[x] = 0;
    sp.s0[x] = 0;
    if (!bc_staged<PLAN, MASK>(P, x)) continue;
    const BcFile& f = P.f[x];
    sp.s0[x] = rfl64(tg.f[x].start());
    const uint64_t e3l = rl64(tg.f[x].e[3], last_lane);
    const uint64_t n = (e3l < f.fv.nbytes ? e3l + 1 : e3l) - sp.s0[x];
    sp.skew[x] = (uint32_t)((uintptr_t)(f.fv.img + sp.s0[x]) & 15u);
    sp.first_unit[x] = sp.units;
    sp.gbase[x] = f.fv.img + sp.s0[x] - sp.skew[x] - 16ull * sp.units;
    if (n > (uint64_t)in_cap) sp.fit = false;
    else sp.units += (sp.skew[x] + (uint32_t)n + 15u) >> 4;
    if ((uint64_t)sp.units * 16u + 32u > (uint64_t)in_cap) sp.fit = false;
  }
}
typedef uint32_t bc_u32x4 __attribute__((ext_vector_type(4)));
// unit u of a tile (clamped to the tile's last unit: no branch, every load of a round in flight)
__device__ __forceinline__ bc_u32x4 bc_span_unit(const SpanPlan& sp, uint32_t u) {
  u = u < sp.units ? u : (sp.units ? sp.units - 1 : 0u);  // (no unit at all: unit 0 of READ1's image, never stored)
  const uint8_t* base = sp.gbase[1];  // (the staged file that comes first has first_unit 0: READ1, or the loop finds it)
#pragma unroll
  for (int x = 2; x < kBcFiles; ++x) {
    // (a VALUE on either side: "c ? sp.gbase[x] : base" is an lvalue - the compiler selects between the two ADDRESSES and
    // loads once, the plan stays in scratch memory, and every unit of the any-set-of-files kernels waited for that load)
    const uint8_t* const gx = sp.gbase[x];
    base = u >= sp.first_unit[x] ? static_cast<const uint8_t*>(gx) : static_cast<const uint8_t*>(base);
  }
  return __builtin_nontemporal_load(reinterpret_cast<const bc_u32x4*>(base + 16ull * u));
}
// units from_unit .. of the tile -> s_in, 8 loads in flight per lane
__device__ __forceinline__ void bc_span_copy(const SpanPlan& sp, uint32_t from_unit, int lane, uint8_t* s_in) {
  for (uint32_t u0 = from_unit; u0 < sp.units; u0 += 8 * kWave) {
    bc_u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bc_span_unit(sp, u0 + j * kWave + lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t u = u0 + j * kWave + lane;
      if (u < sp.units) *reinterpret_cast<bc_u32x4*>(s_in + 16u * u) = v[j];
    }
  }
}
// the lane's lines: pointers into the staged spans (a file that is not staged: the lengths of its lines, no bytes -
// pointers that are never followed)
// (one file's lines: the kernels for any set of files ask for them where they use them - five files' lines at once are 80
// registers)
template <bool PLAN, int MASK>
__device__ __forceinline__ void bc_file_lines(const BcParams& P, const TileGeo& tg, const SpanPlan& sp, uint8_t* s_in, int x,
                                              BcLine (&ln)[4]) {
  const BcGeo& g = tg.f[x];
  const uint64_t nb = P.f[x].fv.nbytes;
  const bool st = bc_staged<PLAN, MASK>(P, x);
  uint8_t* b = st ? s_in + 16u * sp.first_unit[x] + sp.skew[x] : s_in;
  const uint64_t s0 = st ? sp.s0[x] : g.start();
  ln[0] = BcLine{b + (uint32_t)(g.start() - s0), (uint32_t)(g.e[0] - g.start()), g.e[0] < nb ? 1u : 0u};
  ln[1] = BcLine{b + (uint32_t)(g.e[0] + 1 - s0), (uint32_t)(g.e[1] - g.e[0] - 1), g.e[1] < nb ? 1u : 0u};
  ln[2] = BcLine{b + (uint32_t)(g.e[1] + 1 - s0), (uint32_t)(g.e[2] - g.e[1] - 1), g.e[2] < nb ? 1u : 0u};
  ln[3] = BcLine{b + (uint32_t)(g.e[2] + 1 - s0), (uint32_t)(g.e[3] - g.e[2] - 1), g.e[3] < nb ? 1u : 0u};
}
template <bool PLAN, int MASK>
__device__ __forceinline__ void bc_span_lines(const BcParams& P, const TileGeo& tg, const SpanPlan& sp, uint8_t* s_in,
                                              BcLine (&L)[kBcFiles][4]) {
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x) {
    if (!bc_has<MASK>(P, x)) continue;
    bc_file_lines<PLAN, MASK>(P, tg, sp, s_in, x, L[x]);
  }
}

// Copies the spans into s_in and gives the lane its lines (pointers into LDS).  Returns false
// (uniformly, before copying anything) when the spans do not fit in_cap.
template <bool PLAN, int MASK>
__device__ __forceinline__ bool bc_stage_tile(const BcParams& P, const TileGeo& tg, int last_lane, int lane, uint8_t* s_in,
                                              uint32_t in_cap, BcLine (&L)[kBcFiles][4]) {
  SpanPlan sp;
  bc_span_plan<PLAN, MASK>(P, tg, last_lane, in_cap, sp);
  if (!sp.fit) return false;
  bc_span_copy(sp, 0, lane, s_in);
  bc_span_lines<PLAN, MASK>(P, tg, sp, s_in, L);
  return true;
}

// The emit kernel requests the first kSpanPf * 64 units of a tile (10 KiB: every tile of the default LDS budget) as
// straight-line code - a loop makes the compiler wait for every request in flight where the loop begins, the next
// tile's index among them.
constexpr int kSpanPf = 10;
__device__ __forceinline__ void bc_span_fetch(const SpanPlan& sp, int lane, bc_u32x4 (&v)[kSpanPf]) {
#pragma unroll
  for (int j = 0; j < kSpanPf; ++j) v[j] = bc_span_unit(sp, (uint32_t)(j * kWave + lane));
}
__device__ __forceinline__ void bc_span_land(const SpanPlan& sp, int lane, const bc_u32x4 (&v)[kSpanPf], uint8_t* s_in) {
#pragma unroll
  for (int j = 0; j < kSpanPf; ++j) {
    const uint32_t u = (uint32_t)(j * kWave + lane);
    if (u < sp.units) *reinterpret_cast<bc_u32x4*>(s_in + 16u * u) = v[j];
  }
  bc_span_copy(sp, kSpanPf * kWave, lane, s_in);  // (larger tiles: the rest now)
}

// Would the records of the iterations held by lanes first_lane .. last_lane fit the input area of ONE emit
// tile?  The same unit count as bc_stage_tile makes for that tile (the emit kernels stage every input).
template <int MASK>
__device__ __forceinline__ bool bc_emit_tile_fits(const BcParams& P, const TileGeo& tg, int first_lane, int last_lane,
                                                  uint32_t in_cap) {
  uint32_t units = 0;
  bool fit = true;
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x) {
    if (!bc_has<MASK>(P, x)) continue;
    const BcFile& f = P.f[x];
    const uint64_t s0 = rl64(tg.f[x].start(), first_lane);
    const uint64_t e3l = rl64(tg.f[x].e[3], last_lane);
    const uint64_t n = (e3l < f.fv.nbytes ? e3l + 1 : e3l) - s0;
    const uint32_t skew = (uint32_t)((uintptr_t)(f.fv.img + s0) & 15u);
    if (n > (uint64_t)in_cap) fit = false;
    else units += (skew + (uint32_t)n + 15u) >> 4;
    if ((uint64_t)units * 16u + 32u > (uint64_t)in_cap) fit = false;
  }
  return fit;
}

template <int MASK = 0>
__device__ __forceinline__ void bc_lines_all(const BcParams& P, uint64_t k, BcLine (&L)[kBcFiles][4]) {
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x)
    if (bc_has<MASK>(P, x)) bc_lines(P.f[x], k, L[x]);
}

// One wavefront per PLAN tile = plan_m consecutive tiles of the emit kernel, one lane per iteration.  The plan decides
// what the iteration's output is made of - keep / discard (get_barcode's bounds and minimum quality) and the byte count
// of every output - from the line index and the bytes of the barcode-carrying files alone (bc_staged<PLAN>): its LDS
// holds little, so many wavefronts are resident.  The name checks, which need the header bytes of every file, are made
// by the emit kernels on the records they stage anyway (bc_check_names).  What the plan decides per emit tile - does
// the tile fit the emit kernel's LDS areas - is worked out per group of T lanes.  MASK: see bc_has.
//
// How each instantiation hides its round trips to memory (measured on 200 M pairs of READ1 + INDEX1 and 50 M iterations of
// four files; the other sets of files keep what they had):
//   any set of files (MASK 0): no index of a second tile in flight, a file's lines made where they are looked at
//     (bc_decide_with) - 128 registers, FOUR wavefronts per SIMD: 2.0 ms where five files' lines at once (233 registers,
//     two wavefronts) took 3.6;
//   READ1 + INDEX1: no second tile in flight either, 96 registers, FIVE wavefronts: 7.8 ms against 8.4 with the next
//     tile's index requested ahead and four wavefronts;
//   the other specialised sets: the next tile's index requested ahead (many wavefronts are resident here - little LDS -
//     and hide each other's round trips: the three-tile scheme of the emit kernel made this kernel slower, 7.7 -> 11.2 ms).
template <int MASK>
struct PlanShape {
  static constexpr bool ahead = MASK != 0 && MASK != 0x0A;
  static constexpr int waves = MASK == 0 ? 4 : (MASK == 0x0A ? 5 : 1);
};
template <int MASK>
__global__ __launch_bounds__(kWave, PlanShape<MASK>::waves) void k_bc_plan_tile(BcParams P, BcTile tc, uint64_t n_iter, uint8_t* __restrict__ status,
                                                        uint32_t* __restrict__ len0, uint32_t* __restrict__ len1,
                                                        uint32_t* __restrict__ len2, uint8_t* __restrict__ tile_big,
                                                        BcCall* __restrict__ call) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
  const int lane = (int)threadIdx.x;
  const uint32_t Tp = tc.T * tc.plan_m, plan_cap = tc.plan_cap;
  const uint64_t n_tiles = (n_iter + Tp - 1) / Tp;
  auto tile_size = [&](uint64_t tile) {
    const uint64_t left = n_iter - tile * Tp;
    return (uint32_t)(left < (uint64_t)Tp ? left : (uint64_t)Tp);
  };
  auto geo_of = [&](uint64_t tile, TileGeo& tg) {
    const uint32_t Tn = tile_size(tile);
    const uint64_t k = tile * Tp + ((uint32_t)lane < Tn ? (uint32_t)lane : Tn - 1);
#pragma unroll
    for (int x = 1; x < kBcFiles; ++x)
      if (bc_has<MASK>(P, x)) bc_geo_load(P.f[x], k, tg.f[x]);
  };
  TileGeo cur, nxt;
  uint32_t n_dropped = 0, n_short = 0;  // of this wavefront's tiles: added to the call's totals once, at the end
  constexpr bool kAhead = PlanShape<MASK>::ahead;  // (false: a tile's index is asked for when the wavefront gets there)
  if (kAhead && blockIdx.x < n_tiles) geo_of(blockIdx.x, cur);
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t k0 = tile * Tp;
    const uint32_t Tn = tile_size(tile);
    const bool valid = (uint32_t)lane < Tn;
    const uint64_t k = k0 + (valid ? (uint32_t)lane : Tn - 1);
    BcLine L[kBcFiles][4];
    if (kAhead) geo_of(tile + gridDim.x < n_tiles ? tile + gridDim.x : n_tiles - 1, nxt);  // the next tile's index (requested without a branch)
    else geo_of(tile, cur);
    // (an image with NUL bytes: lines are C strings, found by scanning them where they lie - bc_lines)
    BcTags t;
    uint32_t a = 0, b = 0, c = 0;
    uint8_t st;
    if constexpr (MASK == 0) {
      // any set of files: a file's lines are made where they are looked at (bc_decide_with) - never all five at once
      SpanPlan sp;
      bc_span_plan<true, MASK>(P, cur, (int)Tn - 1, plan_cap, sp);
      const bool fit = !P.has_nul && sp.fit;
      if (fit) bc_span_copy(sp, 0, lane, s_lds);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (fit) {
        auto staged = [&](int x, BcLine(&ln)[4]) { bc_file_lines<true, MASK>(P, cur, sp, s_lds, x, ln); };
        st = bc_decide_with<true, MASK>(P, staged, &t);
        if (st == kBcKeep) bc_out_lens_with<MASK>(P, k, staged, t, &a, &b, &c);
      } else {  // long reads: from the images
        auto from_image = [&](int x, BcLine(&ln)[4]) { bc_lines(P.f[x], k, ln); };
        st = bc_decide_with<false, MASK>(P, from_image, &t);
        if (st == kBcKeep) bc_out_lens_with<MASK>(P, k, from_image, t, &a, &b, &c);
      }
    } else {
      const bool fit = !P.has_nul && bc_stage_tile<true, MASK>(P, cur, (int)Tn - 1, lane, s_lds, plan_cap, L);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (fit) {
        st = bc_decide_lines<true, MASK>(P, L, &t);
        if (st == kBcKeep) bc_out_lens(P, k, L, t, &a, &b, &c);
      } else {  // long reads: from the images
        BcLine G[kBcFiles][4];
        bc_lines_all<MASK>(P, k, G);
        st = bc_decide_lines<false, MASK>(P, G, &t);
        if (st == kBcKeep) bc_out_lens(P, k, G, t, &a, &b, &c);
      }
    }
    if (valid) {
      status[k] = st;
      if (P.out_sam) len0[k] = a;
      else {
        if (P.emit[1]) len1[k] = b;
        if (P.emit[2]) len2[k] = c;
      }
    } else {
      a = b = c = 0;
    }
    const unsigned long long none = ~0ull;
    const bool dropped = valid && (st == kBcDiscardShort || st == kBcDiscardQual);
    n_dropped += dropped ? 1u : 0u;
    n_short += valid && st == kBcDiscardShort ? 1u : 0u;
    const unsigned long long dsc = wave_min64(dropped ? (unsigned long long)k : none);
    // the emit tiles inside this plan tile
    for (uint32_t j = 0; j * tc.T < Tn; ++j) {
      const uint32_t first = j * tc.T, last = (first + tc.T < Tn ? first + tc.T : Tn) - 1;
      const bool mine = (uint32_t)lane >= first && (uint32_t)lane <= last;
      const uint32_t sa = wave_sum32(mine ? a : 0u), sb = wave_sum32(mine ? b : 0u), sc = wave_sum32(mine ? c : 0u);
      const bool fits_in = bc_emit_tile_fits<MASK>(P, cur, (int)first, (int)last, tc.in_cap);
      if (lane == 0) {
        const bool big = P.has_nul || !fits_in || sa + 32 > tc.out_cap || sb + 32 > tc.out_cap || sc + 32 > tc.out_cap;
        tile_big[tile * tc.plan_m + j] = big ? 1 : 0;
        if (big) atomicAdd(&call->big, 1ull);
      }
    }
    // tiles run roughly in order: look before the atomic, almost every later tile has nothing to add
    if (lane == 0 && dsc != none && dsc < __atomic_load_n(&call->first_discard, __ATOMIC_RELAXED))
      atomicMin(&call->first_discard, dsc);
    __builtin_amdgcn_wave_barrier();
    if (kAhead) cur = nxt;
  }
  // the discarded iterations of all n_iter (k_bc_count counts again when the batch ends earlier: interleaved input)
  n_dropped = wave_sum32(n_dropped);
  n_short = wave_sum32(n_short);
  if (lane == 0) {
    if (n_dropped) atomicAdd(&call->discarded, (unsigned long long)n_dropped);
    if (n_short) atomicAdd(&call->short_warnings, (unsigned long long)n_short);
  }
}

// a name finding of the emit kernels: the smallest (iteration << 8 | code << 3 | file) wins
__device__ __forceinline__ void bc_report_finding(BcCall* __restrict__ call, bool active, uint64_t k, uint32_t finding) {
  if (!__ballot(active && finding != 0)) return;  // (the usual case: one ballot)
  const unsigned long long none = ~0ull;
  const unsigned long long fnd = wave_min64(active && finding ? (unsigned long long)((k << 8) | finding) : none);
  if ((threadIdx.x & 63) == 0 && fnd < __atomic_load_n(&call->first_finding, __ATOMIC_RELAXED)) atomicMin(&call->first_finding, fnd);
}

// discarded iterations among the first n_done: one atomic per workgroup
__global__ __launch_bounds__(kBlock) void k_bc_count(const uint8_t* __restrict__ status, uint64_t n_done,
                                                     BcCall* __restrict__ call) {
  __shared__ unsigned long long s_d[kBlock / kWave], s_s[kBlock / kWave];
  unsigned long long d = 0, sh = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < n_done; k += (uint64_t)gridDim.x * kBlock) {
    const uint8_t st = status[k];
    d += (st == kBcDiscardShort || st == kBcDiscardQual) ? 1u : 0u;
    sh += st == kBcDiscardShort ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    d += __shfl_xor(d, o, 64);
    sh += __shfl_xor(sh, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_d[threadIdx.x >> 6] = d;
    s_s[threadIdx.x >> 6] = sh;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long td = 0, ts = 0;
    for (int w = 0; w < kBlock / kWave; ++w) {
      td += s_d[w];
      ts += s_s[w];
    }
    if (td) atomicAdd(&call->discarded, td);
    if (ts) atomicAdd(&call->short_warnings, ts);
  }
}

// ---- 64-bit exclusive scan of u32 lengths: 2048 per workgroup ---------------------------------
constexpr int kScan64Span = kBlock * 8;
typedef unsigned long long bc_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void scan64_a_body(const uint32_t* __restrict__ in, uint64_t n,
                                              unsigned long long* __restrict__ local,
                                              unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long s_w[kBlock / kWave];
  const uint64_t first = (uint64_t)blockIdx.x * kScan64Span + threadIdx.x * 8;
  // a span that lies inside the array (all but the last one), arrays on 16-byte addresses: the lane's eight lengths in two
  // loads, its eight offsets in four stores
  const bool whole = ((uint64_t)blockIdx.x + 1) * kScan64Span <= n && (((uintptr_t)in | (uintptr_t)local) & 15u) == 0;
  unsigned long long v[8], sum = 0;
  if (whole) {
    const bc_u32x4 a = *reinterpret_cast<const bc_u32x4*>(in + first), b = *reinterpret_cast<const bc_u32x4*>(in + first + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = first + i < n ? in[first + i] : 0u;
      sum += v[i];
    }
  }
  unsigned long long incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(incl, d, 64);
    if ((int)(threadIdx.x & 63) >= d) incl += o;
  }
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned long long before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) {
    if (w < (int)(threadIdx.x >> 6)) before += s_w[w];
    all += s_w[w];
  }
  unsigned long long run = before + incl - sum;
  if (whole) {
    unsigned long long o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[i] = run;
      run += v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) *reinterpret_cast<bc_u64x2*>(local + first + i) = bc_u64x2{o[i], o[i + 1]};
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (first + i < n) local[first + i] = run;
      run += v[i];
    }
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = all;
}
// one workgroup: exclusive prefix over the span sums, total into *total
__global__ __launch_bounds__(kBlock) void k_scan64_a(const uint32_t* __restrict__ in, uint64_t n,
                                                     unsigned long long* __restrict__ local,
                                                     unsigned long long* __restrict__ sums) {
  scan64_a_body(in, n, local, sums);
}

// (2048 sums per round - eight per thread, a wavefront scan by shuffles, one exchange through LDS: the scan of the
// 97 656 span sums of 200 M lengths took 0.49 ms as 382 rounds of a 256-wide scan with sixteen barriers each)
__device__ __forceinline__ void scan64_b_body(unsigned long long* __restrict__ sums, uint64_t nb,
                                              unsigned long long* __restrict__ total) {
  __shared__ unsigned long long s_w[kBlock / kWave];
  unsigned long long carry = 0;  // (the same in every thread)
  for (uint64_t base = 0; base < nb; base += kScan64Span) {
    const uint64_t first = base + threadIdx.x * 8;
    unsigned long long v[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = first + i < nb ? sums[first + i] : 0ull;
      sum += v[i];
    }
    unsigned long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = __shfl_up(incl, d, 64);
      if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) {
      if (w < (int)(threadIdx.x >> 6)) before += s_w[w];
      all += s_w[w];
    }
    unsigned long long run = carry + before + incl - sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (first + i < nb) sums[first + i] = run;
      run += v[i];
    }
    carry += all;
    __syncthreads();  // (s_w is written again in the next round)
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(kBlock) void k_scan64_b(unsigned long long* __restrict__ sums, uint64_t nb,
                                                     unsigned long long* __restrict__ total) {
  scan64_b_body(sums, nb, total);
}
// the same for three arrays in one launch (blockIdx.y picks the array): bam_umi_count numbers UMIs, cells and features
// by three such scans, and a launch costs more than the scan of a few million flags
struct Scan3 {
  const uint32_t* first[3];          // k_umi_flag3: table -> first record of the key in the slot
  uint32_t* flag[3];
  unsigned long long* local[3];
  unsigned long long* sums[3];
  unsigned long long* total;         // [3]
};
__global__ __launch_bounds__(kBlock) void k_scan64_a3(Scan3 t, uint64_t n) {
  scan64_a_body(t.flag[blockIdx.y], n, t.local[blockIdx.y], t.sums[blockIdx.y]);
}
__global__ __launch_bounds__(kBlock) void k_scan64_b3(Scan3 t, uint64_t nb) {
  scan64_b_body(t.sums[blockIdx.y], nb, t.total + blockIdx.y);
}

// ---- emit -----------------------------------------------------------------------------------
// up to 16 characters of a literal as two 64-bit immediates (no memory access when written)
constexpr uint64_t lit_pack(const char* s, int n, int from) {
  uint64_t v = 0;
  for (int i = 0; i < 8 && from + i < n; ++i) v |= (uint64_t)(uint8_t)s[from + i] << (8 * i);
  return v;
}
#define BC_LIT(w, str) (w).lit_imm(lit_pack(str, (int)sizeof(str) - 1, 0), lit_pack(str, (int)sizeof(str) - 1, 8), (uint32_t)sizeof(str) - 1)

// the whole wavefront writes one text: lane i copies byte i of every piece (k_bc_emit_direct)
struct Writer {
  uint8_t* p;
  int lane;
  __device__ __forceinline__ void lit_imm(uint64_t lo, uint64_t hi, uint32_t n) {
    if ((uint32_t)lane < n) p[lane] = (uint8_t)((lane < 8 ? lo >> (8 * lane) : hi >> (8 * (lane - 8))) & 0xFFu);
    p += n;
  }
  __device__ __forceinline__ void bytes(const uint8_t* s, uint32_t n) {
    for (uint32_t i = lane; i < n; i += kWave) p[i] = s[i];
    p += n;
  }
  __device__ __forceinline__ void ch(char c) {
    if (lane == 0) p[0] = (uint8_t)c;
    p += 1;
  }
  __device__ __forceinline__ void dec(unsigned long v) {
    const uint32_t d = dec_digits(v);
    if (lane == 0) {
      for (uint32_t i = 0; i < d; ++i) {
        p[d - 1 - i] = (uint8_t)('0' + v % 10);
        v /= 10;
      }
    }
    p += d;
  }
  // header text for on:Z: - blanks become '@' (format_read_name, src/fastq_pre_barcodes.c:300-308)
  __device__ __forceinline__ void name(const uint8_t* s, uint32_t n) {
    for (uint32_t i = lane; i < n; i += kWave) p[i] = s[i] == ' ' ? (uint8_t)'@' : s[i];
    p += n;
  }
  __device__ __forceinline__ void bytes_twice(const uint8_t* s, uint32_t n, uint32_t dist) {
    for (uint32_t i = lane; i < n; i += kWave) p[i] = p[dist + i] = s[i];
    p += n;
  }
  __device__ __forceinline__ void skip(uint32_t n) { p += n; }
};

// one lane writes one text (k_bc_emit_tile: sources and destination are LDS, at any alignment).
// Copies move 8 bytes per LDS access; the last access of a piece overlaps the one before it, or
// is split 4/2/1 for pieces shorter than 8.
struct LaneWriter {
  uint8_t* p;
  template <int N>
  __device__ __forceinline__ void put(uint64_t v) {  // the low N bytes of v
    if (N == 8) __builtin_memcpy(p, &v, 8);
    else {
      uint32_t done = 0;
      if (N & 4) {
        const uint32_t w = (uint32_t)v;
        __builtin_memcpy(p, &w, 4);
        done = 4;
      }
      if (N & 2) {
        const uint16_t w = (uint16_t)(v >> (8 * done));
        __builtin_memcpy(p + done, &w, 2);
        done += 2;
      }
      if (N & 1) p[done] = (uint8_t)(v >> (8 * done));
    }
  }
  __device__ __forceinline__ void lit_imm(uint64_t lo, uint64_t hi, uint32_t n) {
    // n is a literal's length: the branches fold
    if (n >= 8) {
      put<8>(lo);
      p += 8;
      lo = hi;
      n -= 8;
    }
    switch (n) {
      case 8: put<8>(lo); break;
      case 7: put<7>(lo); break;
      case 6: put<6>(lo); break;
      case 5: put<5>(lo); break;
      case 4: put<4>(lo); break;
      case 3: put<3>(lo); break;
      case 2: put<2>(lo); break;
      case 1: put<1>(lo); break;
      default: break;
    }
    p += n;
  }
  static __device__ __forceinline__ uint64_t at_for_blank(uint64_t x) {
    const uint64_t z = bytes_eq(x, (uint8_t)' ');  // ' ' 0x20 -> '@' 0x40
    return x ^ ((z >> 1) | (z >> 2));
  }
  // TWICE: the same bytes also go to p + dist (the SAM line prints the quality string twice)
  template <bool NAME, bool TWICE = false>
  __device__ __forceinline__ void copy(const uint8_t* s, uint32_t n, uint32_t dist = 0) {
    auto st8 = [&](uint32_t at, uint64_t v) {
      if (NAME) v = at_for_blank(v);
      __builtin_memcpy(p + at, &v, 8);
      if (TWICE) __builtin_memcpy(p + dist + at, &v, 8);
    };
    if (n >= 8) {
      uint32_t i = 0;
      // (the wavefront waits for every LDS round trip: four reads are in flight per trip while the piece is long)
      for (; i + 32 <= n; i += 32) {
        const uint64_t a = ld8(s + i), b = ld8(s + i + 8), c = ld8(s + i + 16), d = ld8(s + i + 24);
        st8(i, a);
        st8(i + 8, b);
        st8(i + 16, c);
        st8(i + 24, d);
      }
      if (i + 16 <= n) {
        const uint64_t a = ld8(s + i), b = ld8(s + i + 8);
        st8(i, a);
        st8(i + 8, b);
        i += 16;
      }
      if (i + 8 <= n) {
        st8(i, ld8(s + i));
        i += 8;
      }
      if (i < n) st8(n - 8, ld8(s + n - 8));  // the last 8 bytes again: they overlap what is already there with the same values
    } else if (n) {
      uint64_t a = ld8(s);  // the sources have 8 bytes of slack
      if (NAME) a = at_for_blank(a);
#pragma unroll
      for (int rep = 0; rep < (TWICE ? 2 : 1); ++rep) {
        uint8_t* q = p + (rep ? dist : 0u);
        uint32_t done = 0;
        if (n & 4) {
          const uint32_t w = (uint32_t)a;
          __builtin_memcpy(q, &w, 4);
          done = 4;
        }
        if (n & 2) {
          const uint16_t w = (uint16_t)(a >> (8 * done));
          __builtin_memcpy(q + done, &w, 2);
          done += 2;
        }
        if (n & 1) q[done] = (uint8_t)(a >> (8 * done));
      }
    }
    p += n;
  }
  __device__ __forceinline__ void bytes_twice(const uint8_t* s, uint32_t n, uint32_t dist) { copy<false, true>(s, n, dist); }
  __device__ __forceinline__ void skip(uint32_t n) { p += n; }
  __device__ __forceinline__ void bytes(const uint8_t* s, uint32_t n) { copy<false>(s, n); }
  __device__ __forceinline__ void name(const uint8_t* s, uint32_t n) { copy<true>(s, n); }
  __device__ __forceinline__ void ch(char c) {
    p[0] = (uint8_t)c;
    p += 1;
  }
  __device__ __forceinline__ void dec(unsigned long v) {
    const uint32_t d = dec_digits(v);
    if (!(v >> 32)) {
      uint32_t x = (uint32_t)v;
      for (uint32_t i = 0; i < d; ++i) {
        const uint32_t q = x / 10u;
        p[d - 1 - i] = (uint8_t)('0' + (x - q * 10u));
        x = q;
      }
    } else {
      for (uint32_t i = 0; i < d; ++i) {
        p[d - 1 - i] = (uint8_t)('0' + v % 10);
        v /= 10;
      }
    }
    p += d;
  }
};

template <class W>
__device__ __forceinline__ void bc_put_cut(W& w, const uint8_t* s, const Cut& c) {
  w.bytes(s + c.from, c.n);
  if (c.add_nl) w.ch('\n');
}

// the FASTQ record of one file (src/fastq_pre_barcodes.c:713-718)
template <class W>
__device__ __forceinline__ void bc_emit_fastq(bool sliced, long off, long size, const BcLine (&ln)[4], const BcTags& t, W& w) {
  const bool tagged = (t.n[0] | t.n[1] | t.n[2]) != 0;
  // add_tags2readname (src/fastq_pre_barcodes.c:192-216) moves the header up by the tags' length inside hdr1[1000] -
  // and, when the tagged header has 1000 characters or more (a header line near the gzgets limit, or a piece of a longer
  // one), on into hdr2[], the next member of the struct (src/fastq.h:98-101), whose bytes [1] and [2] are then set to
  // '\n' and 0: what is printed as hdr1 runs through hdr2[0] and ends with that '\n', what is printed as hdr2 is the
  // header's character number 1000 and the '\n' (exactly 1000 characters: the string's 0 lands in hdr2[0] - hdr2 is empty)
  const uint32_t tags_n = 31u + t.n[0] + t.n[1] + t.n[2];
  const uint32_t hn = ln[0].len + ln[0].nl + (tagged ? tags_n : 0u);
  const bool spills = tagged && hn >= (uint32_t)FQG_MAX_LABEL_LENGTH;
  if (tagged) {
    w.ch((char)ln[0].p[0]);
    BC_LIT(w, "STAGS_CELL=");
    w.bytes(t.s[1], t.n[1]);
    BC_LIT(w, "_UMI=");
    w.bytes(t.s[0], t.n[0]);
    BC_LIT(w, "_SAMPLE=");
    w.bytes(t.s[2], t.n[2]);
    BC_LIT(w, "_ETAGS_");
    const uint32_t rest = ln[0].len + ln[0].nl - 1;
    if (spills && hn > (uint32_t)FQG_MAX_LABEL_LENGTH) {
      w.bytes(ln[0].p + 1, (uint32_t)FQG_MAX_LABEL_LENGTH - tags_n);  // ... up to character number 1000 of the tagged header
      w.ch('\n');
    } else w.bytes(ln[0].p + 1, rest);
  } else {
    w.bytes(ln[0].p, ln[0].len + ln[0].nl);
  }
  if (sliced) bc_put_cut(w, ln[1].p, bc_cut(ln[1].len + ln[1].nl, off, size));
  else w.bytes(ln[1].p, ln[1].len + ln[1].nl);
  if (spills) {
    if (hn > (uint32_t)FQG_MAX_LABEL_LENGTH) {
      w.ch((char)ln[0].p[(uint32_t)FQG_MAX_LABEL_LENGTH - tags_n]);
      w.ch('\n');
    }
  } else if (tagged || sliced) {
    w.ch((char)ln[2].p[0]);
    w.ch('\n');
  } else {
    w.bytes(ln[2].p, ln[2].len + ln[2].nl);
  }
  if (sliced) bc_put_cut(w, ln[3].p, bc_cut(ln[3].len + ln[3].nl, off, size));
  else w.bytes(ln[3].p, ln[3].len + ln[3].nl);
}

// one SAM line (src/fastq_pre_barcodes.c:666-711)
template <class W>
__device__ __forceinline__ void bc_emit_sam_line(const BcParams& P, unsigned long number, bool se, bool mate1,
                                                 const BcLine (&ln)[4], const SamGeom& g, const BcTags& t, W& w) {
  w.dec(number);
  w.ch('\t');
  w.dec(bc_sam_flag(se, mate1));
  BC_LIT(w, "\t*\t0\t255\t*\t*\t0\t");
  w.dec(g.shown);
  w.ch('\t');
  w.bytes(ln[1].p + g.cs.from, g.seq_n);
  w.ch('\t');
  w.bytes_twice(ln[3].p + g.cq.from, g.qual_n, g.qual_n + 6 + g.name_n + 6);  // the QUAL column and the op:Z: value
  BC_LIT(w, "\ton:Z:");
  w.name(ln[0].p + 1, g.name_n);
  BC_LIT(w, "\top:Z:");
  w.skip(g.qual_n);
  if (t.n[0]) {
    if (P.tenx) BC_LIT(w, "\tUB:Z:"); else BC_LIT(w, "\tRX:Z:");
    w.bytes(t.s[0], t.n[0]);
    if (P.tenx) BC_LIT(w, "\tUY:Z:"); else BC_LIT(w, "\tQX:Z:");
    w.bytes(t.q[0], t.qn[0]);
  }
  if (t.n[1]) {
    w.ch(mate1 ? '\t' : ' ');  // the second mate gets a blank (src/fastq_pre_barcodes.c:705)
    BC_LIT(w, "CR:Z:");
    w.bytes(t.s[1], t.n[1]);
    BC_LIT(w, "\tCY:Z:");
    w.bytes(t.q[1], t.qn[1]);
  }
  if (t.n[2]) {
    BC_LIT(w, "\tBC:Z:");
    w.bytes(t.s[2], t.n[2]);
    BC_LIT(w, "\tQT:Z:");
    w.bytes(t.q[2], t.qn[2]);
  }
  w.ch('\n');
}

__device__ __forceinline__ void emit_flush(const uint8_t* __restrict__ buf, uint32_t skew, uint32_t len,
                                           uint8_t* __restrict__ dst, int lane) {
  // buf[skew .. skew+len) -> dst[0 .. len), with (dst - skew) 16-byte aligned
  uint8_t* g0 = dst - skew;
  const uint32_t end = skew + len;
  const uint32_t first_full = (skew + 15u) & ~15u, last_full = end & ~15u;
  for (uint32_t i = skew + lane; i < (first_full < end ? first_full : end); i += kWave) g0[i] = buf[i];
  typedef uint32_t fl_u32x4 __attribute__((ext_vector_type(4)));
  uint32_t u = first_full + 16u * lane;
  for (; u + 3u * 16u * kWave + 16u <= last_full; u += 4u * 16u * kWave) {  // four LDS reads in flight per lane
    fl_u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const fl_u32x4*>(buf + u + (uint32_t)q * 16u * kWave);
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(v[q], reinterpret_cast<fl_u32x4*>(g0 + u + (uint32_t)q * 16u * kWave));
  }
  for (; u + 16u <= last_full; u += 16u * kWave)
    __builtin_nontemporal_store(*reinterpret_cast<const fl_u32x4*>(buf + u), reinterpret_cast<fl_u32x4*>(g0 + u));
  if (last_full >= first_full)
    for (uint32_t i = last_full + lane; i < end; i += kWave) g0[i] = buf[i];
}

struct EmitOut {
  const uint32_t* len;
  const unsigned long long* off;
  const unsigned long long* sum;
  uint8_t* out;
};

// One wavefront per tile.  SAM: one lane per line (two per iteration for paired reads); FASTQ: one
// lane per record, one output after the other.
template <bool SAM, int MASK>
__global__ __launch_bounds__(kWave, 2) void k_bc_emit_tile(BcParams P, BcTile tc, uint64_t n_done,
                                                        const uint8_t* __restrict__ status,
                                                        const uint8_t* __restrict__ tile_big, EmitOut o0, EmitOut o1,
                                                        EmitOut o2, BcCall* __restrict__ call) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
  uint8_t* s_in = s_lds;
  uint8_t* s_out = s_lds + tc.in_cap;
  const int lane = (int)threadIdx.x;
  const bool se = !bc_has<MASK>(P, 2);
  const uint32_t lpi = SAM && !se ? 2u : 1u;  // lanes per iteration
  const uint32_t it_raw = lpi == 2 ? (uint32_t)lane >> 1 : (uint32_t)lane;
  const bool mate1 = lpi == 2 ? !(lane & 1) : true;
  const uint64_t n_tiles = (n_done + tc.T - 1) / tc.T;
  auto tile_size = [&](uint64_t tile) {
    const uint64_t left = n_done - tile * tc.T;
    return (uint32_t)(left < (uint64_t)tc.T ? left : (uint64_t)tc.T);
  };
  auto geo_of = [&](uint64_t tile, TileGeo& tg) {
    const uint32_t Tn = tile_size(tile);
    const uint64_t k = tile * tc.T + (it_raw < Tn ? it_raw : Tn - 1);
    tg.big = tile_big[tile];
    tg.st = status[k];
#pragma unroll
    for (int x = 1; x < kBcFiles; ++x)
      if (bc_has<MASK>(P, x)) bc_geo_load(P.f[x], k, tg.f[x]);
    if (SAM) {
      tg.off[0] = o0.off[k], tg.sum[0] = o0.sum[k / kScan64Span], tg.olen[0] = o0.len[k];
    } else {
      if (P.emit[1]) tg.off[1] = o1.off[k], tg.sum[1] = o1.sum[k / kScan64Span], tg.olen[1] = o1.len[k];
      if (P.emit[2]) tg.off[2] = o2.off[k], tg.sum[2] = o2.sum[k / kScan64Span], tg.olen[2] = o2.len[k];
    }
  };
  // Three tiles are under way per wavefront: the current tile is written while the spans of the next one (registers)
  // and the line index of the one after it are in flight.  Every request is made without a branch and nothing is
  // computed from a loaded value before the tile it belongs to begins - a value that is loaded on one path only is
  // copied where the paths meet, and an instruction that reads it is a wait for it: the tile behind the last one is the
  // last one again, a tile that does not fit LDS fetches (and drops) what its clamped plan says.
  // (The kernel for any set of files, MASK 0, has no registers for this: it copies the spans of a tile when it gets
  // there.)
  constexpr bool kAhead = MASK != 0;
  const uint64_t stride = gridDim.x;
  auto clamp_tile = [&](uint64_t t) { return t < n_tiles ? t : n_tiles - 1; };
  auto last_lane_of = [&](uint64_t t) { return (int)((tile_size(t) - 1) * lpi); };
  TileGeo cur, nxt, nx2;
  bc_u32x4 pf[kSpanPf];
  if (blockIdx.x < n_tiles) {
    geo_of(blockIdx.x, cur);
    if (kAhead) {
      geo_of(clamp_tile(blockIdx.x + stride), nxt);
      SpanPlan sp;
      bc_span_plan<false, MASK>(P, cur, last_lane_of(blockIdx.x), tc.in_cap, sp);
      bc_span_fetch(sp, lane, pf);
    }
  }
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += stride, cur = kAhead ? nxt : nx2, nxt = nx2) {
    const uint64_t k0 = tile * tc.T;
    const uint32_t Tn = tile_size(tile);
    const bool valid = it_raw < Tn;
    const uint64_t k = k0 + (valid ? it_raw : Tn - 1);
    const bool big = __builtin_amdgcn_readfirstlane((int)cur.big) != 0;
    BcLine L[kBcFiles][4];
    {
      SpanPlan sp;
      bc_span_plan<false, MASK>(P, cur, last_lane_of(tile), tc.in_cap, sp);
      if (!big) {  // (fits: the plan checked)
        if (P.ablate & 8) {
        } else if (kAhead) bc_span_land(sp, lane, pf, s_in);
        else bc_span_copy(sp, 0, lane, s_in);
      }
      bc_span_lines<false, MASK>(P, cur, sp, s_in, L);
    }
    if (kAhead) {
      const uint64_t tn = clamp_tile(tile + stride);
      SpanPlan sp;
      bc_span_plan<false, MASK>(P, nxt, last_lane_of(tn), tc.in_cap, sp);
      bc_span_fetch(sp, lane, pf);
    }
    geo_of(clamp_tile(tile + (kAhead ? 2 : 1) * stride), nx2);
    if (big) continue;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the name checks of every iteration, kept or not (the reference makes them first, src/fastq_pre_barcodes.c:606-635)
    if (!(P.ablate & 4)) bc_report_finding(call, valid && mate1, k, valid && mate1 ? bc_check_names<true, MASK>(P, L) : 0u);
    const bool keep = valid && cur.st == kBcKeep;
    BcTags t;
    bc_tags_of_kept<MASK>(P, L, &t);
    if (SAM) {
      BcLine own[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        own[l].p = mate1 ? L[1][l].p : L[2][l].p;
        own[l].len = mate1 ? L[1][l].len : L[2][l].len;
        own[l].nl = mate1 ? L[1][l].nl : L[2][l].nl;
      }
      const bool sliced = mate1 ? bc_slices(P, 1) : bc_slices(P, 2);
      const SamGeom g = bc_sam_geom(sliced, mate1 ? P.read_off[1] : P.read_off[2], mate1 ? P.read_size[1] : P.read_size[2],
                                    mate1, own);
      const unsigned long number = P.first_read_number + k + 1;
      // (single-end: the line's length is what the plan computed for the iteration; two mates share that number)
      const uint32_t my_len = !keep ? 0u : (lpi == 1 ? cur.olen[0] : bc_sam_line_len(number, bc_sam_flag(se, mate1), g, t));
      const unsigned long long where = cur.off[0] + cur.sum[0];
      const unsigned long long tile_at = rfl64(where);
      const uint32_t before = __shfl_up(my_len, 1, 64);
      const uint32_t start = (uint32_t)(where - tile_at) + (mate1 ? 0u : before);
      const uint32_t total = wave_max32(start + my_len);
      uint8_t* dst = o0.out + tile_at;
      const uint32_t skew = (uint32_t)((uintptr_t)dst & 15u);
      if (keep && !(P.ablate & 1)) {
        LaneWriter w{s_out + skew + start};
        bc_emit_sam_line(P, number, se, mate1, own, g, t, w);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (!(P.ablate & 2)) emit_flush(s_out, skew, total, dst, lane);
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int which = 1; which <= 2; ++which) {
        if (!P.emit[which]) continue;
        const EmitOut& o = which == 1 ? o1 : o2;
        const bool sliced = bc_slices(P, which);
        const uint32_t my_len = keep ? cur.olen[which] : 0u;
        const unsigned long long where = cur.off[which] + cur.sum[which];
        const unsigned long long tile_at = rfl64(where);
        const uint32_t start = (uint32_t)(where - tile_at);
        const uint32_t total = wave_max32(start + my_len);
        uint8_t* dst = o.out + tile_at;
        const uint32_t skew = (uint32_t)((uintptr_t)dst & 15u);
        if (keep) {
          LaneWriter w{s_out + skew + start};
          bc_emit_fastq(sliced, P.read_off[which], P.read_size[which], L[which], t, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        emit_flush(s_out, skew, total, dst, lane);
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// iterations of the tiles that do not fit LDS (long reads): one wavefront per iteration, records read
// from the images, text written straight to the output image
__global__ __launch_bounds__(kBlock) void k_bc_emit_direct(BcParams P, BcTile tc, uint64_t n_done,
                                                           const uint8_t* __restrict__ status,
                                                           const uint8_t* __restrict__ tile_big, EmitOut o0, EmitOut o1,
                                                           EmitOut o2, BcCall* __restrict__ call) {
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const int lane = (int)(threadIdx.x & 63), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool se = !P.f[2].present;
  for (uint64_t k = (uint64_t)blockIdx.x * (kBlock / kWave) + wv; k < n_done; k += n_waves) {
    if (!tile_big[k / tc.T]) continue;
    BcLine L[kBcFiles][4];
    bc_lines_all(P, k, L);
    {  // the name checks (every lane the same bytes, from the images)
      const uint32_t finding = bc_check_names<false>(P, L);
      if (finding && lane == 0) atomicMin(&call->first_finding, (unsigned long long)((k << 8) | finding));
    }
    if (status[k] != kBcKeep) continue;
    BcTags t;
    bc_tags_of_kept(P, L, &t);
    if (P.out_sam) {
      Writer w{o0.out + o0.off[k] + o0.sum[k / kScan64Span], lane};
      bc_emit_sam_line(P, P.first_read_number + k + 1, se, true, L[1],
                       bc_sam_geom(bc_slices(P, 1), P.read_off[1], P.read_size[1], true, L[1]), t, w);
      if (!se)
        bc_emit_sam_line(P, P.first_read_number + k + 1, se, false, L[2],
                         bc_sam_geom(bc_slices(P, 2), P.read_off[2], P.read_size[2], false, L[2]), t, w);
    } else {
      if (P.emit[1]) {
        Writer w{o1.out + o1.off[k] + o1.sum[k / kScan64Span], lane};
        bc_emit_fastq(bc_slices(P, 1), P.read_off[1], P.read_size[1], L[1], t, w);
      }
      if (P.emit[2]) {
        Writer w{o2.out + o2.off[k] + o2.sum[k / kScan64Span], lane};
        bc_emit_fastq(bc_slices(P, 2), P.read_off[2], P.read_size[2], L[2], t, w);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// whitelist membership of a barcode cut out of the reads (BASELINE configs[2]: "known_cells whitelist"): the cell
// barcode that fastq_pre_barcodes puts into the read name is what bam_umi_count later packs with char2uint_64
// (src/bam_umi_count.c:364-382: base 10, A C G T N -> 1..5, from the END of the string, stopping at the first other
// character) and looks up with valid_barcode (:523-535) in the table load_whitelist (:543-579) filled.  One thread
// per record: the `size` characters at `offset` of the sequence line (get_barcode's bounds, src/fastq_pre_barcodes.c:232:
// a read too short for them has no barcode), packed in registers, looked up in a small open-addressing set that the
// L2 keeps (ten thousand entries).
// ------------------------------------------------------------------------------------------
struct WlSlot {
  unsigned long long key, used;
};
struct WlCall {
  unsigned long long n_valid, n_short;
};
__host__ __device__ inline uint64_t wl_hash(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}
constexpr int kWlMaxSize = 56;  // barcode characters (the reference's arrays hold 49)

template <int WORDS>  // 8-byte words that hold the barcode
__global__ __launch_bounds__(kBlock) void k_bc_whitelist(FrameView f, uint64_t first, uint64_t step, uint64_t n, uint32_t offset,
                                                         uint32_t size, const WlSlot* __restrict__ set, uint64_t mask,
                                                         uint8_t* __restrict__ valid, WlCall* __restrict__ call) {
  unsigned long long n_valid = 0, n_short = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
    const uint64_t r = first + k * step;
    const uint64_t b = f.line_end[4 * r] + 1, e = f.line_end[4 * r + 1];  // the sequence line
    bool ok = false;
    if ((uint64_t)offset + size > e - b) ++n_short;  // (offset + size > read_len - 1, read_len = strlen(seq) with its '\n')
    else {
      const uint8_t* p = f.img + b + offset;
      uint64_t w[WORDS];
#pragma unroll
      for (int i = 0; i < WORDS; ++i) {
        w[i] = 0;
        // (whole words while they lie inside the image; the last bytes of an image one by one)
        if (8u * i < size) {
          if (b + offset + 8u * i + 8u <= f.nbytes) __builtin_memcpy(&w[i], p + 8 * i, 8);
          else
            for (uint32_t q = 0; q < 8 && 8u * i + q < size; ++q) w[i] |= (uint64_t)p[8 * i + q] << (8 * q);
        }
      }
      uint64_t v = 0;
      bool stopped = false;
#pragma unroll
      for (int pos = 8 * WORDS - 1; pos >= 0; --pos) {
        const uint32_t c = (uint32_t)(w[pos >> 3] >> (8 * (pos & 7))) & 0xFFu;
        // A C G T N (either case) -> 1 2 3 4 5: c & 0xDF folds the case
        const uint32_t u = c & 0xDFu;
        const uint32_t base = u == 'A' ? 1u : u == 'C' ? 2u : u == 'G' ? 3u : u == 'T' ? 4u : u == 'N' ? 5u : 0u;
        if ((uint32_t)pos < size && !stopped) {
          if (!base) stopped = true;
          else v = v * 10 + base;
        }
      }
      for (uint64_t at = wl_hash(v) & mask;; at = (at + 1) & mask) {
        const WlSlot s = set[at];
        if (!s.used) break;
        if (s.key == v) {
          ok = true;
          break;
        }
      }
    }
    if (valid) valid[k] = ok ? 1 : 0;
    n_valid += ok ? 1 : 0;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    n_valid += __shfl_down(n_valid, d, 64);
    n_short += __shfl_down(n_short, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_valid) atomicAdd(&call->n_valid, n_valid);
    if (n_short) atomicAdd(&call->n_short, n_short);
  }
}

}  // namespace fqg
