// fq_pgzip.h - a gzip file inflated on many cores (the boundary work of SURVEY 8f-2; reference: fastq_open's
// gzopen + one gzgets per line on one thread, src/fastq.c:631-663, :245-261).
//
// A gzip member is ONE deflate stream: a block may point up to 32 KiB back into whatever came before it, and nothing
// in the file says where blocks start.  The inflated bytes of the files people have (gzip, pigz, bcl2fastq output:
// one member each) therefore come from one zlib thread at 0.3 GB/s however many cores the host has, and the GPU
// waits.  This reader cuts the compressed bytes into chunks of a few MiB and lets every thread
//   1. FIND the first deflate block that starts in its chunk: every bit position is tried as the header of a
//      dynamic-Huffman block (code-length code complete, literal/length and distance codes complete, the block
//      and its successors decode);
//   2. INFLATE from there without knowing the 32 KiB in front of it: the output is 16-bit symbols, a byte or
//      "byte i of the unknown window" - copies out of the window copy the markers along;
//   3. stop at the first block boundary at or behind the next chunk's first bit.
// Then, in file order: a chunk is JOINED to its predecessor only if the predecessor's decoding ended at exactly the
// bit where the chunk started - so a wrong guess in step 1 can cost time (the predecessor decodes on through that
// chunk) but never a byte; the last 32 KiB of each chunk, resolved with its predecessor's, are the next chunk's
// window; and all chunks replace their markers and narrow to bytes in parallel, each summing the CRC-32 of its
// part (crc32_combine glues the parts: every member's CRC-32 and length are checked as zlib checks them).
// (The scheme is that of pugz - Kerbiriou & Chikhi, 2019 - restated; no code of theirs was at hand.)
//
// Whatever this decoder does not want to decide - a corrupt or truncated stream, a block larger than the window of
// compressed bytes, an over-subscribed code, chunks whose first blocks are not found - goes to zlib: the reader falls
// back to ONE raw zlib stream primed with the bit position and the 32 KiB in front of it (inflatePrime /
// inflateSetDictionary) and stays there, so errors are zlib's errors with zlib's texts, and any gzip file zlib
// reads is read the way zlib reads it (a file that ends inside a member: its bytes up to there, then the end of the
// data, as gzread does it).  Multi-member files, members that end inside a chunk and bytes behind the last member (zlib's
// gzread ignores them) are handled in the parallel path.
#pragma once
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "fq_parallel.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace fqhost {
namespace pgz {

// CRC-32 (the gzip polynomial, reflected: 0xEDB88320) sixteen bytes per step from sixteen tables - zlib 1.2.11's
// crc32 goes four bytes per step (1 GB/s per core where this gives 3), and every inflated byte passes through it.
// Same function as zlib's: crc32(crc, p, n) == crc32_16(crc, p, n) (tests/cxx/pgzip_check.cpp compares them).
struct Crc32Tables {
  uint32_t t[16][256];
  Crc32Tables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 16; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFFu];
  }
};
inline uint32_t crc32_16(uint32_t crc, const uint8_t* p, size_t n) {
  static const Crc32Tables T;
  crc = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) {
    crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 0xFFu];
    --n;
  }
  while (n >= 16) {
    uint64_t a, b;
    memcpy(&a, p, 8);
    memcpy(&b, p + 8, 8);
    a ^= crc;
    crc = T.t[15][a & 0xFFu] ^ T.t[14][(a >> 8) & 0xFFu] ^ T.t[13][(a >> 16) & 0xFFu] ^ T.t[12][(a >> 24) & 0xFFu] ^
          T.t[11][(a >> 32) & 0xFFu] ^ T.t[10][(a >> 40) & 0xFFu] ^ T.t[9][(a >> 48) & 0xFFu] ^ T.t[8][a >> 56] ^
          T.t[7][b & 0xFFu] ^ T.t[6][(b >> 8) & 0xFFu] ^ T.t[5][(b >> 16) & 0xFFu] ^ T.t[4][(b >> 24) & 0xFFu] ^
          T.t[3][(b >> 32) & 0xFFu] ^ T.t[2][(b >> 40) & 0xFFu] ^ T.t[1][(b >> 48) & 0xFFu] ^ T.t[0][b >> 56];
    p += 16;
    n -= 16;
  }
  while (n--) crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 0xFFu];
  return ~crc;
}

constexpr uint32_t kWin = 32768;
// readable (zero) bytes behind the valid compressed bytes: the header of a dynamic block is read without a look at the
// end of the input (at most 14 + 57 + 316 * 14 bits = 562 bytes), the symbol loop looks once per symbol
constexpr size_t kSlack = 1024;

// ---- entries of the decoding tables ------------------------------------------------------------------------------
// bits 0-4 code bits to drop | 5-7 type | 8-12 extra bits (sub-table: its index bits) | 16-31 literal / base / sub-table start
enum : uint32_t { kLit = 0, kLen = 1, kEob = 2, kSub = 3, kBadSym = 4 };
constexpr uint32_t entry(uint32_t type, uint32_t val, uint32_t extra = 0) { return (val << 16) | (extra << 8) | (type << 5); }
constexpr uint32_t e_type(uint32_t e) { return (e >> 5) & 7u; }
constexpr uint32_t e_bits(uint32_t e) { return e & 31u; }
constexpr uint32_t e_extra(uint32_t e) { return (e >> 8) & 31u; }
constexpr uint32_t e_val(uint32_t e) { return e >> 16; }

constexpr unsigned kLitBits = 11, kDistBits = 8, kMaxBits = 15;
constexpr unsigned kLitTable = (1u << kLitBits) + 288u * (1u << (kMaxBits - kLitBits));
constexpr unsigned kDistTable = (1u << kDistBits) + 32u * (1u << (kMaxBits - kDistBits));

enum CodeKind { kComplete = 0, kEmpty, kSingle, kIncomplete, kOver };

inline uint32_t reverse_bits(uint32_t c, unsigned n) {
  uint32_t r = 0;
  for (unsigned i = 0; i < n; ++i) {
    r = (r << 1) | (c & 1u);
    c >>= 1;
  }
  return r;
}

// canonical Huffman code of lens[0, n) (RFC 1951 3.2.2) as a two-level table
inline CodeKind build_table(const uint8_t* lens, unsigned n, const uint32_t* sym_entry, unsigned pbits, unsigned maxbits,
                            uint32_t* tab, unsigned cap) {
  unsigned count[16] = {0};
  for (unsigned s = 0; s < n; ++s) ++count[lens[s]];
  count[0] = 0;
  unsigned total = 0;
  for (unsigned b = 1; b <= maxbits; ++b) total += count[b];
  for (unsigned i = 0; i < (1u << pbits); ++i) tab[i] = entry(kBadSym, 0);
  if (!total) return kEmpty;
  long left = 1;
  for (unsigned b = 1; b <= maxbits; ++b) {
    left <<= 1;
    left -= (long)count[b];
    if (left < 0) return kOver;
  }
  CodeKind kind = kComplete;
  if (left > 0) {
    if (total == 1 && count[1] == 1) kind = kSingle;
    else return kIncomplete;
  }
  unsigned next[16];
  unsigned code = 0;
  for (unsigned b = 1; b <= maxbits; ++b) {
    code = (code + count[b - 1]) << 1;
    next[b] = code;
  }
  unsigned sub_next = 1u << pbits;
  const unsigned sub_bits = maxbits - pbits, sub_size = 1u << sub_bits;
  for (unsigned s = 0; s < n; ++s) {
    const unsigned L = lens[s];
    if (!L) continue;
    const uint32_t r = reverse_bits(next[L]++, L), e = sym_entry[s];
    if (L <= pbits) {
      for (uint32_t i = r; i < (1u << pbits); i += 1u << L) tab[i] = e | L;
    } else {
      const uint32_t prefix = r & ((1u << pbits) - 1u);
      if (e_type(tab[prefix]) != kSub) {
        if (sub_next + sub_size > cap) return kOver;  // (cannot be: cap allows a sub-table per symbol)
        for (unsigned i = 0; i < sub_size; ++i) tab[sub_next + i] = entry(kBadSym, 0);
        tab[prefix] = entry(kSub, sub_next, sub_bits) | pbits;
        sub_next += sub_size;
      }
      const uint32_t start = e_val(tab[prefix]), hi = r >> pbits, Ls = L - pbits;
      for (uint32_t i = hi; i < sub_size; i += 1u << Ls) tab[start + i] = e | Ls;
    }
  }
  return kind;
}

struct SymEntries {
  uint32_t lit[288], dist[32], cl[19];
  SymEntries() {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (unsigned s = 0; s < 256; ++s) lit[s] = entry(kLit, s);
    lit[256] = entry(kEob, 0);
    for (unsigned s = 257; s < 286; ++s) lit[s] = entry(kLen, lbase[s - 257], lext[s - 257]);
    lit[286] = lit[287] = entry(kBadSym, 0);
    for (unsigned s = 0; s < 30; ++s) dist[s] = entry(kLen, dbase[s], dext[s]);
    dist[30] = dist[31] = entry(kBadSym, 0);
    for (unsigned s = 0; s < 19; ++s) cl[s] = entry(kLit, s);
  }
};
inline const SymEntries& sym_entries() {
  static const SymEntries s;
  return s;
}

// ---- bits --------------------------------------------------------------------------------------------------------
struct BitIn {
  const uint8_t* base = nullptr;
  const uint8_t* p = nullptr;
  uint64_t bb = 0;
  unsigned bc = 0;
  void refill() {  // 56..63 bits afterwards (the bytes behind the valid input are readable zeros)
    uint64_t w;
    memcpy(&w, p, 8);
    bb |= w << bc;
    p += (63 - bc) >> 3;
    bc |= 56;
  }
  void seek(const uint8_t* b, uint64_t bit) {
    base = b;
    p = b + (bit >> 3);
    bb = 0;
    bc = 0;
    refill();
    const unsigned r = (unsigned)(bit & 7);
    bb >>= r;
    bc -= r;
  }
  uint32_t peek(unsigned n) const { return (uint32_t)(bb & ((1ull << n) - 1ull)); }
  void drop(unsigned n) {
    bb >>= n;
    bc -= n;
  }
  uint32_t take(unsigned n) {
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  uint64_t bitpos() const { return (uint64_t)(p - base) * 8u - bc; }
};

// ---- 16-bit output -------------------------------------------------------------------------------------------------
struct Marker {
  uint32_t at;   // this byte of the chunk's output ...
  uint16_t idx;  // ... is byte idx of the 32 KiB in front of the chunk
};

// d is a SLIDING buffer of symbols: d[0, kWin) the 32 KiB in front of d[kWin] (at first the window in front of the chunk,
// markers 0x8000 | i where it is unknown), d[kWin, pos) what has been decoded since the last slide.  After every block the
// new symbols are narrowed to `bytes` while they are in the cache - a marker becomes a zero byte and a note of where it
// was and which one - and when the buffer is full its last 32 KiB move to the front.  The symbols of a chunk thus never
// travel to memory and back (16 MB per 4 MiB chunk did), and a chunk holds one byte per byte of output instead of two.
// (Measured side by side on the box - 16 hardware threads of a busy EPYC - the two designs take the same 0.8 s for 3.2 GB:
// 10 - 11 thread-seconds of work either way where one thread alone needs 6.7; sibling threads share their cores.)
struct Out16 {
  // symbols between slides (a stored block's 65 535 must fit): 200 KiB with the window, a fifth of a core's L2
  // (66 Ki .. 2 Mi symbols: no difference on the box - FQGPU_PGZIP_SPAN, tools/pgzip_scan.sh)
  static size_t span() {
    static const size_t v = [] {
      const char* e = getenv("FQGPU_PGZIP_SPAN");
      const long k = e ? atol(e) : 0;
      return (size_t)(k >= 66 && k <= 4096 ? k : 68) << 10;
    }();
    return v;
  }
  uint16_t* d = nullptr;
  size_t pos = kWin, flushed = kWin;  // d[kWin, flushed) are in `bytes` already
  uint64_t base = 0;                  // output bytes that lie in front of d[kWin]
  uint8_t* bytes = nullptr;           // the chunk's output, markers as zero bytes
  size_t bytes_cap = 0, nbytes = 0;
  std::vector<Marker> marks;          // in order of `at`
  Out16() = default;
  Out16(const Out16&) = delete;
  Out16& operator=(const Out16&) = delete;
  ~Out16() {
    free(d);
    free(bytes);
  }
  static size_t room() { return kWin + span(); }  // a decoder asks for a slide when pos + 300 goes beyond it
  bool init(size_t bytes_hint) {
    if (!d) d = static_cast<uint16_t*>(malloc((room() + 1024) * sizeof(uint16_t)));
    if (d && bytes_cap < bytes_hint) {
      uint8_t* nb = static_cast<uint8_t*>(realloc(bytes, bytes_hint + 64));
      if (nb) {
        bytes = nb;
        bytes_cap = bytes_hint;
      }
    }
    return d != nullptr && bytes != nullptr;
  }
  void reset() {
    pos = flushed = kWin;
    base = 0;
    nbytes = 0;
    marks.clear();
  }
  uint64_t produced() const { return base + (pos - kWin); }
  void window_unknown() {
    for (uint32_t i = 0; i < kWin; ++i) d[i] = (uint16_t)(0x8000u | i);
  }
  void window_known(const uint8_t* w, size_t n) {  // the last n <= kWin bytes in front; what lies before them must not be asked for
    for (uint32_t i = 0; i < kWin - n; ++i) d[i] = 0;
    for (size_t i = 0; i < n; ++i) d[kWin - n + i] = w[i];
  }
  // d[flushed, pos) -> bytes; false: out of memory, or more than `limit` bytes of output
  bool flush(size_t limit) {
    const size_t n = pos - flushed;
    if (!n) return true;
    if (nbytes + n > limit) return false;
    if (nbytes + n > bytes_cap) {
      const size_t nc = std::max<size_t>(bytes_cap + bytes_cap / 2, nbytes + n + (1u << 20));
      uint8_t* nb = static_cast<uint8_t*>(realloc(bytes, nc + 64));
      if (!nb) return false;
      bytes = nb;
      bytes_cap = nc;
    }
    const uint16_t* s = d + flushed;
    uint8_t* o = bytes + nbytes;
    size_t i = 0;
    auto one = [&](size_t k) {
      const uint16_t v = s[k];
      if (v < 256) o[k] = (uint8_t)v;
      else {
        o[k] = 0;
        marks.push_back(Marker{(uint32_t)(nbytes + k), (uint16_t)(v & 0x7FFFu)});
      }
    };
#if defined(__SSE2__)
    for (; i + 16 <= n; i += 16) {  // sixteen symbols: bytes as they are when none of them is a marker
      const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
      const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 8));
      if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_srli_epi16(_mm_or_si128(a, b), 8), _mm_setzero_si128())) == 0xFFFF) {
        _mm_storeu_si128(reinterpret_cast<__m128i*>(o + i), _mm_packus_epi16(a, b));
      } else {
        for (size_t k = i; k < i + 16; ++k) one(k);
      }
    }
#endif
    for (; i < n; ++i) one(i);
    nbytes += n;
    flushed = pos;
    return true;
  }
  void slide() {  // (everything flushed) the last 32 KiB of symbols to the front
    const size_t shift = pos - kWin;
    memmove(d, d + shift, kWin * sizeof(uint16_t));
    base += shift;
    pos = flushed = kWin;
  }
};

struct MemberEnd {
  size_t out_pos;  // (byte of the chunk's output) the member's last byte is out_pos - 1
  uint32_t crc, isize;
};

// gzip member header at p (RFC 1952 2.3): its length; 0 = more bytes needed; -1 = no gzip magic (or, AT THE END OF
// THE FILE, fewer than two bytes: what zlib's gzread takes for trailing garbage); -2 = magic, but a method or flags
// that zlib refuses.  eof = the n bytes are all the file has: only then do 0 or 1 bytes decide anything - in the
// middle of a file they are where a window ended, and the next member may start right there.
inline long gzip_header_len(const uint8_t* p, size_t n, bool eof) {
  if (n < 2 && !eof && (n == 0 || p[0] == 0x1f)) return 0;
  if (n < 2 || p[0] != 0x1f || p[1] != 0x8b) return -1;
  if (n < 10) return 0;
  if (p[2] != 8 || (p[3] & 0xe0)) return -2;
  const unsigned flg = p[3];
  size_t q = 10;
  if (flg & 4) {
    if (q + 2 > n) return 0;
    q += 2 + (p[q] | ((size_t)p[q + 1] << 8));
    if (q > n) return 0;
  }
  for (unsigned f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
    if (flg & f) {
      while (q < n && p[q]) ++q;
      if (q >= n) return 0;
      ++q;
    }
  if (flg & 2) q += 2;  // FHCRC (zlib checks it; a wrong one is an error there - here the member's CRC-32 still guards the data)
  if (q > n) return 0;
  return (long)q;
}

enum Status { kAtBoundary, kFinished, kNeedInput, kBad, kTooBig };

// what one decoder has reached: always a point between two blocks (or behind a member's trailer and the next header)
struct ChunkState {
  uint64_t bit = 0;   // in the window of compressed bytes
  Out16 out;
  size_t safe_bytes = 0, safe_marks = 0;  // of out.bytes / out.marks: the output up to `bit` (more may have been written)
  long long member_floor = -1;  // byte of the chunk's output at which the current member began; -1: in front of the chunk
  size_t min_src = kWin;        // lowest index of d[0, kWin) a copy ever read (kWin: none)
  size_t min_src_member0 = kWin;  // ... while the chunk's first member lasted (later members must not reach back at all)
  std::vector<MemberEnd> ends;
  bool finished = false;  // behind the last member: end of file, or bytes that are no gzip header
  uint64_t blocks = 0;
  uint64_t start_bit = 0;
  bool found = false;
  Status status = kAtBoundary;
  void reset() {  // (the buffers stay)
    bit = start_bit = blocks = 0;
    out.reset();
    safe_bytes = safe_marks = 0;
    member_floor = -1;
    min_src = min_src_member0 = kWin;
    ends.clear();
    finished = found = false;
    status = kAtBoundary;
  }
};

class Inflate16 {
 public:
  Inflate16() : lit_(kLitTable), dist_(kDistTable), fixed_lit_(kLitTable), fixed_dist_(kDistTable) {
    uint8_t l[288];
    for (unsigned s = 0; s < 144; ++s) l[s] = 8;
    for (unsigned s = 144; s < 256; ++s) l[s] = 9;
    for (unsigned s = 256; s < 280; ++s) l[s] = 7;
    for (unsigned s = 280; s < 288; ++s) l[s] = 8;
    build_table(l, 288, sym_entries().lit, kLitBits, kMaxBits, fixed_lit_.data(), kLitTable);
    for (unsigned s = 0; s < 32; ++s) l[s] = 5;
    build_table(l, 32, sym_entries().dist, kDistBits, kMaxBits, fixed_dist_.data(), kDistTable);
  }

  // Header of a dynamic block behind its three type bits (RFC 1951 3.2.7) -> lit_ / dist_.  false: not a header this
  // decoder takes (zlib may still: an incomplete literal/length code with a single symbol).
  bool dynamic_header(BitIn& in) {
    in.refill();
    const unsigned hlit = in.take(5) + 257, hdist = in.take(5) + 1, hclen = in.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    in.refill();  // (>= 56 bits: 14 are gone, 42 left = 14 lengths; the rest behind another refill)
    for (unsigned i = 0; i < hclen; ++i) {
      if (i == 14) in.refill();
      cl[order[i]] = (uint8_t)in.take(3);
    }
    uint32_t clt[128];
    if (build_table(cl, 19, sym_entries().cl, 7, 7, clt, 128) != kComplete) return false;
    uint8_t lens[320];
    unsigned i = 0;
    const unsigned total = hlit + hdist;
    while (i < total) {
      in.refill();
      const uint32_t e = clt[in.peek(7)];
      if (e_type(e) != kLit) return false;
      in.drop(e_bits(e));
      const unsigned s = e_val(e);
      if (s < 16) lens[i++] = (uint8_t)s;
      else {
        unsigned rep, v = 0;
        if (s == 16) {
          if (!i) return false;
          v = lens[i - 1];
          rep = 3 + in.take(2);
        } else if (s == 17) rep = 3 + in.take(3);
        else rep = 11 + in.take(7);
        if (i + rep > total) return false;
        while (rep--) lens[i++] = (uint8_t)v;
      }
    }
    if (!lens[256]) return false;
    if (build_table(lens, hlit, sym_entries().lit, kLitBits, kMaxBits, lit_.data(), kLitTable) != kComplete) return false;
    const CodeKind dk = build_table(lens + hlit, hdist, sym_entries().dist, kDistBits, kMaxBits, dist_.data(), kDistTable);
    return dk == kComplete || dk == kEmpty || dk == kSingle;
  }

  // Decodes on from cs (a point between blocks) until the first such point at or behind stop_bit, the end of the last
  // member, or trouble; cs always ends at the last point reached.  base[0, nvalid) are the compressed bytes at hand
  // (kSlack readable bytes behind them), eof = the file ends with them; out_limit bounds the bytes of output.
  Status run(ChunkState& cs, const uint8_t* base, size_t nvalid, bool eof, uint64_t stop_bit, size_t out_limit) {
    BitIn in;
    in.seek(base, cs.bit);
    const uint64_t nbits = (uint64_t)nvalid * 8u;
    long long floor_ = cs.member_floor;
    size_t min_src = cs.min_src;
    // leaving with trouble: what was written behind the last point reached does not count (the symbol buffer may have
    // slid past that point - such a chunk is not decoded on)
    auto leave = [&](Status st) {
      cs.out.nbytes = cs.safe_bytes;
      cs.out.marks.resize(cs.safe_marks);
      return cs.status = st;
    };
    for (;;) {
      if (cs.bit >= stop_bit) return cs.status = kAtBoundary;
      if (in.bitpos() + 3 > nbits) return leave(kNeedInput);
      in.refill();
      const unsigned bfinal = in.take(1), btype = in.take(2);
      Status st;
      if (btype == 0) st = stored(in, cs.out, nvalid, out_limit);
      else if (btype == 3) st = kBad;
      else {
        const uint32_t *lt = fixed_lit_.data(), *dt = fixed_dist_.data();
        if (btype == 2) {
          if (!dynamic_header(in)) return leave(in.bitpos() > nbits ? kNeedInput : kBad);
          lt = lit_.data();
          dt = dist_.data();
        }
        st = codes(in, lt, dt, cs.out, floor_, min_src, base + nvalid + 8, out_limit);
      }
      if (st != kAtBoundary) return leave(st);
      if (in.bitpos() > nbits) return leave(kNeedInput);
      if (!cs.out.flush(out_limit)) return leave(kTooBig);  // (the block's symbols to bytes while they are in the cache)
      if (!bfinal) {
        cs.bit = in.bitpos();
        cs.safe_bytes = cs.out.nbytes;
        cs.safe_marks = cs.out.marks.size();
        cs.min_src = min_src;
        ++cs.blocks;
        continue;
      }
      // the member's trailer, and what follows it
      const size_t at = (size_t)((in.bitpos() + 7) >> 3);
      if (at + 8 > nvalid) return leave(kNeedInput);
      auto le32 = [&](size_t o) { return (uint32_t)base[o] | ((uint32_t)base[o + 1] << 8) | ((uint32_t)base[o + 2] << 16) | ((uint32_t)base[o + 3] << 24); };
      const MemberEnd me{cs.out.nbytes, le32(at), le32(at + 4)};
      const size_t next = at + 8;
      long hl = -1;
      if (!(next == nvalid && eof)) {
        hl = gzip_header_len(base + next, nvalid - next, eof);
        if (hl == 0) return leave(eof ? kBad : kNeedInput);  // (a header the file ends in: zlib's "unexpected end of file")
        if (hl == -2) return leave(kBad);
        if (hl > 0 && next + (size_t)hl >= nvalid && !eof) return leave(kNeedInput);
      }
      cs.ends.push_back(me);
      cs.safe_bytes = cs.out.nbytes;
      cs.safe_marks = cs.out.marks.size();
      if (cs.member_floor < 0 && cs.ends.size() == 1) cs.min_src_member0 = min_src;
      cs.min_src = min_src;
      ++cs.blocks;
      if (hl < 0) {
        cs.bit = (uint64_t)next * 8u;
        cs.finished = true;
        return cs.status = kFinished;
      }
      cs.bit = (uint64_t)(next + (size_t)hl) * 8u;
      cs.member_floor = floor_ = (long long)cs.out.nbytes;
      in.seek(base, cs.bit);
    }
  }

 private:
  static Status stored(BitIn& in, Out16& out, size_t nvalid, size_t out_limit) {
    const size_t at = (size_t)((in.bitpos() + 7) >> 3);
    if (at + 4 > nvalid) return kNeedInput;
    const uint8_t* b = in.base + at;
    const unsigned len = b[0] | ((unsigned)b[1] << 8), nlen = b[2] | ((unsigned)b[3] << 8);
    if ((len ^ 0xFFFFu) != nlen) return kBad;
    if (at + 4 + len > nvalid) return kNeedInput;
    if (out.pos + len + 300 > Out16::room()) {
      if (!out.flush(out_limit)) return kTooBig;
      out.slide();
    }
    for (unsigned i = 0; i < len; ++i) out.d[out.pos + i] = b[4 + i];
    out.pos += len;
    in.seek(in.base, (uint64_t)(at + 4 + len) * 8u);
    return kAtBoundary;
  }

  // (two builds of the inner loop, chosen when the program starts: with BMI2 - shifts and masks by a register in one
  // instruction each - it runs a quarter to a third faster, and every x86-64 of the last ten years has it)
  // (not in sanitizer builds: the resolver that chooses runs before their run times are up)
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(__SANITIZE_THREAD__) && !defined(__SANITIZE_ADDRESS__)
  __attribute__((target_clones("default", "bmi2")))
#endif
  static Status codes(BitIn& in, const uint32_t* lt, const uint32_t* dt, Out16& out, long long floor_, size_t& min_src_io,
                      const uint8_t* in_limit, size_t out_limit) {
    size_t pos = out.pos, min_src = min_src_io;
    uint16_t* const d = out.d;
    const size_t room = Out16::room();
    // Where a copy may not reach, in the buffer's coordinates: in front of the member's first byte when the member began
    // in this chunk (zlib's "invalid distance too far back"); and where it reaches into the window in front of the chunk
    // (only until the first slide) the lowest byte asked for is noted - the predecessor must have that many.
    size_t bad_below = 0, note_below = 0;
    auto limits = [&] {
      if (floor_ >= 0) {
        const long long f = floor_ - (long long)out.base + (long long)kWin;
        bad_below = f > 0 ? (size_t)f : 0;
        note_below = 0;
      } else {
        bad_below = 0;
        note_below = out.base == 0 ? kWin : 0;
      }
    };
    limits();
    auto lookup = [&](const uint32_t* t, unsigned pbits) {
      uint32_t e = t[in.bb & ((1u << pbits) - 1u)];
      if (__builtin_expect(e_type(e) == kSub, 0)) {
        in.drop(e_bits(e));
        e = t[e_val(e) + (in.bb & ((1u << e_extra(e)) - 1u))];
      }
      in.drop(e_bits(e));
      return e;
    };
    for (;;) {
      if (__builtin_expect(pos + 300 > room, 0)) {  // the buffer is full: what it holds to bytes, its last 32 KiB to the front
        out.pos = pos;
        if (!out.flush(out_limit)) return kTooBig;
        out.slide();
        pos = out.pos;
        limits();
      }
      if (__builtin_expect(in.p > in_limit, 0)) return kNeedInput;
      in.refill();
      uint32_t e = lookup(lt, kLitBits);
      if (e_type(e) == kLit) {
        d[pos++] = (uint16_t)e_val(e);
        e = lookup(lt, kLitBits);
        if (e_type(e) == kLit) {
          d[pos++] = (uint16_t)e_val(e);
          continue;
        }
      }
      if (e_type(e) == kEob) break;
      if (e_type(e) != kLen) return kBad;
      const unsigned length = e_val(e) + in.take(e_extra(e));
      in.refill();
      const uint32_t de = lookup(dt, kDistBits);
      if (e_type(de) != kLen) return kBad;
      const size_t dist = e_val(de) + in.take(e_extra(de));
      const size_t src = pos - dist;  // (pos >= kWin >= dist)
      if (__builtin_expect(src < bad_below, 0)) return kBad;
      if (src < note_below && src < min_src) min_src = src;
      uint16_t* o = d + pos;
      const uint16_t* s = d + src;
      if (dist >= 4) {
        size_t i = 0;
        do {
          memcpy(o + i, s + i, 8);
          i += 4;
        } while (i < length);
      } else {
        for (size_t i = 0; i < length; ++i) o[i] = s[i];
      }
      pos += length;
    }
    out.pos = pos;
    min_src_io = min_src;
    return kAtBoundary;
  }

  std::vector<uint32_t> lit_, dist_, fixed_lit_, fixed_dist_;
};

// first bit in [from, to) at which a non-final dynamic block's header parses (codes complete); `to` if none
inline uint64_t find_block(const uint8_t* base, size_t nvalid, uint64_t from, uint64_t to, Inflate16& dec) {
  const uint64_t nbits = (uint64_t)nvalid * 8u;
  for (uint64_t bit = from; bit < to && bit + 80 < nbits; ++bit) {
    uint64_t w;
    memcpy(&w, base + (bit >> 3), 8);
    w >>= bit & 7;
    if ((w & 7u) != 4u) continue;  // BFINAL 0, BTYPE 10 (dynamic)
    if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;  // HLIT, HDIST
    // the code-length code must be complete: sum of 2^(7 - len) over its non-zero lengths == 2^7
    const unsigned hclen = (unsigned)((w >> 13) & 15u) + 4;
    uint64_t v = w >> 17;  // 40 bits = 13 lengths at hand
    unsigned sum = 0, have = 13;
    for (unsigned i = 0; i < hclen; ++i) {
      if (!have) {
        memcpy(&v, base + ((bit + 17 + 3 * i) >> 3), 8);
        v >>= (bit + 17 + 3 * i) & 7;
        have = 19;
      }
      const unsigned l = (unsigned)(v & 7u);
      v >>= 3;
      --have;
      if (l) sum += 128u >> l;
    }
    if (sum != 128u) continue;
    BitIn in;
    in.seek(base, bit + 3);
    if (dec.dynamic_header(in) && in.bitpos() <= nbits) return bit;
  }
  return to;
}

}  // namespace pgz

// The reader.  read() is gzread(): up to `want` inflated bytes, *at_end once nothing follows; on an error error()
// holds zlib's text (as gzerror's: "<path>: <message>") and read() returns what it had.
class ParallelGunzip {
 public:
  ParallelGunzip(int fd, uint64_t file_size, std::string path, unsigned threads, size_t chunk_bytes)
      : fd_(fd), size_(file_size), path_(std::move(path)), threads_(std::max(1u, threads)), chunk_(std::max<size_t>(chunk_bytes, 4096)),
        pool_threads_(threads_) {}
  ~ParallelGunzip() {
    if (zs_live_) inflateEnd(&zs_);
    free(cbuf_);
  }
  ParallelGunzip(const ParallelGunzip&) = delete;
  ParallelGunzip& operator=(const ParallelGunzip&) = delete;

  bool failed() const { return failed_; }
  const std::string& error() const { return error_; }
  struct Stats {
    uint64_t batches = 0, chunks_joined = 0, chunks_not_found = 0, chunks_discarded = 0, serial_bits = 0, members = 0;
    bool fell_back = false, truncated = false;
    std::string why;
    double s_load = 0, s_decode = 0, s_join = 0, s_windows = 0, s_narrow = 0, s_serial = 0;  // seconds per phase
    double t_decode_sum = 0, t_decode_max = 0, t_narrow_sum = 0, t_narrow_max = 0;  // thread seconds inside two of them (max: summed per round)
  };
  const Stats& stats() const { return stats_; }

  size_t read(char* dst, size_t want, bool* at_end) {
    size_t len = 0;
    while (len < want && !failed_) {
      if (ready_at_ < ready_n_) {
        const size_t n = std::min(want - len, ready_n_ - ready_at_);
        const unsigned T = (unsigned)std::min<size_t>(pool_threads_.size(), std::max<size_t>(1, n >> 22));
        char* to = dst + len;
        const char* from = ready_.get() + ready_at_;
        pool_threads_.run(T, [&](unsigned t) { memcpy(to + n * t / T, from + n * t / T, n * (t + 1) / T - n * t / T); });
        ready_at_ += n;
        len += n;
        continue;
      }
      if (done_) break;
      if (serial_) len += serial_read(dst + len, want - len);
      else len += batch(dst + len, want - len);
    }
    if (done_ && ready_at_ >= ready_n_) *at_end = true;
    return len;
  }

 private:
  void fail(const char* msg) {
    failed_ = true;
    error_ = path_ + ": " + msg;
  }
  // The file ends inside a member.  zlib's gzread hands out what could be inflated and then reports the end of the data:
  // Z_BUF_ERROR ("unexpected end of file") is the one error it does not return -1 for (gzread.c: gz_read), and what the
  // reference's gzgets loop sees is an ordinary end of file.  The same here.
  void truncated() {
    done_ = true;
    stats_.truncated = true;
  }
  void fall_back(const char* why) {
    serial_ = true;
    stats_.fell_back = true;
    if (stats_.why.empty()) stats_.why = why;
  }

  // compressed bytes [coff_, coff_ + n) of the file into cbuf_ (+ zeroed slack)
  bool load(size_t n) {
    if (cbuf_cap_ < n + pgz::kSlack) {
      free(cbuf_);
      cbuf_cap_ = n + pgz::kSlack;
      cbuf_ = static_cast<uint8_t*>(malloc(cbuf_cap_));
      if (!cbuf_) {
        cbuf_cap_ = 0;
        fail("out of memory");
        return false;
      }
    }
    const unsigned T = (unsigned)std::min<size_t>(pool_threads_.size(), std::max<size_t>(1, n >> 20));
    std::atomic<bool> bad{false};
    pool_threads_.run(T, [&](unsigned t) {  // (a tmpfs or page-cache file is copied, not waited for: every thread its share)
      const size_t a = (n * t / T) & ~(size_t)4095, b = t + 1 == T ? n : (n * (t + 1) / T) & ~(size_t)4095;
      size_t done = a;
      while (done < b) {
        const ssize_t got = pread(fd_, cbuf_ + done, b - done, (off_t)(coff_ + done));
        if (got <= 0) {
          bad = true;
          return;
        }
        done += (size_t)got;
      }
    });
    if (bad) {
      fail("read error");
      return false;
    }
    memset(cbuf_ + n, 0, pgz::kSlack);
    cn_ = n;
    return true;
  }

  // One round of the parallel path: output goes to dst (room bytes) first, the rest to ready_.  Returns bytes put into dst.
  size_t batch(char* dst, size_t room) {
    using namespace pgz;
    ++stats_.batches;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_start = now();
    const size_t n = (size_t)std::min<uint64_t>((uint64_t)chunk_ * threads_ + (bit0_ >> 3) + 1, size_ - coff_);
    if (!load(n)) return 0;
    const bool eof = coff_ + n == size_;
    if (!in_member_) {  // a member's header is expected at byte 0
      const long hl = gzip_header_len(cbuf_, cn_, eof);
      if (hl == -1) {
        if (first_member_) fall_back("no gzip header");  // (not reached: the caller looked at the magic)
        else done_ = true;  // bytes behind the last member: gzread ignores them
        return 0;
      }
      if (hl <= 0 || (size_t)hl >= cn_) {
        fall_back("gzip header");
        return 0;
      }
      bit0_ = (uint64_t)hl * 8u;
      in_member_ = true;
      first_member_ = false;
      win_n_ = 0;
      member_crc_ = crc32(0L, Z_NULL, 0);
      member_len_ = 0;
    }
    const uint64_t nbits = (uint64_t)cn_ * 8u;
    // chunk k looks for its first block from byte first + k * chunk_ on (chunk 0 starts where the last round ended)
    const size_t first = (size_t)(bit0_ >> 3);
    unsigned K = (unsigned)std::min<size_t>(threads_, (cn_ - first + chunk_ - 1) / chunk_);
    if (K < 1) K = 1;
    // (the chunks' symbol buffers are kept from round to round: fresh ones are tens of megabytes of page faults each)
    while (pool_.size() < K) pool_.emplace_back(new ChunkState());
    std::vector<ChunkState*> cs(K);
    for (unsigned k = 0; k < K; ++k) {
      cs[k] = pool_[k].get();
      cs[k]->reset();
    }
    // bytes a chunk may inflate to: 40 times its compressed bytes (FASTQ inflates 3 - 6 times; sixteen chunks of 4 MiB
    // are 2.5 GB at this bound).  Beyond it the file goes to zlib, which needs no memory for it.
    const size_t out_limit = chunk_ * 40 + (1u << 20);
    auto start_of = [&](unsigned k) { return (uint64_t)(first + (size_t)k * chunk_) * 8u; };
    std::atomic<bool> oom{false};
    const auto t_loaded = now();
    stats_.s_load += secs(t_start, t_loaded);
    std::vector<double> tsec(K, 0.0);
    pool_threads_.run(K, [&](unsigned k) {
      const auto t_in = now();
      struct Tick {
        double& d;
        std::chrono::steady_clock::time_point t0;
        ~Tick() { d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
      } tick{tsec[k], t_in};
      ChunkState& c = *cs[k];
      Inflate16 dec;
      const uint64_t stop = k + 1 < K ? start_of((unsigned)k + 1) : ~0ull;
      if (!c.out.init(chunk_ * 5)) {
        oom = true;
        return;
      }
      if (k == 0) {
        c.out.window_known(win_, win_n_);
        c.bit = c.start_bit = bit0_;
        c.found = true;
        dec.run(c, cbuf_, cn_, eof, stop, out_limit);
        return;
      }
      uint64_t from = start_of((unsigned)k);
      while (from < stop) {
        const uint64_t at = find_block(cbuf_, cn_, from, std::min(stop, nbits), dec);
        if (at >= std::min(stop, nbits)) break;
        c.reset();
        c.out.window_unknown();  // (again for every guess: a guess that ran for a while has slid its symbols over it)
        c.bit = c.start_bit = at;
        const Status st = dec.run(c, cbuf_, cn_, eof, stop, out_limit);
        // (a guess that dies within a few blocks was no block start; real damage is found by whoever decodes up to here)
        if (st == kBad && c.blocks < 4) {
          from = at + 1;
          continue;
        }
        if ((st == kNeedInput || st == kTooBig) && c.blocks == 0) break;
        c.found = true;
        break;
      }
    });
    if (oom) {
      fail("out of memory");
      return 0;
    }
    // ---- join, in file order ----
    const auto t_decoded = now();
    stats_.s_decode += secs(t_loaded, t_decoded);
    {
      double mx = 0;
      for (double v : tsec) {
        stats_.t_decode_sum += v;
        mx = std::max(mx, v);
      }
      stats_.t_decode_max += mx;
    }
    std::vector<unsigned> joined{0};
    Inflate16 dec;
    unsigned cur = 0;
    uint64_t serial_bits = 0;
    const char* stop_why = nullptr;
    for (unsigned j = 1; j < K && !stop_why;) {
      ChunkState& c = *cs[cur];
      if (c.status != kAtBoundary) break;  // the end of the last member, the end of the bytes at hand, or trouble: the round ends here
      ChunkState& nx = *cs[j];
      if (!nx.found) {
        ++stats_.chunks_not_found;
        ++j;
        if (j == K) extend(dec, c, eof, ~0ull, out_limit, serial_bits);
        continue;
      }
      if (c.bit == nx.start_bit) {
        joined.push_back(j);
        cur = j;
        ++j;
        continue;
      }
      if (c.bit > nx.start_bit) {  // the guess was wrong: its work is lost, the predecessor decodes on
        ++stats_.chunks_discarded;
        ++j;
        if (j == K) extend(dec, c, eof, ~0ull, out_limit, serial_bits);
        continue;
      }
      extend(dec, c, eof, nx.start_bit, out_limit, serial_bits);
      if (serial_bits > (uint64_t)chunk_ * 8u * 3u) stop_why = "block starts not found";
    }
    stats_.serial_bits += serial_bits;
    const auto t_joined = now();
    stats_.s_join += secs(t_decoded, t_joined);
    // ---- windows, in file order; what the chunks may have asked of them ----
    // valid = bytes of the current member in front of the chunk (what a copy may reach back into), capped at kWin
    std::vector<std::vector<uint8_t>> wins(joined.size());
    size_t use = joined.size();
    {
      std::vector<uint8_t> w(win_, win_ + win_n_);
      uint64_t valid = std::min<uint64_t>(member_len_, kWin);
      if (win_n_ < valid) valid = win_n_;
      std::vector<uint64_t> valid_at(joined.size() + 1, 0);
      for (size_t q = 0; q < joined.size(); ++q) {
        ChunkState& c = *cs[joined[q]];
        wins[q] = w;
        valid_at[q] = valid;
        // the chunk's first member continues the one in front of it: its copies must stay inside `valid`
        const size_t reach = c.ends.empty() && c.member_floor < 0 ? c.min_src : c.min_src_member0;
        if ((uint64_t)(kWin - reach) > valid) {
          use = q;  // zlib's "invalid distance too far back" - let zlib say it
          stop_why = "distance too far back";
          break;
        }
        // the next window: the last kWin bytes of (w, this chunk's output)
        // (from the chunk's bytes and the notes of its markers: its symbol buffer may have slid past the point it stands at)
        const size_t produced = c.safe_bytes;
        std::vector<uint8_t> nw;
        const size_t keep = produced >= kWin ? 0 : std::min<size_t>(w.size(), kWin - produced);
        const size_t from = produced > kWin ? produced - kWin : 0;
        nw.reserve(keep + (produced - from));
        nw.insert(nw.end(), w.end() - (long)keep, w.end());
        nw.insert(nw.end(), c.out.bytes + from, c.out.bytes + produced);
        for (size_t k = c.safe_marks; k-- > 0 && c.out.marks[k].at >= from;)
          nw[keep + (c.out.marks[k].at - from)] = window_byte(c.out.marks[k].idx, w);
        valid = c.member_floor >= 0 ? std::min<uint64_t>(produced - (size_t)c.member_floor, kWin) : std::min<uint64_t>(valid + produced, kWin);
        w.swap(nw);
        if (c.status != kAtBoundary && q + 1 < joined.size()) {  // (cannot be: the join stops at such a chunk)
          use = q + 1;
          break;
        }
      }
      valid_at[joined.size()] = valid;
      if (use == joined.size()) next_win_.swap(w);
      else next_win_ = wins[use];  // the window in front of the first chunk that is not used
      // (what lies in front of the current member's first byte is no window of it: zlib must not be handed it either)
      const size_t keep = (size_t)std::min<uint64_t>(valid_at[use], next_win_.size());
      next_win_.erase(next_win_.begin(), next_win_.end() - (long)keep);
    }
    // ---- markers -> bytes, CRC-32 per stretch between member ends: in parallel ----
    const auto t_windows = now();
    stats_.s_windows += secs(t_joined, t_windows);
    struct Part {
      std::vector<uint32_t> crc;  // one per stretch (ends.size() + 1)
      std::vector<size_t> len;
      size_t off = 0;  // of the chunk's bytes in this round's output
    };
    std::vector<Part> parts(use);
    size_t total = 0;
    for (size_t q = 0; q < use; ++q) {
      parts[q].off = total;
      total += cs[joined[q]]->safe_bytes;
    }
    const size_t direct = std::min(room, total);
    if (total - direct > ready_cap_) {  // (never zero-filled: every byte is written below)
      ready_cap_ = total - direct + (total - direct) / 4;
      ready_.reset(new char[ready_cap_]);
    }
    ready_n_ = total - direct;
    ready_at_ = 0;
    std::vector<double> nsec(use + 1, 0.0);
    if (use) pool_threads_.run((unsigned)use, [&](unsigned q) {
      struct Tick {
        double& d;
        std::chrono::steady_clock::time_point t0;
        ~Tick() { d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
      } tick{nsec[q], std::chrono::steady_clock::now()};
      ChunkState& c = *cs[joined[q]];
      Part& p = parts[q];
      // the bytes that were markers, now that the window in front of the chunk is known
      {
        const std::vector<uint8_t>& w = wins[q];
        uint8_t* by = c.out.bytes;
        for (size_t k = 0; k < c.safe_marks; ++k) by[c.out.marks[k].at] = window_byte(c.out.marks[k].idx, w);
      }
      size_t a = 0;
      for (size_t s = 0; s <= c.ends.size(); ++s) {
        const size_t b = s < c.ends.size() ? c.ends[s].out_pos : c.safe_bytes;
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        // [a, b) of the chunk = bytes [o, o + b - a) of the round: to dst below `direct`, to ready_ above.  Summed and
        // copied piece by piece, so that the copy finds in the cache what the sum has just read.
        size_t o = p.off + a, i = a;
        while (i < b) {
          char* out;
          size_t m = std::min<size_t>(b - i, 256u << 10);
          if (o < direct) {
            out = dst + o;
            m = std::min(m, direct - o);
          } else {
            out = ready_.get() + (o - direct);
          }
          crc = crc32_16(crc, c.out.bytes + i, m);
          memcpy(out, c.out.bytes + i, m);
          i += m;
          o += m;
        }
        p.crc.push_back(crc);
        p.len.push_back(b - a);
        a = b;
      }
    });
    stats_.s_narrow += secs(t_windows, now());
    {
      double mx = 0;
      for (double v : nsec) {
        stats_.t_narrow_sum += v;
        mx = std::max(mx, v);
      }
      stats_.t_narrow_max += mx;
    }
    // ---- members' checks, in file order ----
    for (size_t q = 0; q < use && !failed_; ++q) {
      const ChunkState& c = *cs[joined[q]];
      for (size_t s = 0; s < parts[q].crc.size(); ++s) {
        member_crc_ = crc32_combine(member_crc_, parts[q].crc[s], (z_off_t)parts[q].len[s]);
        member_len_ += parts[q].len[s];
        if (s < c.ends.size()) {
          ++stats_.members;
          if ((uint32_t)member_crc_ != c.ends[s].crc) fail("incorrect data check");
          else if ((uint32_t)member_len_ != c.ends[s].isize) fail("incorrect length check");
          member_crc_ = crc32(0L, Z_NULL, 0);
          member_len_ = 0;
        }
      }
    }
    if (failed_) {
      ready_n_ = 0;
      return 0;
    }
    stats_.chunks_joined += use;
    // ---- where the next round starts ----
    if (use == 0) {  // nothing usable (the very first chunk was refused): zlib from here
      fall_back(stop_why ? stop_why : "no progress");
      coff_ += bit0_ >> 3;
      bit0_ &= 7u;
      return 0;
    }
    const ChunkState& last = *cs[joined[use - 1]];
    if (!next_win_.empty()) memcpy(win_, next_win_.data(), next_win_.size());
    win_n_ = next_win_.size();
    in_member_ = !last.finished;
    if (last.finished) {
      done_ = true;
    } else {
      const uint64_t adv = last.bit >> 3;
      if (use < joined.size() || stop_why) fall_back(stop_why ? stop_why : "join");
      else if (last.status == kBad) fall_back("a block zlib must judge");
      else if (last.status == kTooBig) fall_back("a chunk inflates too far");
      else if (last.status == kNeedInput) {
        if (eof) fall_back("the file ends inside a block");  // zlib's "unexpected end of file"
        else if (adv == (bit0_ >> 3) && use == 1 && last.safe_bytes == 0) fall_back("a block larger than the window");
      }
      coff_ += adv;
      bit0_ = last.bit & 7u;
    }
    return direct;
  }

  // the predecessor decodes on, on this thread, to the first boundary at or behind stop_bit
  void extend(pgz::Inflate16& dec, pgz::ChunkState& c, bool eof, uint64_t stop_bit, size_t out_limit, uint64_t& serial_bits) {
    const uint64_t before = c.bit;
    dec.run(c, cbuf_, cn_, eof, stop_bit, out_limit + c.safe_bytes);
    serial_bits += c.bit - before;
  }

  // byte i of the kWin bytes in front of a chunk; w holds the last w.size() of them (what lies before is not to be asked for)
  static uint8_t window_byte(size_t i, const std::vector<uint8_t>& w) {
    const size_t missing = pgz::kWin - w.size();
    return i >= missing ? w[i - missing] : 0;
  }

  // ---- one zlib stream from (coff_, bit0_) on, primed with win_ ----
  size_t serial_read(char* dst, size_t want) {
    using namespace pgz;
    if (!zs_live_) {
      memset(&zs_, 0, sizeof(zs_));
      if (inflateInit2(&zs_, -15) != Z_OK) {
        fail("out of memory");
        return 0;
      }
      zs_live_ = true;
      sbuf_.resize(1u << 20);
      s_at_ = s_n_ = 0;
      if (in_member_) {
        unsigned char first = 0;
        const unsigned off = (unsigned)(bit0_ & 7u);
        if (off) {
          if (pread(fd_, &first, 1, (off_t)coff_) != 1) {
            truncated();
            return 0;
          }
          ++coff_;
          inflatePrime(&zs_, 8 - (int)off, first >> off);
        }
        if (win_n_) inflateSetDictionary(&zs_, win_, (uInt)win_n_);
      }
    }
    size_t len = 0;
    while (len < want && !done_ && !failed_) {
      if (transparent_) {
        if (s_at_ < s_n_) {
          const size_t m = std::min(want - len, s_n_ - s_at_);
          memcpy(dst + len, sbuf_.data() + s_at_, m);
          s_at_ += m;
          len += m;
          continue;
        }
        if (coff_ >= size_) {
          done_ = true;
          break;
        }
        const size_t n = (size_t)std::min<uint64_t>(want - len, size_ - coff_);
        const ssize_t got = pread(fd_, dst + len, n, (off_t)coff_);
        if (got <= 0) {
          fail("read error");
          break;
        }
        coff_ += (uint64_t)got;
        len += (size_t)got;
        continue;
      }
      if (s_at_ == s_n_) {
        const size_t n = (size_t)std::min<uint64_t>(sbuf_.size(), size_ - coff_);
        if (n) {
          const ssize_t got = pread(fd_, sbuf_.data(), n, (off_t)coff_);
          if (got <= 0) {
            fail("read error");
            break;
          }
          coff_ += (uint64_t)got;
          s_at_ = 0;
          s_n_ = (size_t)got;
        }
      }
      if (!in_member_) {
        // the trailer's eight bytes and the next header come through a small look-ahead of their own
        std::vector<uint8_t> hdr(sbuf_.begin() + (long)s_at_, sbuf_.begin() + (long)s_n_);
        while (hdr.size() < (1u << 16) && coff_ < size_) {  // (a header beyond 64 KiB: zlib would take it, this gives up)
          const size_t n = (size_t)std::min<uint64_t>(1u << 16, size_ - coff_);
          const size_t old = hdr.size();
          hdr.resize(old + n);
          if (pread(fd_, hdr.data() + old, n, (off_t)coff_) != (ssize_t)n) {
            fail("read error");
            return len;
          }
          coff_ += n;
        }
        if (hdr.empty()) {
          done_ = true;
          break;
        }
        const long hl = gzip_header_len(hdr.data(), hdr.size(), coff_ == size_);
        if (hl == -1) {
          if (first_member_) {  // no gzip file at all: gzread hands out the bytes as they are
            transparent_ = true;
            sbuf_.assign(hdr.begin(), hdr.end());
            s_at_ = 0;
            s_n_ = hdr.size();
            if (sbuf_.size() < (1u << 20)) sbuf_.resize(1u << 20);
            continue;
          }
          done_ = true;  // trailing garbage
          break;
        }
        if (hl == -2) {
          fail(hdr[2] != 8 ? "unknown compression method" : "unknown header flags set");
          break;
        }
        if (hl == 0) {
          truncated();
          break;
        }
        sbuf_.assign(hdr.begin() + hl, hdr.end());
        if (sbuf_.size() < (1u << 20)) sbuf_.resize(1u << 20);
        s_at_ = 0;
        s_n_ = hdr.size() - (size_t)hl;
        inflateReset(&zs_);
        in_member_ = true;
        first_member_ = false;
        member_crc_ = crc32(0L, Z_NULL, 0);
        member_len_ = 0;
        continue;
      }
      if (s_at_ == s_n_ && coff_ >= size_) {
        truncated();
        break;
      }
      zs_.next_in = sbuf_.data() + s_at_;
      zs_.avail_in = (uInt)(s_n_ - s_at_);
      zs_.next_out = reinterpret_cast<Bytef*>(dst + len);
      zs_.avail_out = (uInt)std::min<size_t>(want - len, 1u << 30);
      const uInt out_before = zs_.avail_out;
      const int rc = inflate(&zs_, Z_NO_FLUSH);
      s_at_ = s_n_ - zs_.avail_in;
      const size_t got = out_before - zs_.avail_out;
      member_crc_ = crc32(member_crc_, reinterpret_cast<const Bytef*>(dst + len), (uInt)got);
      member_len_ += got;
      len += got;
      if (rc == Z_STREAM_END) {
        uint8_t t[8];
        size_t have = 0;
        while (have < 8) {
          if (s_at_ < s_n_) t[have++] = sbuf_[s_at_++];
          else if (coff_ < size_) {
            const size_t n = (size_t)std::min<uint64_t>(sbuf_.size(), size_ - coff_);
            const ssize_t g = pread(fd_, sbuf_.data(), n, (off_t)coff_);
            if (g <= 0) {
              fail("read error");
              return len;
            }
            coff_ += (uint64_t)g;
            s_at_ = 0;
            s_n_ = (size_t)g;
          } else {
            truncated();
            return len;
          }
        }
        const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t isz = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        ++stats_.members;
        if ((uint32_t)member_crc_ != crc) fail("incorrect data check");
        else if ((uint32_t)member_len_ != isz) fail("incorrect length check");
        in_member_ = false;
      } else if (rc == Z_NEED_DICT || rc == Z_DATA_ERROR || rc == Z_MEM_ERROR || rc == Z_STREAM_ERROR) {
        fail(zs_.msg ? zs_.msg : "compressed data error");
      } else if (rc == Z_BUF_ERROR && got == 0 && zs_.avail_in == 0 && coff_ >= size_) {
        truncated();
      }
    }
    return len;
  }

  int fd_;
  uint64_t size_;
  std::string path_;
  unsigned threads_;
  size_t chunk_;
  // where the next byte comes from: file offset of the window's first byte, bit in it, inside a member or in front of a header
  uint64_t coff_ = 0, bit0_ = 0;
  bool in_member_ = false, first_member_ = true, done_ = false, serial_ = false, failed_ = false, transparent_ = false;
  uint8_t win_[pgz::kWin];  // the last win_n_ inflated bytes
  size_t win_n_ = 0;
  std::vector<uint8_t> next_win_;
  uLong member_crc_ = 0;
  uint64_t member_len_ = 0;
  uint8_t* cbuf_ = nullptr;
  size_t cbuf_cap_ = 0, cn_ = 0;
  std::unique_ptr<char[]> ready_;
  size_t ready_n_ = 0, ready_cap_ = 0, ready_at_ = 0;
  ReaderPool pool_threads_;
  std::vector<std::unique_ptr<pgz::ChunkState>> pool_;
  std::string error_;
  Stats stats_;
  // serial path
  z_stream zs_;
  bool zs_live_ = false;
  std::vector<uint8_t> sbuf_;
  size_t s_at_ = 0, s_n_ = 0;
};

}  // namespace fqhost
