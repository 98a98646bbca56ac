// fq_fastdeflate.h - a deflate compressor of zlib level 1's size class at three times the speed of the reference's level.
//
// Why it exists: bin/fastq_pre_barcodes, fastq_filterpair and fastq_trim_poly_at write .fastq.gz, as the reference does
// (gzopen "w4" / "w3", src/fastq_pre_barcodes.c:582, one thread, 40 MB/s).  The GPU side of those programs takes
// milliseconds per gigabyte, their input is inflated at gigabytes per second (fq_pgzip.h), and zlib's deflate at the
// reference's level gives 70 MB/s per core - on the box's 16 cores the programs ARE their deflate (22 s for 50 M pairs,
// 1.7 s for everything else).  FQGPU_GZIP_FAST=1 selects this compressor for the members GzipMembers writes
// (fq_parallel.h); the default stays zlib at the reference's level, whose file sizes people know: on the reference's own
// 10 000-read fixtures this one gives 0.373 of the input where zlib gives 0.376 at level 1 and 0.341 at level 4.
//
// What it is: greedy LZ77 with one hash probe per position (four-byte hash, 32 KiB window, matches of 4..258 bytes; a
// short match far away costs more bits than the literals it replaces - sequence lines are two bits a base - so short
// matches must be near), blocks of 96 KiB of input, a dynamic Huffman code per block built from the block's own
// counts (lengths limited to 15 bits the way zlib does it), the code lengths run-length coded as RFC 1951 3.2.7
// prescribes, three literals per store in the bit writer.  Any inflater reads the result; the tests inflate it with
// zlib (tests/test_fastdeflate.py).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace fqhost {
namespace fdef {

struct BitOut {  // without a branch: the pending bits are stored whole, the pointer moves by the bytes that are complete
  uint8_t* p;
  uint64_t bb = 0;
  unsigned bc = 0;  // 0..7 between calls
  explicit BitOut(uint8_t* dst) : p(dst) {}
  inline void put(uint64_t v, unsigned n) {  // n <= 56; eight bytes behind p must be writable
    bb |= v << bc;
    bc += n;
    memcpy(p, &bb, 8);
    p += bc >> 3;
    bb >>= bc & ~7u;
    bc &= 7u;
  }
  uint8_t* finish() {  // pads to a byte boundary
    if (bc) *p++ = (uint8_t)bb;
    bb = 0;
    bc = 0;
    return p;
  }
};

inline uint32_t reverse_code(uint32_t c, unsigned n) {
  uint32_t r = 0;
  for (unsigned i = 0; i < n; ++i) {
    r = (r << 1) | (c & 1u);
    c >>= 1;
  }
  return r;
}

// Huffman code lengths (<= maxbits) for freq[0, n): symbols with freq 0 get length 0; at least two symbols get a code
inline void code_lengths(const uint32_t* freq_in, unsigned n, unsigned maxbits, uint8_t* lens) {
  uint32_t freq[288];
  unsigned used = 0;
  for (unsigned s = 0; s < n; ++s) {
    freq[s] = freq_in[s];
    used += freq[s] != 0;
  }
  for (unsigned s = 0; used < 2 && s < n; ++s)  // (an inflater wants a complete code: two symbols of one bit each)
    if (!freq[s]) {
      freq[s] = 1;
      ++used;
    }
  // leaves sorted by count, internal nodes appended in creation order (their counts never decrease): two queues
  struct Node {
    uint32_t f;
    int left, right;  // -1: leaf
    int sym;
  };
  Node nodes[2 * 288];
  int order[288], m = 0;
  for (unsigned s = 0; s < n; ++s)
    if (freq[s]) order[m++] = (int)s;
  std::sort(order, order + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
  for (int i = 0; i < m; ++i) nodes[i] = Node{freq[order[i]], -1, -1, order[i]};
  int leaf = 0, inner = m, next = m;
  auto take = [&]() {
    if (leaf < m && (inner >= next || nodes[leaf].f <= nodes[inner].f)) return leaf++;
    return inner++;
  };
  while ((m - leaf) + (next - inner) > 1) {
    const int a = take(), b = take();
    nodes[next] = Node{nodes[a].f + nodes[b].f, a, b, -1};
    ++next;
  }
  // depths, from the root down (children have smaller indices than their parent).  Limiting the lengths (zlib's
  // gen_bitlen): every leaf deeper than maxbits is lifted to it; a subtree of L leaves that hung below that level then
  // claims L places where it had one, and each round of the loop below makes one place - a leaf of the deepest level
  // that still has one goes a level down and takes a lifted leaf as its brother.  L - 1 rounds per subtree = half the
  // number of NODES, leaves and inner ones, below the level.
  int depth[2 * 288];
  depth[next - 1] = 0;
  unsigned bl_count[16] = {0};
  int overflow = 0;
  for (int i = next - 1; i >= 0; --i) {
    if (depth[i] > (int)maxbits) ++overflow;
    if (nodes[i].left >= 0) {
      depth[nodes[i].left] = depth[i] + 1;
      depth[nodes[i].right] = depth[i] + 1;
    } else {
      ++bl_count[std::min(depth[i], (int)maxbits)];
    }
  }
  while (overflow > 0) {
    unsigned bits = maxbits - 1;
    while (bl_count[bits] == 0) --bits;
    --bl_count[bits];
    bl_count[bits + 1] += 2;
    --bl_count[maxbits];
    overflow -= 2;
  }
  // the rarest symbols take the longest codes
  memset(lens, 0, n);
  int at = 0;  // leaves in order of count
  for (unsigned b = maxbits; b >= 1; --b)
    for (unsigned k = 0; k < bl_count[b]; ++k) lens[nodes[at++].sym] = (uint8_t)b;
}

inline void canonical_codes(const uint8_t* lens, unsigned n, uint16_t* codes) {  // bit-reversed: ready to be put LSB first
  unsigned count[16] = {0}, next[16];
  for (unsigned s = 0; s < n; ++s) ++count[lens[s]];
  count[0] = 0;
  unsigned code = 0;
  for (unsigned b = 1; b <= 15; ++b) {
    code = (code + count[b - 1]) << 1;
    next[b] = code;
  }
  for (unsigned s = 0; s < n; ++s) codes[s] = lens[s] ? (uint16_t)reverse_code(next[lens[s]]++, lens[s]) : 0;
}

class FastDeflate {
 public:
  FastDeflate() : table_(kHashSize) {
    // length 3..258 -> symbol, extra bits; distance via two small tables (zlib's layout of the same mapping)
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (unsigned s = 0; s < 28; ++s)
      for (unsigned l = lbase[s]; l < lbase[s] + (1u << lext[s]) && l <= 258; ++l) len_sym_[l] = (uint8_t)s;
    len_sym_[258] = 28;
    memcpy(lbase_, lbase, sizeof(lbase));
    memcpy(lext_, lext, sizeof(lext));
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    memcpy(dbase_, dbase, sizeof(dbase));
    memcpy(dext_, dext, sizeof(dext));
    for (unsigned s = 0; s < 30; ++s)
      for (unsigned d = dbase[s]; d < dbase[s] + (1u << dext[s]); ++d) {
        if (d <= 256) dist_sym_lo_[d] = (uint8_t)s;
        else dist_sym_hi_[(d - 1) >> 7] = (uint8_t)s;
      }
  }

  // the raw deflate stream of [src, src + n) (its last block marked final); dst must hold bound(n) bytes.  Returns the end.
  static size_t bound(size_t n) { return n + n / 4 + 4096; }  // (+ the eight bytes a store may reach ahead)
  uint8_t* compress(const uint8_t* src, size_t n, uint8_t* dst) {
    BitOut out(dst);
    for (auto& t : table_) t = -1;
    size_t i = 0, block_start = 0;
    unsigned misses = 0;
    matches_.clear();
    const size_t hash_end = n >= 8 ? n - 8 : 0;  // positions whose four bytes (and the extension's loads) are safe to read
    while (i < hash_end) {
      if (i - block_start >= kBlockBytes) {  // (a block = a stretch of input and the matches that start in it)
        flush_block(out, src, block_start, i, false);
        block_start = i;
        matches_.clear();
      }
      uint32_t w;
      memcpy(&w, src + i, 4);
      const uint32_t h = (w * 0x9E3779B1u) >> (32 - kHashBits);
      const int32_t cand = table_[h];
      table_[h] = (int32_t)i;
      if (cand >= 0 && i - (size_t)cand <= 32768) {
        uint32_t v;
        memcpy(&v, src + cand, 4);
        if (v == w) {
          const size_t maxlen = std::min<size_t>(258, n - i);
          size_t len = 4;
          while (len + 8 <= maxlen) {
            uint64_t a, b;
            memcpy(&a, src + cand + len, 8);
            memcpy(&b, src + i + len, 8);
            if (a != b) {
              len += (size_t)__builtin_ctzll(a ^ b) >> 3;
              goto extended;
            }
            len += 8;
          }
          while (len < maxlen && src[cand + len] == src[i + len]) ++len;
        extended:
          const size_t dist = i - (size_t)cand;
          // a match must pay for its distance: four or five bytes from far away cost more than their literals
          if (len >= 6 || (len == 5 && dist <= 4096) || dist <= 512) {
            matches_.push_back(Match{(uint32_t)i, (uint16_t)len, (uint16_t)(dist - 1)});
            misses = 0;
            // two of the positions the match covers go into the table (the next header line will look for them)
            if (len >= 8 && i + len < hash_end) {
              uint32_t x;
              memcpy(&x, src + i + len - 4, 4);
              table_[(x * 0x9E3779B1u) >> (32 - kHashBits)] = (int32_t)(i + len - 4);
              memcpy(&x, src + i + (len >> 1), 4);
              table_[(x * 0x9E3779B1u) >> (32 - kHashBits)] = (int32_t)(i + (len >> 1));
            }
            i += len;
            continue;
          }
        }
      }
      // (where nothing has matched for a while - quality and sequence lines - only every second, third .. sixteenth position
      // is looked up; the bytes between go out as literals.  A match resets the stride.  Stride + 1 per 4 / 8 / 32 misses:
      // 182 / 162 / 133 MB/s at 0.439 / 0.433 / 0.431 of the input on synthetic 150-base reads, one 2.1 GHz core.)
      i += 1 + (misses >> 2);
      if (misses < 60) ++misses;
    }
    flush_block(out, src, block_start, n, true);
    return out.finish();
  }

 private:
  static constexpr unsigned kHashBits = 15, kHashSize = 1u << kHashBits;
  static constexpr size_t kBlockBytes = 96u << 10;
  struct Match {
    uint32_t pos;
    uint16_t len, dist1;  // dist - 1
  };

  // input [from, to) as one block: literals where no match of matches_ covers, in order
  void flush_block(BitOut& out, const uint8_t* src, size_t from, size_t to, bool final) {
    // the block's counts (four tables for the literals: neighbouring bytes are often the same one)
    uint32_t lf[4][256];
    memset(lf, 0, sizeof(lf));
    memset(lit_freq_, 0, sizeof(lit_freq_));
    memset(dist_freq_, 0, sizeof(dist_freq_));
    auto count_run = [&](size_t a, size_t b) {
      size_t k = a;
      for (; k + 4 <= b; k += 4) {
        ++lf[0][src[k]];
        ++lf[1][src[k + 1]];
        ++lf[2][src[k + 2]];
        ++lf[3][src[k + 3]];
      }
      for (; k < b; ++k) ++lf[0][src[k]];
    };
    size_t at = from;
    for (const Match& m : matches_) {
      count_run(at, m.pos);
      ++lit_freq_[257 + len_sym_[m.len]];
      const uint32_t d = (uint32_t)m.dist1 + 1;
      ++dist_freq_[d <= 256 ? dist_sym_lo_[d] : dist_sym_hi_[(d - 1) >> 7]];
      at = (size_t)m.pos + m.len;
    }
    if (at < to) count_run(at, to);  // (a match may reach beyond the block's end: the next block starts behind it)
    for (unsigned c = 0; c < 256; ++c) lit_freq_[c] = lf[0][c] + lf[1][c] + lf[2][c] + lf[3][c];
    ++lit_freq_[256];
    uint8_t ll[288], dl[32];
    code_lengths(lit_freq_, 286, 15, ll);
    code_lengths(dist_freq_, 30, 15, dl);
    uint16_t lc[288], dc[32];
    canonical_codes(ll, 286, lc);
    canonical_codes(dl, 30, dc);
    unsigned hlit = 286, hdist = 30;
    while (hlit > 257 && !ll[hlit - 1]) --hlit;
    while (hdist > 1 && !dl[hdist - 1]) --hdist;
    // the two length arrays as one sequence, run-length coded (RFC 1951 3.2.7)
    uint8_t seq[320];
    memcpy(seq, ll, hlit);
    memcpy(seq + hlit, dl, hdist);
    const unsigned total = hlit + hdist;
    struct Cl {
      uint8_t sym, extra;
    } cl[320];
    unsigned ncl = 0;
    uint32_t cl_freq[19] = {0};
    for (unsigned k = 0; k < total;) {
      unsigned run = 1;
      while (k + run < total && seq[k + run] == seq[k]) ++run;
      const unsigned v = seq[k];
      unsigned left = run;
      if (v == 0) {
        while (left >= 11) {
          const unsigned r = std::min(left, 138u);
          cl[ncl++] = Cl{18, (uint8_t)(r - 11)};
          left -= r;
        }
        if (left >= 3) {
          cl[ncl++] = Cl{17, (uint8_t)(left - 3)};
          left = 0;
        }
      } else {
        cl[ncl++] = Cl{(uint8_t)v, 0};
        --left;
        while (left >= 3) {
          const unsigned r = std::min(left, 6u);
          cl[ncl++] = Cl{16, (uint8_t)(r - 3)};
          left -= r;
        }
      }
      while (left--) cl[ncl++] = Cl{(uint8_t)v, 0};
      k += run;
    }
    for (unsigned k = 0; k < ncl; ++k) ++cl_freq[cl[k].sym];
    uint8_t cll[19];
    code_lengths(cl_freq, 19, 7, cll);
    uint16_t clc[19];
    canonical_codes(cll, 19, clc);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    unsigned hclen = 19;
    while (hclen > 4 && !cll[order[hclen - 1]]) --hclen;
    out.put(final ? 1u : 0u, 1);
    out.put(2u, 2);
    out.put(hlit - 257, 5);
    out.put(hdist - 1, 5);
    out.put(hclen - 4, 4);
    for (unsigned k = 0; k < hclen; ++k) out.put(cll[order[k]], 3);
    for (unsigned k = 0; k < ncl; ++k) {
      out.put(clc[cl[k].sym], cll[cl[k].sym]);
      if (cl[k].sym == 16) out.put(cl[k].extra, 2);
      else if (cl[k].sym == 17) out.put(cl[k].extra, 3);
      else if (cl[k].sym == 18) out.put(cl[k].extra, 7);
    }
    // code and length of a literal in one word
    uint32_t lit[256];
    for (unsigned c = 0; c < 256; ++c) lit[c] = (uint32_t)lc[c] | ((uint32_t)ll[c] << 16);
    auto emit_run = [&](size_t a, size_t b) {
      size_t k = a;
      for (; k + 3 <= b; k += 3) {  // three literals (at most 45 bits) per store
        const uint32_t e0 = lit[src[k]], e1 = lit[src[k + 1]], e2 = lit[src[k + 2]];
        const unsigned n0 = e0 >> 16, n1 = e1 >> 16;
        out.put((uint64_t)(e0 & 0xFFFFu) | ((uint64_t)(e1 & 0xFFFFu) << n0) | ((uint64_t)(e2 & 0xFFFFu) << (n0 + n1)), n0 + n1 + (e2 >> 16));
      }
      for (; k < b; ++k) {
        const uint32_t e = lit[src[k]];
        out.put(e & 0xFFFFu, e >> 16);
      }
    };
    at = from;
    for (const Match& m : matches_) {
      emit_run(at, m.pos);
      const unsigned len = m.len, ls = len_sym_[len];
      out.put(lc[257 + ls], ll[257 + ls]);
      if (lext_[ls]) out.put(len - lbase_[ls], lext_[ls]);
      const uint32_t d = (uint32_t)m.dist1 + 1;
      const unsigned ds = d <= 256 ? dist_sym_lo_[d] : dist_sym_hi_[(d - 1) >> 7];
      out.put(dc[ds], dl[ds]);
      if (dext_[ds]) out.put(d - dbase_[ds], dext_[ds]);
      at = (size_t)m.pos + m.len;
    }
    if (at < to) emit_run(at, to);
    out.put(lc[256], ll[256]);
  }

  std::vector<int32_t> table_;
  std::vector<Match> matches_;
  uint32_t lit_freq_[288], dist_freq_[32];
  uint8_t len_sym_[259] = {0}, dist_sym_lo_[257] = {0}, dist_sym_hi_[256] = {0};
  uint16_t lbase_[29], dbase_[30];
  uint8_t lext_[29], dext_[30];
};

// one gzip member of [p, p + n)
inline bool gzip_member_fast(const char* p, size_t n, std::vector<uint8_t>& out) {
  if (n > (1u << 30)) return false;  // (positions are 32-bit: GzipMembers hands over a MiB at a time)
  static thread_local FastDeflate fd;
  static thread_local std::vector<uint8_t> scratch;  // (kept: a fresh vector of this size would be zero-filled every time)
  const size_t need = 10 + FastDeflate::bound(n) + 16;
  if (scratch.size() < need) scratch.resize(need);
  static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4 /* XFL: fastest */, 3 /* OS: Unix */};
  memcpy(scratch.data(), hdr, 10);
  uint8_t* end = fd.compress(reinterpret_cast<const uint8_t*>(p), n, scratch.data() + 10);
  uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
  for (size_t o = 0; o < n; o += 1u << 30) crc = (uint32_t)crc32(crc, reinterpret_cast<const Bytef*>(p + o), (uInt)std::min<size_t>(n - o, 1u << 30));
  const uint32_t isize = (uint32_t)n;
  for (int k = 0; k < 4; ++k) *end++ = (uint8_t)(crc >> (8 * k));
  for (int k = 0; k < 4; ++k) *end++ = (uint8_t)(isize >> (8 * k));
  out.assign(scratch.data(), end);
  return true;
}

}  // namespace fdef
}  // namespace fqhost
