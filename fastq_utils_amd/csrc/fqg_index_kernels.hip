// fqg_index_kernels.hip - the GPU read-name index: replaces hash.c as used by
// fastq_index_readnames (reference src/fastq.c:396-439, :577-611) and by the file-2 pairing
// loop of fastq_info (reference src/fastq_info.c:333-356).
//
// Three arrays (what each costs per access is measured in tools/kbench/rmwbench.hip: on MI355X a random 8-byte load
// runs at 50 G/s, a random 64-byte load at 20 G/s, an atomic at 19 G/s wherever it goes - but atomics to neighbouring
// words at 100 G/s):
//   slots[2^k >= 2 * names]  open addressing, linear probing, ONE 8-byte word per slot
//         { tag : 24 | record : 40 }   (all ones = empty)      the insert claims it with a CAS: its one random access
//   names[record]            64 bytes per inserted record, in RECORD order: length + the first 56 bytes of the
//                            canonical name.  Written by the insert as a sequential stream; a look-up that found a
//                            tag reads names[record] to decide on the BYTES - and when the second file comes in the
//                            first one's order (the usual case) neighbouring lanes read neighbouring records
//   claims[record]           smallest record of the asking file that took the entry (all ones = nobody), in record
//                            order for the same reason: the atomicMin of neighbouring lanes fall into one line
// `record` is the global index of the record that owns the name, `tag` 24 further bits of the name's 64-bit hash.
// Equality is always decided on the name bytes (names beyond 56 bytes, and tag matches while the insert is still
// running, through the images the index keeps references to), so hash collisions can neither fake nor hide a duplicate.
//
// Where the names come from: the streaming pass copies every header line it sees into a 64-byte record while the
// chunk is in LDS (NameCapture, fqg_device.h), and k_names_pass works from those records - a sequential read instead
// of one header line per 349-byte stride.  Headers the capture could not vouch for (mis-speculated chunks, lines that
// straddle a chunk, lines longer than a record) and frames that were not streamed go through the line index and the
// image (k_index_insert / k_index_match_delete: same arrays, same results).
//
// Serial semantics in a parallel insert: the reference stops at the first record (file order)
// whose name is already present.  Every thread that meets its own name in the table does
// atomicMin(slot, tag|me) and reports max(previous owner, me); the minimum over all reports is
// exactly the second-smallest index of the earliest repeated name, i.e. the record the serial
// loop would have stopped at.  The same argument gives the first unpaired record in the
// match-and-delete pass (claims[] holds the smallest file-2 record that asked for the entry).
#include "fqg_device.h"

namespace fqg {

constexpr unsigned long long kSlotEmpty = ~0ull;
constexpr unsigned long long kIdxMask = (1ull << 40) - 1;

struct IndexSeg {
  const uint8_t* img;
  const uint64_t* line_end;
  uint64_t nbytes;
  uint64_t n_records;
  uint64_t record_base;  // global index of the segment's first record
};

struct NameRec {
  unsigned long long n;  // length of the canonical name
  unsigned long long name[kNameInline / 8];  // its first 56 bytes, zero padded
};
static_assert(sizeof(NameRec) == 64, "a name record is one 64-byte line");

struct IndexView {
  unsigned long long* slots;
  NameRec* names;              // per inserted record (global index); null: not kept (an index nobody will ask)
  unsigned long long* claims;  // per inserted record, match-and-delete only (may be null)
  uint64_t mask;               // capacity - 1 (capacity is a power of two)
  uint64_t n_positional;       // match-and-delete: records 0 .. n_positional - 1 have name records and every name in the
                               // table is there once (no insert has reported a repeat or a header without '@'), so a
                               // record whose name is the asker's IS the asker's entry - see match_name; 0: always probe
  const IndexSeg* segs;
  int n_segs;
  int fmt, is_pe;  // read-name format / is_pe of the file the index was built from
  int may_have_nul;
};

struct IndexCall {
  unsigned long long first_dup;     // min over reports, kNoRecord if none (global record index)
  unsigned long long first_wrong;   // first record whose header does not start with '@' (frame-local)
  unsigned long long first_missing; // match-delete: first record without a partner (frame-local)
  unsigned long long inserted;      // names added
  unsigned long long matched;       // slots claimed for the first time
  unsigned long long name_bytes;    // sum of the `len` the reference accounts per name (src/fastq.c:609)
  unsigned long long seen;          // records the call looked at (a check on the enumeration by chunk: must be the frame's)
  unsigned long long captured;      // ... of them straight from a capture record
  unsigned int table_full;
  unsigned int build_overflow;      // table built in LDS (fqg_names_build_kernels.hip): a key bucket was too small
  unsigned long long spilled;       // ... keys that ran past the end of their part
};

__device__ __forceinline__ uint64_t mix_hash(uint64_t h, uint64_t w) {
  h ^= w;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return h;
}
// Canonical read name of the header line [line, line+len) (+'\n' when has_nl), after
// fastq_get_readname (reference src/fastq.c:488-512).  Returns its length; the name starts at
// line+1.  *acct is the `len` value the reference hands to new_indexentry (index_mem accounting).
__device__ inline uint32_t canon_name(const uint8_t* __restrict__ line, uint32_t len, uint32_t has_nl, int fmt,
                                      int is_pe, int may_have_nul, uint32_t* acct) {
  uint32_t cstr = len + has_nl;  // C-string length of the line
  if (may_have_nul)
    for (uint32_t i = 0; i < len; ++i)
      if (line[i] == 0) {
        cstr = i;
        break;
      }
  const uint32_t L = cstr > 0 ? cstr - 1 : 0;  // strlen(&hdr[1])
  if (fmt == FQG_NAME_CASAVA18) {
    uint32_t sp = L;
    for (uint32_t i = 0; i < L; ++i)
      if (line[1 + i] == ' ') {
        sp = i;
        break;
      }
    if (sp >= 2 && line[1 + sp - 2] == '/') sp -= 2;
    *acct = sp;
    return sp;
  }
  long l = (long)L;
  if (fmt == FQG_NAME_DEFAULT && is_pe) l--;
  *acct = (uint32_t)(l < 0 ? 0 : l);
  return l >= 1 ? (uint32_t)(l - 1) : L;
}

// a second, independent hash over the same 8-byte words (the fingerprints that travel between GPUs carry 23 of its
// bits next to the 64 of the first: fqg_fp kernels below)
__device__ __forceinline__ uint64_t mix_hash2(uint64_t h, uint64_t w) {
  h = (h ^ w) * 0xC2B2AE3D27D4EB4Full;
  h ^= h >> 31;
  return h;
}
__device__ __forceinline__ uint64_t fin_hash2(uint64_t h) {
  h *= 0x94D049BB133111EBull;
  h ^= h >> 29;
  return h;
}
__device__ __forceinline__ uint64_t hash_name(const uint8_t* __restrict__ p, uint32_t n, uint64_t* second = nullptr) {
  uint64_t h = name_seed(n), g = 0x9E3779B97F4A7C15ull + n;
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    __builtin_memcpy(&w, p + i, 8);
    h = name_word(h, w, i >> 3);
    g = mix_hash2(g, w);
  }
  uint64_t w = 0;
  for (uint32_t k = 0; i + k < n; ++k) w |= (uint64_t)p[i + k] << (8 * k);
  if (i < n) h = name_word(h, w, i >> 3);  // (only a word that holds a name byte)
  g = mix_hash2(g, w);
  if (second) *second = fin_hash2(g);
  return name_fin(h);
}

__device__ __forceinline__ bool same_bytes(const uint8_t* a, const uint8_t* b, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i)
    if (a[i] != b[i]) return false;
  return true;
}

// header line of the record with global index g, as kept by the index
__device__ inline bool stored_name(const IndexView& ix, uint64_t g, const uint8_t** name, uint32_t* n) {
  for (int s = 0; s < ix.n_segs; ++s) {
    const IndexSeg& sg = ix.segs[s];
    if (g >= sg.record_base && g < sg.record_base + sg.n_records) {
      const uint64_t r = g - sg.record_base;
      const uint64_t b = r == 0 ? 0 : sg.line_end[4 * r - 1] + 1;
      const uint64_t e = sg.line_end[4 * r];
      uint32_t acct;
      *n = canon_name(sg.img + b, (uint32_t)(e - b), e < sg.nbytes ? 1u : 0u, ix.fmt, ix.is_pe, ix.may_have_nul,
                      &acct);
      *name = sg.img + b + 1;
      return true;
    }
  }
  return false;
}

// ---- the header line in registers ---------------------------------------------------------------
// A thread that walks its header byte by byte pays a memory round trip per byte (64 lanes, 64
// different cache lines per load).  Instead the first kHdrBytes of the line are fetched with eight
// independent 16-byte loads, and name canonicalisation and hashing work on registers.  Lines that
// are longer, hold NUL bytes, or sit at the very end of the image take the byte-wise path.
constexpr int kHdrWords = 16;  // 64-bit words
constexpr uint32_t kHdrBytes = 8 * kHdrWords;
struct HdrRegs {
  uint64_t w[kHdrWords + 1];  // w[kHdrWords] = 0: the funnel shifts read one word ahead
};
// (only the 16-byte pieces that hold bytes of the line are fetched - `cstr` of them count; the words behind read as
// zero: a 45-byte header costs three loads, not eight, and a name kernel is made of exactly this traffic)
__device__ __forceinline__ void hdr_load(const uint8_t* __restrict__ line, HdrRegs& H, uint32_t cstr) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));
#pragma unroll
  for (int j = 0; j < kHdrWords / 2; ++j) {
    u32x4 v = {0, 0, 0, 0};
    if (16u * j < cstr) v = *reinterpret_cast<const u32x4*>(line + 16 * j);
    H.w[2 * j] = ((uint64_t)v.y << 32) | v.x;
    H.w[2 * j + 1] = ((uint64_t)v.w << 32) | v.z;
  }
  H.w[kHdrWords] = 0;
}
__device__ __forceinline__ uint64_t hdr_eq_mask(uint64_t x, uint8_t c) {  // 0x80 in every byte of x equal to c
  const uint64_t y = x ^ (0x0101010101010101ull * c);
  const uint64_t t = ((y & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | y;
  return ~(t | 0x7F7F7F7F7F7F7F7Full);
}
// canon_name on registers; cstr = C-string length of the line (no NUL inside), cstr <= kHdrBytes.
// *ok = false: a Casava header without its blank - left to the byte-wise path (no dynamic register
// indexing here: the byte two places before the blank is taken from the word the blank was found in).
__device__ __forceinline__ uint32_t canon_name_regs(const HdrRegs& H, uint32_t cstr, int fmt, int is_pe, uint32_t* acct,
                                                    bool* ok) {
  const uint32_t L = cstr > 0 ? cstr - 1 : 0;  // strlen(&hdr[1])
  *ok = true;
  if (fmt == FQG_NAME_CASAVA18) {
    uint32_t first = kHdrBytes;  // position in the line of the first blank at or after byte 1
    uint32_t two_before = 0;     // line[first - 2]
#pragma unroll
    for (int k = kHdrWords - 1; k >= 0; --k) {
      uint64_t m = hdr_eq_mask(H.w[k], (uint8_t)' ');
      if (k == 0) m &= ~0xFFull;  // byte 0 is the '@'
      if (m) {
        const uint32_t j = (uint32_t)__builtin_ctzll(m) >> 3;
        first = 8u * k + j;
        const uint64_t below = k ? H.w[k ? k - 1 : 0] : 0ull;
        two_before = (uint32_t)(j >= 2 ? H.w[k] >> (8 * (j - 2)) : below >> (8 * (6 + j))) & 0xFFu;
      }
    }
    if (first > L) {
      *ok = false;
      return 0;
    }
    uint32_t sp = first - 1;  // blank at line[1 + sp], sp < L
    if (sp >= 2 && two_before == '/') sp -= 2;
    *acct = sp;
    return sp;
  }
  long l = (long)L;
  if (fmt == FQG_NAME_DEFAULT && is_pe) l--;
  *acct = (uint32_t)(l < 0 ? 0 : l);
  return l >= 1 ? (uint32_t)(l - 1) : L;
}
// hash_name(line + 1, n) on registers: the same value, word for word
__device__ __forceinline__ uint64_t hash_name_regs(const HdrRegs& H, uint32_t n, uint64_t* second = nullptr) {
  uint64_t h = name_seed(n), g = 0x9E3779B97F4A7C15ull + n;
#pragma unroll
  for (int k = 0; k < kHdrWords; ++k) {
    const uint64_t nw = (H.w[k] >> 8) | (H.w[k + 1] << 56);  // bytes 8k .. 8k+7 of the name
    const bool full = 8u * k + 8u <= n;
    const bool last = !full && 8u * k <= n;  // the partial (possibly empty) word that ends the name
    const uint32_t rem = n - 8u * k;         // 0..7 when `last`
    const uint64_t w = full ? nw : (nw & ((1ull << (8 * (rem & 7))) - 1ull));
    if (full || last) {
      if (8u * k < n) h = name_word(h, w, (uint32_t)k);  // (only a word that holds a name byte)
      g = mix_hash2(g, w);
    }
  }
  if (second) *second = fin_hash2(g);
  return name_fin(h);
}
// name (length, hash) of the header line at img + b; the fast path when the line allows it
// (nm, optional: the first kNameInline bytes of the name as zero-padded words)
__device__ __forceinline__ uint32_t name_and_hash(const uint8_t* __restrict__ img, uint64_t nbytes, uint64_t b, uint64_t e,
                                                  int fmt, int is_pe, int may_have_nul, uint32_t* acct, uint64_t* h,
                                                  bool* at_sign, uint64_t* second = nullptr,
                                                  unsigned long long* nm = nullptr) {
  const uint8_t* line = img + b;
  const uint32_t len = (uint32_t)(e - b), has_nl = e < nbytes ? 1u : 0u;
  if (!may_have_nul && len + has_nl <= kHdrBytes - 1 && b + kHdrBytes <= nbytes) {
    HdrRegs H;
    hdr_load(line, H, len + has_nl);
    bool ok;
    const uint32_t n = canon_name_regs(H, len + has_nl, fmt, is_pe, acct, &ok);
    if (ok) {
      *at_sign = (H.w[0] & 0xFF) == '@';
      *h = hash_name_regs(H, n, second);
      if (nm) {
#pragma unroll
        for (uint32_t k = 0; k < kNameInline / 8; ++k) {
          const uint64_t nw = (H.w[k] >> 8) | (H.w[k + 1] << 56);  // bytes 8k .. 8k+7 of the name
          nm[k] = 8u * k + 8u <= n ? nw : (8u * k < n ? nw & ((1ull << (8 * (n - 8u * k))) - 1ull) : 0ull);
        }
      }
      return n;
    }
  }
  *at_sign = line[0] == '@';
  const uint32_t n = canon_name(line, len, has_nl, fmt, is_pe, may_have_nul, acct);
  *h = hash_name(line + 1, n, second);
  if (nm) {
    const uint32_t m = n < kNameInline ? n : kNameInline;
    for (uint32_t k = 0; k < kNameInline / 8; ++k) {
      unsigned long long w = 0;
      for (uint32_t i = 0; i < 8 && 8 * k + i < m; ++i) w |= (unsigned long long)line[1 + 8 * k + i] << (8 * i);
      nm[k] = w;
    }
  }
  return n;
}

// ---- names as keys ------------------------------------------------------------------------------
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));

struct NameKey {
  uint32_t n, acct;  // canonical name length; the `len` the reference accounts for it (src/fastq.c:609)
  uint64_t h;
  unsigned long long nm[kNameInline / 8];  // first 56 name bytes, zero padded
};

// Name of a captured header line (NameCapture record in w[]).  false: the record cannot give it - the line's end is
// not in the chunk, the line is longer than the record holds, a Casava header has no blank - and the caller goes
// through the line index instead.  *v = which newline of the chunk the line starts behind.
__device__ __forceinline__ bool name_from_record(const unsigned long long (&w)[kNameRecWords], int fmt, int is_pe, NameKey& k,
                                                 bool* at_sign, uint32_t* v) {
  const uint32_t meta = (uint32_t)w[0];
  const uint32_t L = meta & 1023u;  // strlen(&hdr[1]): the bytes behind the '@' and the '\n'
  *v = (meta >> 10) & 511u;
  *at_sign = ((meta >> 20) & 1u) != 0;
  const bool end_known = ((meta >> 19) & 1u) != 0;
  // (length unknown: only a Casava name, which ends at a blank, can still be read - when the record's bytes are the line's)
  if (end_known ? L >= (uint32_t)FQG_MAX_LABEL_LENGTH : !(((meta >> 21) & 1u) && fmt == FQG_NAME_CASAVA18)) return false;
  uint64_t t[kNameRecWords];  // the bytes behind the '@'
#pragma unroll
  for (uint32_t i = 0; i + 1 < kNameRecWords; ++i) t[i] = (w[i] >> 32) | (w[i + 1] << 32);
  t[kNameRecWords - 1] = w[kNameRecWords - 1] >> 32;
  uint32_t n, acct;
  if (fmt == FQG_NAME_CASAVA18) {
    const uint32_t lim = end_known && L < kNameRecText ? L : kNameRecText;
    uint32_t first = ~0u, two_before = 0;
#pragma unroll
    for (int i = (int)kNameRecWords - 1; i >= 0; --i) {
      const uint64_t m = hdr_eq_mask(t[i], (uint8_t)' ');
      if (m) {
        const uint32_t j = (uint32_t)__builtin_ctzll(m) >> 3;
        first = 8u * i + j;
        const uint64_t below = i ? t[i ? i - 1 : 0] : 0ull;
        two_before = (uint32_t)(j >= 2 ? t[i] >> (8 * (j - 2)) : below >> (8 * (6 + j))) & 0xFFu;
      }
    }
    if (first >= lim) return false;  // (no blank in the line, or none in what the record holds of it)
    uint32_t sp = first;
    if (sp >= 2 && two_before == '/') sp -= 2;
    n = acct = sp;
  } else {
    if (L > kNameRecText) return false;
    long l = (long)L;
    if (fmt == FQG_NAME_DEFAULT && is_pe) l--;
    acct = (uint32_t)(l < 0 ? 0 : l);
    n = l >= 1 ? (uint32_t)(l - 1) : L;
  }
  // hash_name(line + 1, n), word for word
  uint64_t h = name_seed(n);
#pragma unroll
  for (uint32_t i = 0; i < kNameRecWords; ++i) {
    const bool full = 8u * i + 8u <= n;
    const bool last = !full && 8u * i <= n;
    const uint32_t rem = n - 8u * i;
    const uint64_t x = full ? t[i] : (t[i] & ((1ull << (8 * (rem & 7))) - 1ull));
    if (8u * i < n) h = name_word(h, x, i);
    if (i < kNameInline / 8) k.nm[i] = (full || last) ? x : 0ull;
  }
  k.h = name_fin(h);
  k.n = n;
  k.acct = acct;
  return true;
}

// where the name of frame-local record r starts in the image (looked up only when bytes must be compared)
struct NameAt {
  const FrameView& f;
  uint64_t r;
  __device__ __forceinline__ const uint8_t* operator()() const { return f.img + (r == 0 ? 0 : f.line_end[4 * r - 1] + 1) + 1; }
};

// the same key from the image, through the line index: any line the capture could not give
__device__ inline void name_from_image(const FrameView& f, uint64_t r, int fmt, int is_pe, int may_have_nul, NameKey& k,
                                       bool* at_sign) {
  const uint64_t b = r == 0 ? 0 : f.line_end[4 * r - 1] + 1;
  const uint64_t e = f.line_end[4 * r];
  k.n = name_and_hash(f.img, f.nbytes, b, e, fmt, is_pe, may_have_nul, &k.acct, &k.h, at_sign, nullptr, k.nm);
}

// Is the canonical name of the stored record g the n bytes at `mine`?  Byte-wise, through the images: names beyond 56
// bytes, tag matches during an insert, indexes that keep no name records.
__device__ inline bool stored_name_is(const IndexView& ix, uint64_t g, const uint8_t* __restrict__ mine, uint32_t n) {
  const uint8_t* other;
  uint32_t on;
  return stored_name(ix, g, &other, &on) && on == n && same_bytes(other, mine, n);
}

struct IndexTally {  // per-thread findings and counts of the name kernels
  unsigned long long first_dup = kNoRecord, first_wrong = kNoRecord, first_missing = kNoRecord;
  unsigned long long inserted = 0, matched = 0, name_bytes = 0, seen = 0, captured = 0;
};

// Insert the name of frame-local record r (global record_base + r), report a repeat.
__device__ __forceinline__ void insert_name(const IndexView& ix, const NameKey& k, bool at_sign, const FrameView& f, uint64_t r,
                                            uint64_t record_base, IndexTally& t, IndexCall* __restrict__ call) {
  ++t.seen;
  const unsigned long long g = record_base + r;
  if (ix.names) {
    // the record's name for later look-ups: neighbouring lanes, neighbouring 64-byte lines (a record without '@' or
    // with a repeated name gets one too - nothing will ever point at it)
    u64x2_t* dst = reinterpret_cast<u64x2_t*>(ix.names + g);
    u64x2_t x;
    x.x = k.n;
    x.y = k.nm[0];
    __builtin_nontemporal_store(x, dst);
#pragma unroll
    for (uint32_t i = 1; i < 4; ++i) {
      x.x = k.nm[2 * i - 1];
      x.y = k.nm[2 * i];
      __builtin_nontemporal_store(x, dst + i);
    }
  }
  if (!at_sign) {  // fastq_get_readname refuses it (src/fastq.c:448)
    t.first_wrong = r < t.first_wrong ? r : t.first_wrong;
    return;
  }
  const unsigned long long me = ((k.h >> 40) << 40) | g;
  uint64_t at = k.h & ix.mask;
  for (uint64_t probes = 0; probes <= ix.mask; ++probes, at = (at + 1) & ix.mask) {
    // one round trip when the slot is free (most are: the table is at most half full)
    const unsigned long long cur = atomicCAS(&ix.slots[at], kSlotEmpty, me);
    if (cur == kSlotEmpty) {
      ++t.inserted;
      t.name_bytes += k.acct;
      return;
    }
    // (the other record's name record may not be written yet - it belongs to this very launch: compare on the images)
    if ((cur >> 40) == (me >> 40) && stored_name_is(ix, cur & kIdxMask, NameAt{f, r}(), k.n)) {
      const unsigned long long prev = atomicMin(&ix.slots[at], me);
      const unsigned long long late = (prev & kIdxMask) > g ? (prev & kIdxMask) : g;
      t.first_dup = late < t.first_dup ? late : t.first_dup;
      return;
    }
  }
  atomicOr(&call->table_full, 1u);
}

// Find the name of frame-local record r of the asking file and take its entry: slot -> the stored record's name
// record (bytes decide) -> atomicMin on its claim.  found[r] (optional) = the record whose entry was found
// (kSlotEmpty: none, kSlotEmpty - 1: no '@').
__device__ __forceinline__ void match_name(const IndexView& ix, const NameKey& k, bool at_sign, const FrameView& f, uint64_t r,
                                           uint64_t asker_base, unsigned long long* __restrict__ found, IndexTally& t) {
  ++t.seen;
  if (found) found[r] = at_sign ? kSlotEmpty : kSlotEmpty - 1;
  if (!at_sign) {
    t.first_wrong = r < t.first_wrong ? r : t.first_wrong;
    return;
  }
  const unsigned long long g2 = asker_base + r;
  auto record_has_name = [&](unsigned long long g) {
    const u64x2_t* src = reinterpret_cast<const u64x2_t*>(ix.names + g);
    const u64x2_t a = src[0], b = src[1], c = src[2], d = src[3];
    bool same = a.x == k.n && !((a.y ^ k.nm[0]) | (b.x ^ k.nm[1]) | (b.y ^ k.nm[2]) | (c.x ^ k.nm[3]) | (c.y ^ k.nm[4]) |
                                (d.x ^ k.nm[5]) | (d.y ^ k.nm[6]));
    if (same && k.n > kNameInline) same = stored_name_is(ix, g, NameAt{f, r}(), k.n);
    return same;
  };
  auto take = [&](unsigned long long g) {
    // the smallest asker gets the entry; every other asker is what the serial loop would have found missing
    // after the delete.  An asker of an EARLIER piece is smaller than every record of this one, so `late`
    // always lies in this piece.
    const unsigned long long prev = atomicMin(&ix.claims[g], g2);
    if (prev == kSlotEmpty) ++t.matched;
    else {
      const unsigned long long late = (prev > g2 ? prev : g2) - asker_base;
      t.first_missing = late < t.first_missing ? late : t.first_missing;
    }
    if (found) found[r] = g;
  };
  // The mate files of a sequencing run hold their reads in the same order: the partner of record i is record i.  Its
  // name record lies next to the ones the neighbouring lanes ask for - a sequential read - where the way through the
  // table is a load from a random slot first.  The bytes decide here as they do there, and while every name of the
  // table is in it once (n_positional), the record that has the asker's name is the entry the table would have led
  // to; an asker whose name is not at its own place goes through the table.
  if (g2 < ix.n_positional && record_has_name(g2)) {
    take(g2);
    return;
  }
  uint64_t at = k.h & ix.mask;
  for (uint64_t probes = 0; probes <= ix.mask; ++probes, at = (at + 1) & ix.mask) {
    const unsigned long long cur = ix.slots[at];
    if (cur == kSlotEmpty) break;
    if ((cur >> 40) != (k.h >> 40)) continue;
    const unsigned long long g = cur & kIdxMask;
    if (!(ix.names ? record_has_name(g) : stored_name_is(ix, g, NameAt{f, r}(), k.n))) continue;
    take(g);
    return;
  }
  t.first_missing = r < t.first_missing ? r : t.first_missing;
}

// fold a thread's tally into the call's scalars: minima straight away (rare), sums once per workgroup (the wavefronts of
// a grid finish together: tens of thousands of adds to a few addresses)
__device__ __forceinline__ void tally_flush(IndexTally& t, IndexCall* __restrict__ call) {
  if (t.first_dup != kNoRecord) atomicMin(&call->first_dup, t.first_dup);
  if (t.first_wrong != kNoRecord) atomicMin(&call->first_wrong, t.first_wrong);
  if (t.first_missing != kNoRecord) atomicMin(&call->first_missing, t.first_missing);
  unsigned long long v[5] = {t.inserted, t.matched, t.name_bytes, t.seen, t.captured};
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] += __shfl_down(v[i], d, 64);
  __shared__ unsigned long long s_sum[kBlock / kWave][5];
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 5; ++i) s_sum[threadIdx.x >> 6][i] = v[i];
  __syncthreads();
  if (threadIdx.x < 5) {
    unsigned long long a = 0;
    for (int w = 0; w < kBlock / kWave; ++w) a += s_sum[w][threadIdx.x];
    unsigned long long* dst = threadIdx.x == 0   ? &call->inserted
                              : threadIdx.x == 1 ? &call->matched
                              : threadIdx.x == 2 ? &call->name_bytes
                              : threadIdx.x == 3 ? &call->seen
                                                 : &call->captured;
    if (a) atomicAdd(dst, a);
  }
}

// ---- through the line index: one thread per record of the frame (frames that were not streamed, re-inserts) ----
__global__ __launch_bounds__(kBlock) void k_index_insert(FrameView f, IndexView ix, uint64_t record_base,
                                                         IndexCall* __restrict__ call) {
  IndexTally t;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < f.n_records; r += stride) {
    NameKey k;
    bool at_sign;
    name_from_image(f, r, ix.fmt, ix.is_pe, ix.may_have_nul, k, &at_sign);
    insert_name(ix, k, at_sign, f, r, record_base, t, call);
  }
  tally_flush(t, call);
}

// One thread per record of the (file-2) frame: find the name, claim its entry.  claim holds the GLOBAL index
// (asker_base + r: the pieces of the asking file share the index) of the smallest asker; found[r] (optional)
// remembers whose entry record r found (kSlotEmpty: nobody's) for k_index_probe_resolve.
__global__ __launch_bounds__(kBlock) void k_index_match_delete(FrameView f, IndexView ix, int fmt2, int is_pe2,
                                                               int may_have_nul2, uint64_t asker_base,
                                                               unsigned long long* __restrict__ found,
                                                               IndexCall* __restrict__ call) {
  IndexTally t;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < f.n_records; r += stride) {
    NameKey k;
    bool at_sign;
    name_from_image(f, r, fmt2, is_pe2, may_have_nul2, k, &at_sign);
    match_name(ix, k, at_sign, f, r, asker_base, found, t);
  }
  tally_flush(t, call);
}

// ---- names by POSITION --------------------------------------------------------------------------------------------
// The mate files of a run hold their reads in one order.  Before the names of a pair of files that lie spread over
// several contexts are sent to their owners by hash (fqg_fp kernels, host/fq_names_multi.h), the contexts look whether
// record i of file 2 simply has the name of record i of file 1: the name records of a range of file 1 are written out
// (k_names_records), copied to the context that holds the same range of file 2 - a contiguous copy - and compared
// there (k_names_equal).  When every record of file 2 has its partner at its own place, file 1 holds as many records,
// and no name of file 1 occurs twice (the duplicate test of file 1 has passed), the serial loop of the reference
// (src/fastq_info.c:333-356) would have found and deleted every entry: nothing is left to exchange.
constexpr unsigned long long kNameRecNoAt = 1ull << 63;  // in NameRec::n: the header does not start with '@'
__global__ __launch_bounds__(kBlock) void k_names_records(FrameView f, int fmt, int is_pe, int may_have_nul, uint64_t first,
                                                          uint64_t n, NameRec* __restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    NameKey k;
    bool at_sign;
    name_from_image(f, first + i, fmt, is_pe, may_have_nul, k, &at_sign);
    u64x2_t* dst = reinterpret_cast<u64x2_t*>(out + i);
    u64x2_t x;
    x.x = (unsigned long long)k.n | (at_sign ? 0ull : kNameRecNoAt);
    x.y = k.nm[0];
    dst[0] = x;
#pragma unroll
    for (uint32_t j = 1; j < 4; ++j) {
      x.x = k.nm[2 * j - 1];
      x.y = k.nm[2 * j];
      dst[j] = x;
    }
  }
}
// counts[0] += records whose name is the record's at the same place; counts[1] += records that agree in length and in
// their first 56 bytes but are longer than that (the caller decides them the long way)
__global__ __launch_bounds__(kBlock) void k_names_equal(FrameView f, int fmt, int is_pe, int may_have_nul, uint64_t first,
                                                        uint64_t n, const NameRec* __restrict__ recs,
                                                        unsigned long long* __restrict__ counts) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  unsigned long long same = 0, open = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    NameKey k;
    bool at_sign;
    name_from_image(f, first + i, fmt, is_pe, may_have_nul, k, &at_sign);
    const u64x2_t* src = reinterpret_cast<const u64x2_t*>(recs + i);
    const u64x2_t a = src[0], b = src[1], c = src[2], d = src[3];
    const bool eq = at_sign && a.x == (unsigned long long)k.n &&
                    !((a.y ^ k.nm[0]) | (b.x ^ k.nm[1]) | (b.y ^ k.nm[2]) | (c.x ^ k.nm[3]) | (c.y ^ k.nm[4]) | (d.x ^ k.nm[5]) |
                      (d.y ^ k.nm[6]));
    if (eq && k.n <= kNameInline) ++same;
    else if (eq) ++open;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    same += __shfl_down(same, o, 64);
    open += __shfl_down(open, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (same) atomicAdd(&counts[0], same);
    if (open) atomicAdd(&counts[1], open);
  }
}

// ---- from the capture records of the streaming pass ----------------------------------------------
// One thread per record SLOT (chunk c, ordinal j < K).  A chunk whose speculated line type was the true one and that
// saw at most K headers is taken from its records; the headers of every other chunk are enumerated by rank - the
// records whose header starts behind one of the chunk's newlines (or behind the last byte of the chunk before) - and
// read through the line index.  Every record of the frame is met exactly once (the call's `seen` is checked).
struct NamesView {
  const unsigned long long* recs;
  const uint16_t* hcount;
  const uint32_t* cinfo;
  ChunkRanks cr;
  uint32_t K, k_shift;  // K = 1 << k_shift record slots per chunk
  uint32_t digests;     // the slots hold 16-byte digests made under (fmt, is_pe), not 64-byte records (fqg_device.h)
  int fmt, is_pe;
  // what k_names_pass leaves to k_names_rest - written with plain stores, one owner per word (a list with ONE counter
  // was the whole cost of this pass: a million appends to one address take 12 ms):
  unsigned long long* redo_bits;  // per 64 record slots: the slots whose record cannot give the name
  uint8_t* chunk_redo;            // per chunk: 1 = its records cannot be trusted, its headers are enumerated by rank
  int ablate;  // measurement only (FQGPU_NAMES_ABL): 1 = no table access, 8 = no decoding
};

template <bool MATCH, bool NT = true, bool DIGEST = false>
__global__ __launch_bounds__(kBlock) void k_names_pass(FrameView f, NamesView nv, IndexView ix, int fmt, int is_pe,
                                                       uint64_t base /* record_base or asker_base */,
                                                       unsigned long long* __restrict__ found,
                                                       IndexCall* __restrict__ call) {
  IndexTally t;
  const uint64_t n_slots = (uint64_t)nv.cr.n_chunks << nv.k_shift;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const uint32_t lane = threadIdx.x & 63u;
  // a wavefront owns 64 consecutive slots per step (K is a power of two >= 8: whole chunks or an aligned part of one)
  for (uint64_t s0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); s0 < n_slots; s0 += stride) {
    const uint64_t s = s0 + lane;
    const uint32_t c = (uint32_t)(s >> nv.k_shift), j = (uint32_t)s & (nv.K - 1u);
    bool live = s < n_slots, redo = false;
    uint64_t r = 0;
    NameKey k;
    bool at_sign = false;
    if (live) {
      const uint32_t hc = nv.hcount[c], info = nv.cinfo[c];
      const uint64_t rank0 = nv.cr.rank0(c);
      const bool trusted = hc != kNoCapture && hc <= nv.K && !(info & (kInfoUnknown | kInfoOneLine)) && (info & 3u) == ((uint32_t)rank0 & 3u);
      if (j == 0) nv.chunk_redo[c] = trusted ? 0 : 1;
      live = trusted && j < hc;
      if (live && DIGEST) {
        // the streaming pass has done the work: hash, lengths, '@' (an index without name records asks for no bytes)
        const u64x2_t* src = reinterpret_cast<const u64x2_t*>(nv.recs + s * kDigestWords);
        const u64x2_t x = NT ? __builtin_nontemporal_load(src) : src[0];
        const uint32_t meta = (uint32_t)x.y;
        k.h = x.x;
        k.n = meta & 1023u;
        k.acct = (meta >> 10) & 1023u;
        at_sign = (meta & kDigestAt) != 0;
        r = (rank0 + ((meta >> 20) & 511u)) >> 2;
        if (r >= f.n_records) live = false;
        else if (!(meta & kDigestOk)) {
          redo = true;
          live = false;
        }
      } else if (live) {
        unsigned long long w[kNameRecWords];
        const u64x2_t* src = reinterpret_cast<const u64x2_t*>(nv.recs + s * kNameRecWords);
#pragma unroll
        for (uint32_t i = 0; i < kNameRecWords / 2; ++i) {
          const u64x2_t x = NT ? __builtin_nontemporal_load(src + i) : src[i];
          w[2 * i] = x.x;
          w[2 * i + 1] = x.y;
        }
        uint32_t v;
        bool ok;
        if (nv.ablate & 8) {
          ok = true;
          v = ((uint32_t)w[0] >> 10) & 511u;
          k.h = mix_hash(w[1] ^ w[3], w[2] ^ w[4]);
          k.n = k.acct = 20;
          at_sign = true;
#pragma unroll
          for (uint32_t i = 0; i < kNameInline / 8; ++i) k.nm[i] = w[i + 1];
        } else ok = name_from_record(w, fmt, is_pe, k, &at_sign, &v);
        r = (rank0 + v) >> 2;  // (rank0 + v = 4 r: the line behind newline 4 r - 1)
        if (r >= f.n_records) live = false;  // a header of the incomplete tail
        else if (!ok) {
          redo = true;
          live = false;
        }
      }
    }
    const unsigned long long rm = __ballot(redo);
    if (lane == 0) nv.redo_bits[s0 >> 6] = rm;
    if (live) {
      ++t.captured;
      if (nv.ablate & 1) t.seen += (k.h & 1) + 1;
      else if (MATCH) match_name(ix, k, at_sign, f, r, base, found, t);
      else insert_name(ix, k, at_sign, f, r, base, t, call);
    }
  }
  tally_flush(t, call);
}

// What k_names_pass left, all of it through the line index: the flagged record slots, and the records whose header
// starts in a flagged chunk - those r with 4 r in [rank0, rank0 + count], i.e. behind one of the chunk's newlines or
// behind the last byte of the chunk before; the one that starts in THIS chunk is this chunk's.
template <bool MATCH>
__global__ __launch_bounds__(kBlock) void k_names_rest(FrameView f, NamesView nv, IndexView ix, int fmt, int is_pe,
                                                       uint64_t base, unsigned long long* __restrict__ found,
                                                       IndexCall* __restrict__ call) {
  IndexTally t;
  const uint64_t n_slots = (uint64_t)nv.cr.n_chunks << nv.k_shift, n_words = (n_slots + 63) >> 6;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock, tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  auto one = [&](uint64_t r) {
    NameKey k;
    bool at_sign;
    name_from_image(f, r, fmt, is_pe, 0, k, &at_sign);
    if (MATCH) match_name(ix, k, at_sign, f, r, base, found, t);
    else insert_name(ix, k, at_sign, f, r, base, t, call);
  };
  for (uint64_t wd = tid; wd < n_words; wd += stride) {
    unsigned long long m = nv.redo_bits[wd];
    while (m) {
      const uint64_t s = (wd << 6) + (uint64_t)__builtin_ctzll(m);
      m &= m - 1;
      const uint32_t c = (uint32_t)(s >> nv.k_shift);
      const uint32_t v = nv.digests ? ((uint32_t)nv.recs[s * kDigestWords + 1] >> 20) & 511u
                                    : ((uint32_t)nv.recs[s * kNameRecWords] >> 10) & 511u;
      one((nv.cr.rank0(c) + v) >> 2);
    }
  }
  // a flagged chunk: 16 lanes share its candidates.  The flags are looked at 16 bytes at a time (a group of 16 lanes
  // walks 16 chunks per load: a byte per load was 0.2 ms of dependent round trips for the handful of chunks an ordinary
  // file flags); the flag array is padded to a multiple of 16 by names_prepare
  const uint64_t n_groups = ((uint64_t)nv.cr.n_chunks + 15) >> 4;
  for (uint64_t g = tid >> 4; g < n_groups; g += stride >> 4) {
    const u64x2_t fl = *reinterpret_cast<const u64x2_t*>(nv.chunk_redo + 16 * g);
    if (!(fl.x | fl.y)) continue;
    for (uint32_t q = 0; q < 16; ++q) {
      const uint64_t c = 16 * g + q;
      if (c >= nv.cr.n_chunks || !((q < 8 ? fl.x >> (8 * q) : fl.y >> (8 * (q - 8))) & 0xFFu)) continue;
      const uint64_t rank0 = nv.cr.rank0((uint32_t)c);
      const uint64_t r_lo = (rank0 + 3) >> 2, r_hi = (rank0 + nv.cr.counts[c]) >> 2;
      for (uint64_t r = r_lo + (tid & 15u); r <= r_hi && r < f.n_records; r += 16) {
        const uint64_t start = r == 0 ? 0 : f.line_end[4 * r - 1] + 1;
        if (start / kChunkBytes == c) one(r);
      }
    }
  }
  tally_flush(t, call);
}

// After a matching pass: match[r] = global index (insertion order) of the entry record r took, or kNoRecord
// when its name is not in the index or an earlier asker took it (src/fastq_filterpair.c:150-170: lookup, then
// fastq_index_delete)
__global__ __launch_bounds__(kBlock) void k_index_probe_resolve(uint64_t n, const unsigned long long* __restrict__ found,
                                                                IndexView ix, uint64_t asker_base,
                                                                unsigned long long* __restrict__ match) {
  const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= n) return;
  const unsigned long long g = found[r];
  if (g >= kSlotEmpty - 1) match[r] = g;  // not in the index / no '@' (FQG_NO_MATCH / FQG_MATCH_WRONG_HEADER)
  else match[r] = ix.claims[g] == asker_base + r ? g : kNoRecord;
}
// alive[g] = 1 for every inserted record g whose entry nobody has taken (the table still holds it)
__global__ __launch_bounds__(kBlock) void k_index_alive(IndexView ix, uint64_t n_records, uint8_t* __restrict__ alive) {
  const uint64_t at = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (at > ix.mask) return;
  const unsigned long long cur = ix.slots[at];
  if (cur == kSlotEmpty) return;
  const unsigned long long g = cur & kIdxMask;
  if (g < n_records) alive[g] = (!ix.claims || ix.claims[g] == kSlotEmpty) ? 1 : 0;
}

// Names of paired records must be equal: record 2k against 2k+1 of one frame (interleaved
// input, src/fastq_info.c:81-91) or record k of frame A against record k of frame B (files with
// the same ordering, src/fastq_info.c:133-138).  Reports the first pair that differs.
__global__ __launch_bounds__(kBlock) void k_names_compare(FrameView a, int fmt_a, int pe_a, FrameView b2, int fmt_b,
                                                          int pe_b, int interleaved, int may_have_nul,
                                                          uint64_t n_pairs, IndexCall* __restrict__ call) {
  unsigned long long my_bad = kNoRecord, my_wrong = kNoRecord;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < n_pairs; k += stride) {
    const uint64_t ra = interleaved ? 2 * k : k, rb = interleaved ? 2 * k + 1 : k;
    const FrameView& fb = interleaved ? a : b2;
    const uint64_t ba = ra == 0 ? 0 : a.line_end[4 * ra - 1] + 1, ea = a.line_end[4 * ra];
    const uint64_t bb = rb == 0 ? 0 : fb.line_end[4 * rb - 1] + 1, eb = fb.line_end[4 * rb];
    const uint8_t *la = a.img + ba, *lb = fb.img + bb;
    const uint32_t len_a = (uint32_t)(ea - ba), nl_a = ea < a.nbytes ? 1u : 0u;
    const uint32_t len_b = (uint32_t)(eb - bb), nl_b = eb < fb.nbytes ? 1u : 0u;
    uint32_t acct;
    if (!may_have_nul && len_a + nl_a <= kHdrBytes - 1 && ba + kHdrBytes <= a.nbytes && len_b + nl_b <= kHdrBytes - 1 &&
        bb + kHdrBytes <= fb.nbytes) {
      // both headers in registers: no memory round trip per byte
      HdrRegs A, B;
      hdr_load(la, A, len_a + nl_a);
      hdr_load(lb, B, len_b + nl_b);
      bool ok_a, ok_b;
      const uint32_t na = canon_name_regs(A, len_a + nl_a, fmt_a, pe_a, &acct, &ok_a);
      const uint32_t nb = canon_name_regs(B, len_b + nl_b, fmt_b, pe_b, &acct, &ok_b);
      if (ok_a && ok_b) {
        if ((A.w[0] & 0xFF) != '@' || (B.w[0] & 0xFF) != '@') {
          my_wrong = k < my_wrong ? k : my_wrong;
          continue;
        }
        uint64_t diff = 0;
#pragma unroll
        for (int w = 0; w < kHdrWords; ++w) {
          const uint64_t x = ((A.w[w] >> 8) | (A.w[w + 1] << 56)) ^ ((B.w[w] >> 8) | (B.w[w + 1] << 56));
          const uint64_t m = 8u * w + 8u <= na ? ~0ull : (8u * w < na ? (1ull << (8 * (na - 8u * w))) - 1ull : 0ull);
          diff |= x & m;
        }
        if (na != nb || diff) my_bad = k < my_bad ? k : my_bad;
        continue;
      }
    }
    if (la[0] != '@' || lb[0] != '@') {
      // src/fastq.c:448; for interleaved input both names are taken before anything else,
      // for same-order files each record has already passed validation at this point
      my_wrong = k < my_wrong ? k : my_wrong;
      continue;
    }
    const uint32_t na = canon_name(la, len_a, nl_a, fmt_a, pe_a, may_have_nul, &acct);
    const uint32_t nb = canon_name(lb, len_b, nl_b, fmt_b, pe_b, may_have_nul, &acct);
    if (na != nb || !same_bytes(la + 1, lb + 1, na)) my_bad = k < my_bad ? k : my_bad;
  }
  if (my_bad != kNoRecord) atomicMin(&call->first_missing, my_bad);
  if (my_wrong != kNoRecord) atomicMin(&call->first_wrong, my_wrong);
}

// ------------------------------------------------------------------------------------------
// read names across GPUs: every record's canonical name becomes a 64-bit fingerprint
// + its GLOBAL record index (16 bytes); fingerprints travel to an owner rank (all-to-all over
// RCCL), whose set keeps the smallest index per fingerprint.  A record that is not the earliest
// holder of its fingerprint is a CANDIDATE duplicate; candidates are confirmed on the name bytes
// by their home ranks, so a hash collision can neither fake nor hide a duplicate.
// ------------------------------------------------------------------------------------------
struct FpRec {
  unsigned long long fp, idx;
};
constexpr int kMaxOwners = 64;
// idx word of an exchanged record: bit 63 = file 2 (kFpFile2), bits 40..62 = 23 bits of the second hash of the
// name, bits 0..39 = global record index.  A run of one holder and one asker only counts as a pair when those
// 23 bits agree too: 87 bits in all; anything else is resolved on the name bytes.
constexpr unsigned long long kFpIndexMask = (1ull << 40) - 1;
constexpr int kFpCheckShift = 40;
constexpr unsigned long long kFpCheckMask = ((1ull << 23) - 1) << kFpCheckShift;

__device__ __forceinline__ uint32_t fp_owner(unsigned long long fp, uint32_t n_owners) {
  return (uint32_t)(((fp >> 32) * (unsigned long long)n_owners) >> 32);  // high bits: the table uses the low ones
}

// pass 0: count per owner; pass 1: write into the owner's bucket (cursor[] starts at the bucket offsets)
constexpr int kFpPerThread = 8;  // records per thread: a workgroup reserves its bucket space once per 2048 records
// NAMED (pass 1 of a pairing): the name itself travels beside every pair - a 64-byte record [56 name bytes, zero padded |
// name length] at the pair's place in a second array - so that the owner can hold a holder's name against its
// asker's BYTES (k_fp_pair_runs).  A name of more than 56 bytes says so by its length: such runs go to the resolution
// by record index, which fetches whole names.
constexpr uint32_t kFpNameWords = 8;  // 64-bit words per name record
template <int PASS, bool NAMED = false>
__global__ __launch_bounds__(kBlock) void k_names_fingerprint(FrameView f, int fmt, int is_pe, int may_have_nul,
                                                              uint64_t record_base, uint32_t n_owners,
                                                              unsigned long long* __restrict__ cursor,
                                                              FpRec* __restrict__ out,
                                                              unsigned long long* __restrict__ names_out = nullptr,
                                                              int weak_bits = 0) {
  __shared__ unsigned int s_cnt[kMaxOwners];
  __shared__ unsigned long long s_base[kMaxOwners];
  if (threadIdx.x < kMaxOwners) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  FpRec me[kFpPerThread];
  uint32_t owner[kFpPerThread], slot[kFpPerThread];
  bool have[kFpPerThread];
  unsigned long long name_bytes = 0;  // PASS 0: sum of the `len` the reference accounts per name (src/fastq.c:609) -> cursor[kMaxOwners]
#pragma unroll
  for (int j = 0; j < kFpPerThread; ++j) {
    const uint64_t r = ((uint64_t)blockIdx.x * kFpPerThread + j) * kBlock + threadIdx.x;
    have[j] = false;
    owner[j] = slot[j] = 0;
    me[j] = FpRec{0, 0};
    if (r < f.n_records) {
      const uint64_t b = r == 0 ? 0 : f.line_end[4 * r - 1] + 1;
      const uint64_t e = f.line_end[4 * r];
      uint32_t acct;
      uint64_t h64, h2;
      bool at_sign;
      (void)name_and_hash(f.img, f.nbytes, b, e, fmt, is_pe, may_have_nul, &acct, &h64, &at_sign, &h2);
      if (at_sign) {  // (a wrong header is the local pass's finding, src/fastq.c:448)
        if (PASS == 0) name_bytes += acct;
        unsigned long long h = h64;
        if (weak_bits) {  // (tests, FQGPU_FP_WEAK_BITS: fingerprints of a few bits, no check bits - collisions galore)
          h &= (1ull << weak_bits) - 1ull;
          h2 = 0;
        }
        if (h >= kSlotEmpty - 1) h = kSlotEmpty - 2;
        me[j].fp = h;
        me[j].idx = (record_base + r) | ((h2 << kFpCheckShift) & kFpCheckMask);
        owner[j] = fp_owner(h, n_owners);
        slot[j] = atomicAdd(&s_cnt[owner[j]], 1u);
        have[j] = true;
      }
    }
  }
  __syncthreads();
  // one reservation per owner and workgroup (many more of them on one address would cost more than the hashing)
  if (threadIdx.x < n_owners && s_cnt[threadIdx.x])
    s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
  if (PASS == 0) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) name_bytes += __shfl_down(name_bytes, d, 64);
    if ((threadIdx.x & 63) == 0 && name_bytes) atomicAdd(&cursor[kMaxOwners], name_bytes);
    return;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kFpPerThread; ++j)
    if (have[j]) out[s_base[owner[j]] + slot[j]] = me[j];
  if (NAMED) {  // (the names once more: eight of them would not stay in registers beside the pairs)
#pragma unroll 1
    for (int j = 0; j < kFpPerThread; ++j) {
      if (!have[j]) continue;
      const uint64_t r = ((uint64_t)blockIdx.x * kFpPerThread + j) * kBlock + threadIdx.x;
      const uint64_t b = r == 0 ? 0 : f.line_end[4 * r - 1] + 1;
      const uint64_t e = f.line_end[4 * r];
      uint32_t acct;
      uint64_t h64;
      bool at_sign;
      unsigned long long nm[kNameInline / 8];
      const uint32_t n = name_and_hash(f.img, f.nbytes, b, e, fmt, is_pe, may_have_nul, &acct, &h64, &at_sign, nullptr, nm);
      unsigned long long* dst = names_out + (s_base[owner[j]] + slot[j]) * kFpNameWords;
#pragma unroll
      for (uint32_t k = 0; k < kNameInline / 8; ++k) dst[k] = nm[k];
      dst[kFpNameWords - 1] = n;
    }
  }
}

// Owner side: the received pairs are radix-sorted by fingerprint (rocPRIM, stable); equal
// fingerprints are then neighbours.  (An open-addressing set with CAS + 64-bit atomicMin per pair
// was measured at 0.57-1.1 s for 100 M pairs; the sort needs no atomics at all.)
// For every run of equal fingerprints: (smallest index of the run, every other index of the run).
__global__ __launch_bounds__(kBlock) void k_fp_runs(const unsigned long long* __restrict__ fp,
                                                    const unsigned long long* __restrict__ idx, uint64_t n,
                                                    unsigned long long* __restrict__ pairs, unsigned long long cap,
                                                    unsigned long long* __restrict__ count) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const unsigned long long f = fp[i];
  if (i > 0 && fp[i - 1] == f) return;          // not the start of a run
  if (i + 1 >= n || fp[i + 1] != f) return;     // a run of one: the common case
  const unsigned long long keep = ~kFpCheckMask;  // (the check bits are not part of the index)
  unsigned long long mn = idx[i] & keep;
  uint64_t e = i + 1;
  for (; e < n && fp[e] == f; ++e) mn = (idx[e] & keep) < mn ? (idx[e] & keep) : mn;
  for (uint64_t k = i; k < e; ++k) {
    if ((idx[k] & keep) == mn) continue;
    const unsigned long long at = atomicAdd(count, 1ull);
    if (at < cap) {
      pairs[2 * at] = mn;
      pairs[2 * at + 1] = idx[k] & keep;
    }
  }
}

// Pairing across GPUs (fastq_info's file-2 loop, reference src/fastq_info.c:333-356): the owner
// receives the fingerprints of BOTH files, file-2 entries carrying kFpFile2 in their index.  After the sort
// by fingerprint every run is one name (up to hash collisions).  Per run of h file-1 holders and a file-2
// askers:   (1, 1)  a pair                        (h, 0)  h names nobody asked for (left over at the end)
//           (0, a)  a askers without a holder: all unpaired; the serial loop stops at the smallest
//           anything else (an asker met twice, or a collision): left to the exact resolution on the name
//           bytes - its entries are exported as (run, index) pairs.
constexpr unsigned long long kFpFile2 = 1ull << 63;
struct FpPairSummary {
  unsigned long long matched, leftover, unpaired, first_unpaired, n_complex;
};
// With names (names != nullptr): val[] holds ARRIVAL POSITIONS, idx_of[] and names[] are in arrival order, and a run of
// one holder and one asker is a pair only when the two names are the same BYTES (the strcmp behind the key match of
// src/fastq.c:577-587); names too long for their record leave the run to the resolution by record index.
__global__ __launch_bounds__(kBlock) void k_fp_pair_runs(const unsigned long long* __restrict__ fp,
                                                         const unsigned long long* __restrict__ val, uint64_t n,
                                                         unsigned long long* __restrict__ entries, unsigned long long cap,
                                                         FpPairSummary* __restrict__ sum,
                                                         const unsigned long long* __restrict__ idx_of = nullptr,
                                                         const unsigned long long* __restrict__ names = nullptr) {
  __shared__ unsigned long long s_acc[kBlock / kWave][4];
  unsigned long long matched = 0, leftover = 0, unpaired = 0, first = ~0ull;
  auto index_at = [&](uint64_t e) { return names ? idx_of[val[e]] : val[e]; };
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const unsigned long long f = fp[i];
    if (i > 0 && fp[i - 1] == f) continue;  // not the start of a run
    unsigned long long h = 0, a = 0, min_a = ~0ull, chk_h = 0, chk_a = 0;
    uint64_t e = i, at_h = 0, at_a = 0;
    for (; e < n && fp[e] == f; ++e) {
      const unsigned long long v = index_at(e);
      if (v & kFpFile2) {
        ++a;
        at_a = e;
        chk_a = v & kFpCheckMask;
        min_a = (v & kFpIndexMask) < min_a ? (v & kFpIndexMask) : min_a;
      } else {
        ++h;
        at_h = e;
        chk_h = v & kFpCheckMask;
      }
    }
    bool pair = h == 1 && a == 1 && chk_h == chk_a;
    bool by_index = false;  // the names decide, and these records cannot: whole names, fetched by record index
    if (pair && names) {
      const unsigned long long* nh = names + val[at_h] * kFpNameWords;
      const unsigned long long* na = names + val[at_a] * kFpNameWords;
      bool same = true;
#pragma unroll
      for (uint32_t k = 0; k < kFpNameWords; ++k) same = same && nh[k] == na[k];
      if (nh[kFpNameWords - 1] > kNameInline || na[kFpNameWords - 1] > kNameInline) by_index = true;
      else if (!same) by_index = true;  // (a collision of 87 hash bits: two different names - one left over, one unpaired)
      pair = same && !by_index;
    }
    if (pair) ++matched;
    else if (by_index) {
      const unsigned long long at = atomicAdd(&sum->n_complex, (unsigned long long)(e - i));
      for (uint64_t k = i; k < e; ++k)
        if (at + (k - i) < cap) {
          entries[2 * (at + (k - i))] = i;  // the run
          entries[2 * (at + (k - i)) + 1] = index_at(k) & ~kFpCheckMask;
        }
    }
    else if (a == 0) leftover += h;
    else if (h == 0) {
      unpaired += a;
      first = min_a < first ? min_a : first;
    } else {
      const unsigned long long at = atomicAdd(&sum->n_complex, (unsigned long long)(e - i));
      for (uint64_t k = i; k < e; ++k)
        if (at + (k - i) < cap) {
          entries[2 * (at + (k - i))] = i;  // the run
          entries[2 * (at + (k - i)) + 1] = index_at(k) & ~kFpCheckMask;
        }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    matched += __shfl_xor(matched, d, 64);
    leftover += __shfl_xor(leftover, d, 64);
    unpaired += __shfl_xor(unpaired, d, 64);
    const unsigned long long o = __shfl_xor(first, d, 64);
    first = o < first ? o : first;
  }
  if ((threadIdx.x & 63) == 0) {
    s_acc[threadIdx.x >> 6][0] = matched;
    s_acc[threadIdx.x >> 6][1] = leftover;
    s_acc[threadIdx.x >> 6][2] = unpaired;
    s_acc[threadIdx.x >> 6][3] = first;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long m = 0, l = 0, u = 0, fst = ~0ull;
    for (int w = 0; w < kBlock / kWave; ++w) {
      m += s_acc[w][0];
      l += s_acc[w][1];
      u += s_acc[w][2];
      fst = s_acc[w][3] < fst ? s_acc[w][3] : fst;
    }
    if (m) atomicAdd(&sum->matched, m);
    if (l) atomicAdd(&sum->leftover, l);
    if (u) atomicAdd(&sum->unpaired, u);
    if (fst != ~0ull) atomicMin(&sum->first_unpaired, fst);
  }
}

__global__ __launch_bounds__(kBlock) void k_fp_iota(unsigned long long* __restrict__ v, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) v[i] = i;
}

// split (fp, idx) records into key / value arrays for the sort
__global__ __launch_bounds__(kBlock) void k_fp_split(const FpRec* __restrict__ in, uint64_t n,
                                                     unsigned long long* __restrict__ fp,
                                                     unsigned long long* __restrict__ idx) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const FpRec r = in[i];
  fp[i] = r.fp;
  idx[i] = r.idx;
}

}  // namespace fqg
