// fqg_umi_kernels.hip - the alignment loop of bam_umi_count (reference src/bam_umi_count.c:942-1060)
// and its per-cell output decisions (cell2MM :666-705, write2MM :584-663) on gfx950.
//
// Input: the inflated alignment records of a BAM file (the host inflates BGZF and walks the
// block_size chain to get one offset per record).  One pass of kernels over N records:
//
//   k_umi_parse     1 thread / record: BAM core fields, aux scan with libbam's rules (bam_aux_get,
//                   bam_aux2Z, bam_aux2i), filters (:950-970), barcode packing (char2uint_64
//                   :364-382), first GX token + n_feat with strtok's rules (:1031-1058)
//   k_umi_insert    whitelists (:981-999), then every key goes into an open-addressing table that
//                   keeps the SMALLEST record index per key (atomicMin): UMIs, cells (packed u64)
//                   and feature names (32-bit tag + byte compare against the claimant's bytes)
//   k_umi_flag + scans   dense ids in order of first appearance (label_str2id :143, blabel2id :225)
//                   = 1 + number of keys whose first record comes earlier: a flag per first record
//                   and an exclusive prefix sum over the record axis
//   k_umi_assign    ids per record, the "sorted by cell" test (:1002-1008), the limits of
//                   process_entry (:447-462); the first finding in record order wins
//   k_umi_count     (cell, feature, UMI) triples into a hash SET with the smallest record index: the
//                   record that holds it is the one process_entry sees as a new UMI (:495-502);
//                   (cell, feature) pairs get a slot with read / new-UMI counters
//   k_umi_sums_*    the float32 counters of the reference.  When every increment is 1.0 (no NH > 1,
//                   one GX per read) they are exact integers = the atomic counters; otherwise the
//                   records are sorted by pair / cell (stable radix sort, rocPRIM) and summed
//                   sequentially in record order, which reproduces the float32 rounding
//   k_umi_rank / k_umi_emit   pairs grouped by cell; one workgroup per cell ranks its features
//                   through an LDS bitmap (no sort), applies the early-break rule of cell2MM
//                   (:697) / write2MM (:643), the thresholds and the UMI / reads fallback, and
//                   writes (row, column, value) triples in file order
//
// The UMI container is a set, as src/range_list.h:150-162 documents it; the reference's RL_Tree
// implementation loses and invents members when ids arrive out of order.
#include "fqg_device.h"

namespace fqg {

constexpr unsigned long long kKeyEmpty = ~0ull;
constexpr uint32_t kNoIdx = 0xFFFFFFFFu;
constexpr uint32_t kUmisFeature = 1048576u;  // src/bam_umi_count.c:48
constexpr int kFeatIdMaxLen = 25;            // :40

// stages a record reaches (what the counters of main() need)
constexpr uint8_t kStSkipped = 0;    // filtered, or no feature tag
constexpr uint8_t kStNoUmi = 1;      // feature tag found (num_tags_found), no UMI
constexpr uint8_t kStUmi = 2;        // UMI present (intermediate)
constexpr uint8_t kStUmiDiscarded = 3;
constexpr uint8_t kStCellDiscarded = 4;
constexpr uint8_t kStCounted = 5;    // reached the cell id (and process_entry when it has a token)

struct UmiParams {
  uint8_t feat_tag[2], cell_tag[2], umi_tag[2];
  int sorted_by_cell, uniq_mapped_only;
  uint32_t max_cells, max_features, min_reads, min_umis;
  const unsigned long long* known_umis;  // distinct packed values in whitelist order (id = index + 1)
  uint32_t n_known_umis;
  const unsigned long long* known_umis_sorted;  // the same, sorted, with ...
  const uint32_t* known_umis_order;             // ... their whitelist order
  const unsigned long long* known_cells_sorted;
  uint32_t n_known_cells;
  int have_known_umis, have_known_cells;
  // shards of one file: the file's numbering of the UMIs that are not whitelisted (sorted keys -> 1-based ids); null:
  // numbered in order of first appearance in this call
  const unsigned long long* umi_table_keys;
  const uint32_t* umi_table_ids;
  uint32_t n_umi_table;
};

struct UmiRec {            // per record, after parsing
  unsigned long long umi_i, cell_i;
  unsigned long long tok_off;  // first GX token: offset in the record buffer
  uint32_t tok_len;
  float incr;
};

struct UmiCall {
  unsigned long long first_key;  // min (record << 8 | code)
  unsigned long long aux;        // id that broke a limit (for the message)
  unsigned long long n_tags, n_umis_disc, n_cells_disc;
  unsigned long long n_counted, n_new;     // records that reached process_entry / that brought a new UMI
  unsigned long long n_lines[2], tot[2];  // matrix lines / sum of truncated counts: [0] ucounts, [1] rcounts
  // counters that every wavefront adds to: 64 copies each, picked by workgroup, summed on the host
  // (a hundred thousand atomic adds to ONE address cost more than the kernels that issue them)
  unsigned long long spread[3][64];  // [0] n_tags, [1] n_counted, [2] n_new
  unsigned long long tot_spread[2][64];  // the same for tot[] (k_umi_emit: one add per workgroup)
  unsigned int all_unit;         // 1 while every increment seen is exactly 1.0f
  unsigned int table_full;
  unsigned int max_cell_records; // sorted mode: records of the largest cell (k_umi_cell_sizes)
  unsigned int pad_;
  float db_reads, db_umi;
};

__device__ __forceinline__ uint64_t umi_mix(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

// ---- libbam 0.1.19 aux access (bam_aux.c, bam.h:772-778) ----------------------------------------
__device__ __forceinline__ int aux_type2size(int x) {
  if (x == 'C' || x == 'c' || x == 'A') return 1;
  if (x == 'S' || x == 's') return 2;
  if (x == 'I' || x == 'i' || x == 'f' || x == 'F') return 4;
  return 0;
}
__device__ __forceinline__ int c_toupper(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

// The walks below run one record per lane over bytes in LDS: a chain of dependent reads, whose LATENCY is the cost
// (two or three wavefronts per SIMD hide little of it).  Strings are therefore scanned 8 bytes per read, and every
// function is a template over the pointer type, so that the staged walk is compiled to DS instructions (a pointer
// that may be LDS or global memory costs FLAT accesses).
typedef const __attribute__((address_space(3))) uint8_t* LdsBytes;
typedef uint64_t __attribute__((aligned(1), may_alias)) u64_unaligned;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32_unaligned;
__device__ __forceinline__ uint64_t ld8(LdsBytes p) { return *(const __attribute__((address_space(3))) u64_unaligned*)p; }
__device__ __forceinline__ uint32_t ld4(LdsBytes p) { return *(const __attribute__((address_space(3))) u32_unaligned*)p; }
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
// the first NUL in [s, lim), or lim
template <class B>
__device__ __forceinline__ B find_nul(B s, B lim) {
  while (lim - s >= 8) {
    const uint64_t m = bytes_eq(ld8(s), 0);
    if (m) return s + (__builtin_ctzll(m) >> 3);
    s += 8;
  }
  while (s < lim && *s) ++s;
  return s;
}
// index of the first byte equal to c in s[from, n), or n (s + n <= lim)
template <class B>
__device__ __forceinline__ uint32_t find_byte(B s, uint32_t from, uint32_t n, uint8_t c, B lim) {
  uint32_t p = from;
  while (p + 8 <= n) {
    const uint64_t m = bytes_eq(ld8(s + p), c);
    if (m) return p + (uint32_t)(__builtin_ctzll(m) >> 3);
    p += 8;
  }
  if (p < n && lim - (s + p) >= 8) {
    const uint64_t m = bytes_eq(ld8(s + p), c) & ((1ull << (8 * (n - p))) - 1ull);
    return m ? p + (uint32_t)(__builtin_ctzll(m) >> 3) : n;
  }
  while (p < n && s[p] != c) ++p;
  return p;
}

// Four bam_aux_get calls (bam_aux.c: the type byte of the first field named tag, or null) in ONE walk over the aux area
// [s, end): bit k of the result says that tag[k] has a hit, hit[k] = its type byte.  A call for one tag skips every
// field before its hit exactly as the calls for the other tags do, so one walk that remembers first hits is the same.
// `lim` bounds every read (a malformed field may run past `end`, as in libbam); a 'B' field whose count leads outside
// [s, lim) ends the walk.
template <class B>
__device__ __forceinline__ uint32_t aux_get4(B s, B end, B lim, const uint8_t (*tag)[2], B* hit) {
  uint32_t open = 15u, got = 0;  // tags still looked for / found
  while (s < end && open) {
    if (lim - s < 2) break;
    const int x0 = s[0], x1 = s[1];
    s += 2;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((open >> k) & 1u)
        if (x0 == tag[k][0] && x1 == tag[k][1]) {
          open &= ~(1u << k);
          if (s < lim) {
            hit[k] = s;
            got |= 1u << k;
          }
        }
    if (s >= lim) break;
    const int type = c_toupper(*s);
    ++s;
    if (type == 'Z' || type == 'H') {
      s = find_nul(s, lim) + 1;
    } else if (type == 'B') {
      if (lim - s < 5) break;
      const long skip = 5 + (long)aux_type2size(*s) * (long)(int32_t)ld4(s + 1);
      if (skip < 0 || skip > (long)(lim - s)) break;
      s += skip;
    } else {
      s += aux_type2size(type);
    }
  }
  return got;
}

// bam_aux2Z + get_tag (src/bam_umi_count.c:513-522): string bytes and length (0 = EMPTY_STRING)
template <class B>
__device__ __forceinline__ uint32_t aux_string(bool have, B t, B lim, B* str) {
  if (!have) return 0;
  if (*t != 'Z' && *t != 'H') return 0;
  *str = t + 1;
  return (uint32_t)(find_nul(t + 1, lim) - (t + 1));
}

template <class B>
__device__ __forceinline__ int32_t aux_int(B t, B lim) {  // bam_aux2i
  if (lim - t < 2) return 0;
  const int type = *t;
  B s = t + 1;
  if (((type == 's' || type == 'S') && lim - t < 3) || ((type == 'i' || type == 'I') && lim - t < 5)) return 0;
  if (type == 'c') return (int32_t)(int8_t)s[0];
  if (type == 'C') return (int32_t)s[0];
  if (type == 's') return (int32_t)(int16_t)((uint16_t)s[0] | ((uint16_t)s[1] << 8));
  if (type == 'S') return (int32_t)((uint16_t)s[0] | ((uint16_t)s[1] << 8));
  if (type == 'i' || type == 'I') return (int32_t)ld4(s);
  return 0;
}

// char2uint_64 (src/bam_umi_count.c:364-382): base 10, A C G T N -> 1..5, parsed from the end, stops
// at the first other character
__device__ __forceinline__ int base2int(uint32_t c) {  // :321-337
  c |= 0x20u;
  return c == 'a' ? 1 : c == 'c' ? 2 : c == 'g' ? 3 : c == 't' ? 4 : c == 'n' ? 5 : 0;
}
template <class B>
__device__ __forceinline__ unsigned long long pack_barcode(B s, uint32_t n, B lim) {
  uint32_t pos = find_byte(s, 0, n, (uint8_t)'\n', lim);
  unsigned long long v = 0;
  while (pos >= 8) {  // 8 characters per read, the last one first
    const uint64_t w = ld8(s + pos - 8);
#pragma unroll
    for (int k = 7; k >= 0; --k) {
      const int b = base2int((uint32_t)(w >> (8 * k)) & 0xFFu);
      if (!b) return v;
      v = v * 10ull + (unsigned long long)b;
    }
    pos -= 8;
  }
  while (pos > 0) {
    const int b = base2int(s[pos - 1]);
    if (!b) break;
    v = v * 10ull + (unsigned long long)b;
    --pos;
  }
  return v;
}

// The 64 records of a wavefront are contiguous in the stream: they are staged in LDS with coalesced
// 16-byte loads, and every thread then walks ITS record there (a thread-per-record walk over global
// memory touches 64 different cache lines per instruction).  Spans larger than the buffer (long
// records) are walked in global memory.
constexpr int kParseStage = 16 * 1024;  // most bytes per wavefront

// one record: `base` is byte span0 of the stream (its LDS copy or the stream itself); every read stays inside
// [base, lim); offsets written out are stream offsets
template <class B>
__device__ __forceinline__ uint8_t parse_record(B base, B lim, B r, uint64_t span0, const UmiParams& P, UmiRec& out) {
  uint8_t st = kStSkipped;
  do {
    if (lim - r < 36) break;
    const int32_t block_len = (int32_t)ld4(r);
    const int32_t tid = (int32_t)ld4(r + 4);
    const uint32_t l_qname = ld4(r + 12) & 0xFFu;
    const uint32_t flag_nc = ld4(r + 16);
    const uint32_t flag = flag_nc >> 16, n_cigar = flag_nc & 0xFFFFu;
    const uint32_t l_qseq = ld4(r + 20);
    if (tid < 0) break;          // :950
    if (flag & 4u) break;        // BAM_FUNMAP :951
    const uint64_t room = (uint64_t)(lim - r);
    uint64_t end_o = block_len < -4 ? 0ull : (uint64_t)(4l + (long)block_len);  // (negative: nothing to walk)
    if (end_o > room) end_o = room;
    uint64_t aux_o = 36ull + l_qname + 4ull * n_cigar + (l_qseq + 1) / 2 + (uint64_t)l_qseq;
    if (aux_o > end_o) aux_o = end_o;
    const B end = r + end_o, aux = r + aux_o;
    int nh_i = 1;
    const uint8_t tags[4][2] = {{'N', 'H'}, {P.feat_tag[0], P.feat_tag[1]}, {P.umi_tag[0], P.umi_tag[1]},
                                {P.cell_tag[0], P.cell_tag[1]}};
    B hit[4] = {r, r, r, r};
    const uint32_t got = aux_get4(aux, end, lim, tags, hit);
    if (got & 1u) {
      nh_i = aux_int(hit[0], lim);
      if (nh_i > 1 && P.uniq_mapped_only) break;
    }
    B feat = r, umi = r, cell = r;
    const uint32_t lf = aux_string((got & 2u) != 0, hit[1], lim, &feat);
    if (!lf) break;
    st = kStNoUmi;
    const uint32_t lu = aux_string((got & 4u) != 0, hit[2], lim, &umi);
    if (!lu) break;
    st = kStUmi;
    const uint32_t lc = aux_string((got & 8u) != 0, hit[3], lim, &cell);
    out.umi_i = pack_barcode(umi, lu, lim);
    out.cell_i = lc ? pack_barcode(cell, lc, lim) : 0ull;
    // strtok(feat, ","): tokens are the maximal runs of non-comma bytes.  n_feat counts the first
    // token and every token equal to its predecessor; only the first token is processed (the first
    // pass replaced the commas by NULs)
    uint32_t p = 0, n_feat = 0, t0 = 0, l0 = 0, prev_s = 0, prev_l = 0;
    bool have_prev = false;
    while (p < lf) {
      while (p < lf && feat[p] == ',') ++p;
      if (p >= lf) break;
      const uint32_t s0 = p;
      p = find_byte(feat, p, lf, (uint8_t)',', lim);
      const uint32_t len = p - s0;
      bool same = have_prev && len == prev_l;
      if (same)
        for (uint32_t k = 0; k < len; ++k)
          if (feat[s0 + k] != feat[prev_s + k]) {
            same = false;
            break;
          }
      if (!have_prev) {
        t0 = s0;
        l0 = len;
      }
      if (!have_prev || same) ++n_feat;
      have_prev = true;
      prev_s = s0;
      prev_l = len;
    }
    out.tok_off = span0 + (unsigned long long)(feat - base) + t0;
    out.tok_len = l0;
    out.incr = (float)(1.0 / (double)((int)n_feat * nh_i));  // float incr=1.0/(n_feat*nh_i) :1044
  } while (false);
  return st;
}

// One wavefront per tile of T consecutive records (T <= 64, one lane per record): T and the LDS area follow the mean
// record size (host: umi_parse_tile) - 64 records of 124 bytes in 9 KiB, 48 of 275 in 16 KiB - so that the usual
// tile is staged and many wavefronts share a CU; a tile that does not fit anyway is walked in global memory.
__global__ __launch_bounds__(kWave) void k_umi_parse(const uint8_t* __restrict__ gbuf, uint64_t nbytes,
                                                     const unsigned long long* __restrict__ offs, uint32_t n, uint32_t T,
                                                     uint32_t cap, UmiParams P, UmiRec* __restrict__ rec,
                                                     uint8_t* __restrict__ stage, UmiCall* __restrict__ call) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_parse[];
  const int lane = (int)threadIdx.x;
  const uint32_t i0 = blockIdx.x * T;
  if (i0 >= n) return;
  const uint32_t Tn = n - i0 < T ? n - i0 : T;
  const bool valid = (uint32_t)lane < Tn;
  const uint32_t i = i0 + (valid ? (uint32_t)lane : Tn - 1);
  const uint32_t i_last = i0 + Tn - 1;
  const uint64_t my_off = offs[i];
  const uint64_t span0 = __shfl(my_off, 0) & ~15ull;  // aligned down: 16-byte loads
  const uint64_t span1 = i_last + 1 < n ? offs[i_last + 1] : nbytes;
  const bool staged = span1 - span0 <= (uint64_t)cap;
  if (staged) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t units = (uint32_t)((span1 - span0 + 15) >> 4);
    const uint64_t safe_units = (nbytes - span0) >> 4;  // whole units inside the stream
    for (uint32_t u0 = 0; u0 < units; u0 += 4 * kWave) {
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // four loads in flight per lane
        uint32_t u = u0 + j * kWave + (uint32_t)lane;
        u = u < units ? u : units - 1;
        if ((uint64_t)u < safe_units) v[j] = *reinterpret_cast<const u32x4*>(gbuf + span0 + 16ull * u);
        else {
          u32x4 t = {0, 0, 0, 0};
          for (uint64_t b = 0; span0 + 16ull * u + b < nbytes && b < 16; ++b) reinterpret_cast<uint8_t*>(&t)[b] = gbuf[span0 + 16ull * u + b];
          v[j] = t;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t u = u0 + j * kWave + (uint32_t)lane;
        if (u < units) *reinterpret_cast<u32x4*>(s_parse + 16u * u) = v[j];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  UmiRec out;
  out.umi_i = out.cell_i = 0;
  out.tok_off = 0;
  out.tok_len = 0;
  out.incr = 1.0f;
  uint8_t st = kStSkipped;
  if (valid) {
    if (staged) {
      const LdsBytes lb = (LdsBytes)s_parse;
      st = parse_record(lb, lb + (uint32_t)(span1 - span0), lb + (uint32_t)(my_off - span0), span0, P, out);
    } else {
      const uint8_t* gb = gbuf + span0;
      st = parse_record(gb, gbuf + nbytes, gbuf + my_off, span0, P, out);
    }
    rec[i] = out;
    stage[i] = st;
  }
  const unsigned long long tags = __ballot(valid && st >= kStNoUmi);
  if (lane == 0 && tags) atomicAdd(&call->spread[0][blockIdx.x & 63], (unsigned long long)__builtin_popcountll(tags));
}

// ---- hash tables --------------------------------------------------------------------------------
struct KeyTable {          // u64 key -> smallest record index
  unsigned long long* keys;
  uint32_t* first;
  uint64_t mask;
};
struct NameTable {         // feature names: slot = tag32 << 32 | claimant record
  unsigned long long* slots;
  unsigned long long* words;  // 3 per slot: the claimant's name (k_umi_insert)
  uint32_t* first;
  uint64_t mask;
};

// The two tables of the counting step are touched at random by every record: key and smallest record
// index share one 16-byte slot there, so that an insert costs one cache line, not two.
struct KeySlot {
  unsigned long long key;
  uint32_t first;
  uint32_t pad;
};
struct SlotTable {
  KeySlot* s;  // all bytes 0xFF: empty
  uint64_t mask;
};
__device__ __forceinline__ uint32_t slot_insert(const SlotTable& T, unsigned long long key, uint32_t idx, UmiCall* call) {
  uint64_t h = umi_mix(key) & T.mask;
  for (uint64_t probes = 0; probes <= T.mask; ++probes) {
    unsigned long long k = T.s[h].key;
    if (k == kKeyEmpty) {
      k = atomicCAS(&T.s[h].key, kKeyEmpty, key);
      if (k == kKeyEmpty) k = key;
    }
    if (k == key) {
      if (T.s[h].first > idx) atomicMin(&T.s[h].first, idx);  // look first: see table_insert
      return (uint32_t)h;
    }
    h = (h + 1) & T.mask;
  }
  atomicOr(&call->table_full, 1u);
  return kNoIdx;
}

__device__ __forceinline__ uint32_t table_insert(const KeyTable& T, unsigned long long key, uint32_t idx,
                                                 UmiCall* call) {
  uint64_t h = umi_mix(key) & T.mask;
  for (uint64_t probes = 0; probes <= T.mask; ++probes) {
    unsigned long long k = T.keys[h];
    if (k == kKeyEmpty) {
      k = atomicCAS(&T.keys[h], kKeyEmpty, key);
      if (k == kKeyEmpty) k = key;
    }
    if (k == key) {
      // hot keys (a cell's run of records, a highly expressed gene) are hit by thousands of threads:
      // look first - the value only ever decreases, so a stale larger one costs an atomic, never a miss
      if (T.first[h] > idx) atomicMin(&T.first[h], idx);
      return (uint32_t)h;
    }
    h = (h + 1) & T.mask;
  }
  atomicOr(&call->table_full, 1u);
  return kNoIdx;
}

__device__ __forceinline__ uint32_t table_find(const KeyTable& T, unsigned long long key) {
  uint64_t h = umi_mix(key) & T.mask;
  for (uint64_t probes = 0; probes <= T.mask; ++probes) {
    const unsigned long long k = T.keys[h];
    if (k == key) return (uint32_t)h;
    if (k == kKeyEmpty) return kNoIdx;
    h = (h + 1) & T.mask;
  }
  return kNoIdx;
}

__device__ __forceinline__ bool sorted_contains(const unsigned long long* a, uint32_t n, unsigned long long v,
                                                uint32_t* at) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  if (at) *at = lo;
  return lo < n && a[lo] == v;
}

// whitelists, then the three key tables.  slot arrays: table slot of the record's key, or
// kNoIdx | (whitelist order) for UMIs that the whitelist already numbered.
constexpr uint32_t kWhiteBit = 0x80000000u;

// a feature name (at most kFeatIdMaxLen - 2 = 23 bytes) as three words, bytes beyond its length zero: hashing and
// comparing names byte by byte is a chain of dependent memory round trips per name, this is one
__device__ __forceinline__ void name_words(const uint8_t* buf, uint64_t nbytes, uint64_t off, uint32_t len, uint64_t w[3]) {
  const uint8_t* s = buf + off;
  if (off + 24 <= nbytes) {
    w[0] = ld8(s);
    w[1] = ld8(s + 8);
    w[2] = ld8(s + 16);
  } else {
    w[0] = w[1] = w[2] = 0;
    for (uint32_t k = 0; k < len && k < 24; ++k) {
      const uint64_t b = (uint64_t)s[k] << (8 * (k & 7));
      if (k < 8) w[0] |= b;
      else if (k < 16) w[1] |= b;
      else w[2] |= b;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const uint32_t have = len > 8u * j ? len - 8u * j : 0u;
    if (have < 8) w[j] &= have ? (1ull << (8 * have)) - 1ull : 0ull;
  }
}

__global__ __launch_bounds__(kBlock) void k_umi_insert(const uint8_t* __restrict__ buf, uint64_t nbytes, uint32_t n, UmiParams P,
                                                       const UmiRec* __restrict__ rec, uint8_t* __restrict__ stage,
                                                       KeyTable U, KeyTable C, NameTable F,
                                                       uint32_t* __restrict__ uslot, uint32_t* __restrict__ cslot,
                                                       uint32_t* __restrict__ fslot, UmiCall* __restrict__ call, int ablate) {
  // ablate (measurement only, FQGPU_UMI_INSERT_ABL): 1 = no UMI table, 2 = no cell table, 4 = no feature table
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  uint8_t st = i < n ? stage[i] : kStSkipped;
  uint32_t us = kNoIdx, cs = kNoIdx, fs = kNoIdx;
  UmiRec r;
  r.umi_i = r.cell_i = r.tok_off = 0;
  r.tok_len = 0;
  r.incr = 1.0f;
  bool live = false, white = false, cell_ok = false, do_f = false;
  uint64_t w[3] = {0, 0, 0};
  if (st == kStUmi) {
    r = rec[i];
    // valid_barcode() looks the PACKED umi up among the whitelist's dense IDS (src/bam_umi_count.c:559-571,
    // 984): true iff 1 <= packed <= number of distinct whitelist entries
    if (P.have_known_umis && !(r.umi_i >= 1 && r.umi_i <= (unsigned long long)P.n_known_umis)) st = kStUmiDiscarded;
    else {
      live = true;
      uint32_t at;
      white = (ablate & 1) || (P.have_known_umis && sorted_contains(P.known_umis_sorted, P.n_known_umis, r.umi_i, &at));
      if (white && !(ablate & 1)) us = kWhiteBit | P.known_umis_order[at];
      cell_ok = !(P.have_known_cells && !sorted_contains(P.known_cells_sorted, P.n_known_cells, r.cell_i, nullptr));
      do_f = cell_ok && r.tok_len > 0 && r.tok_len + 1 < (uint32_t)kFeatIdMaxLen && !(ablate & 4);
      if (do_f) name_words(buf, nbytes, r.tok_off, r.tok_len, w);
    }
  }
  // ---- cells.  Consecutive records mostly share their cell (the default mode REQUIRES the file to be grouped by cell):
  // hundreds of threads asking the table for one key at once are hundreds of atomics on one address.  Inside a
  // wavefront only the first lane of every run of equal cells asks (with its record index, the smallest of the run),
  // the others take its answer.
  {
    const bool want_c = live && cell_ok && !(ablate & 2);
    const unsigned long long ckey = r.cell_i;
    const bool prev_want = __shfl_up((int)want_c, 1, 64) != 0;
    const unsigned long long prev_key = __shfl_up(ckey, 1, 64);
    const bool leader = want_c && (lane == 0 || !prev_want || prev_key != ckey);
    uint32_t cs_lead = kNoIdx;
    if (leader) {
      const uint64_t hc = umi_mix(ckey) & C.mask;
      const unsigned long long kc = C.keys[hc];
      const uint32_t fc = C.first[hc];
      cs_lead = (kc == ckey && fc <= i) ? (uint32_t)hc : table_insert(C, ckey, i, call);
    }
    const unsigned long long lm = __ballot(leader);
    const unsigned long long at_or_below = lm & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int src = at_or_below ? 63 - __builtin_clzll(at_or_below) : 0;
    const uint32_t cs_run = __shfl(cs_lead, src, 64);
    if (want_c) cs = cs_run;
  }
  if (live) {
    w[2] |= (uint64_t)r.tok_len << 56;  // (a name has at most 23 bytes)
    const uint64_t hsh = umi_mix(umi_mix(umi_mix(0x9E3779B97F4A7C15ull ^ w[0]) ^ w[1]) ^ w[2]);
    const unsigned long long tag = (hsh >> 32) << 32;
    // Most records repeat keys that are in the tables already (an expressed gene, a UMI seen before) with an earlier
    // first record: nothing to write.  That case is decided from the keys' HOME slots, read for both tables at once -
    // one round trip to memory instead of dependent chains.
    const uint64_t hu = umi_mix(r.umi_i) & U.mask, hf = hsh & F.mask;
    unsigned long long ku = 0, vf = 0, q0 = 0, q1 = 0, q2 = 0;
    uint32_t fu = 0, ff = 0;
    if (!white) {
      ku = U.keys[hu];
      fu = U.first[hu];
    }
    if (do_f) {
      vf = F.slots[hf];
      ff = F.first[hf];
      q0 = F.words[3 * hf];
      q1 = F.words[3 * hf + 1];
      q2 = F.words[3 * hf + 2];
    }
    if (!white) us = (ku == r.umi_i && fu <= i) ? (uint32_t)hu : table_insert(U, r.umi_i, i, call);
    if (!cell_ok) st = kStCellDiscarded;
    else {
      st = kStCounted;
      if (do_f) {
        // F.words[] holds the claimant's name, written after its claim without any ordering: three words that all
        // equal mine can only be the finished name (no word of a name reads ~0: the third carries the length, and
        // names whose first 16 bytes hold eight 0xFF in a row take the slow path)
        if ((vf >> 32) == (tag >> 32) && q0 == w[0] && q1 == w[1] && q2 == w[2] && w[0] != ~0ull && w[1] != ~0ull && ff <= i)
          fs = (uint32_t)hf;
        else {
          uint64_t h = hf;
          for (uint64_t probes = 0; probes <= F.mask; ++probes) {
            unsigned long long v = F.slots[h];
            if (v == kKeyEmpty) {
              v = atomicCAS(&F.slots[h], kKeyEmpty, tag | i);
              if (v == kKeyEmpty) {
                v = tag | i;
                F.words[3 * h] = w[0];
                F.words[3 * h + 1] = w[1];
                F.words[3 * h + 2] = w[2];
              }
            }
            if ((v >> 32) == (tag >> 32)) {
              bool same = (uint32_t)v == i;
              if (!same) {
                const UmiRec o = rec[(uint32_t)v];
                if (o.tok_len == r.tok_len) {
                  uint64_t q[3];
                  name_words(buf, nbytes, o.tok_off, o.tok_len, q);
                  same = q[0] == w[0] && q[1] == w[1] && (q[2] | ((uint64_t)o.tok_len << 56)) == w[2];
                }
              }
              if (same) {
                if (F.first[h] > i) atomicMin(&F.first[h], i);
                fs = (uint32_t)h;
                break;
              }
            }
            h = (h + 1) & F.mask;
          }
          if (fs == kNoIdx) atomicOr(&call->table_full, 1u);
        }
      }
    }
  }
  if (i < n) {
    stage[i] = st;
    uslot[i] = us;
    cslot[i] = cs;
    fslot[i] = fs;
  }
  const unsigned long long du = __ballot(st == kStUmiDiscarded), dc = __ballot(st == kStCellDiscarded);
  if ((threadIdx.x & 63) == 0) {
    if (du) atomicAdd(&call->n_umis_disc, (unsigned long long)__builtin_popcountll(du));
    if (dc) atomicAdd(&call->n_cells_disc, (unsigned long long)__builtin_popcountll(dc));
  }
}

// flag[first record of every key] = 1 (the flag arrays are zeroed by the host)
__global__ __launch_bounds__(kBlock) void k_umi_flag(const uint32_t* __restrict__ first, uint64_t n_slots,
                                                     uint32_t* __restrict__ flag) {
  const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h >= n_slots) return;
  const uint32_t f = first[h];
  if (f != kNoIdx) flag[f] = 1u;
}

__global__ __launch_bounds__(kBlock) void k_umi_flag3(Scan3 t, uint64_t n_slots) {
  const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h >= n_slots) return;
  const uint32_t f = t.first[blockIdx.y][h];
  if (f != kNoIdx) t.flag[blockIdx.y][f] = 1u;
}

// exclusive prefix over u32 flags: local (per span of kUmiSpan) + span sums (scanned by k_scan64_b)
constexpr int kUmiSpan = kBlock * 8;
struct Prefix {
  const unsigned long long* local;
  const unsigned long long* spans;
  __device__ __forceinline__ uint32_t at(uint32_t i) const { return (uint32_t)(local[i] + spans[i / kUmiSpan]); }
};

struct UmiIds {
  uint32_t *umi, *cell, *feat;
};

__global__ __launch_bounds__(kBlock) void k_umi_assign(uint32_t n, UmiParams P, const UmiRec* __restrict__ rec,
                                                       const uint8_t* __restrict__ stage, KeyTable U, KeyTable C,
                                                       NameTable F, const uint32_t* __restrict__ uslot,
                                                       const uint32_t* __restrict__ cslot,
                                                       const uint32_t* __restrict__ fslot, Prefix pu, Prefix pc,
                                                       Prefix pf, const uint32_t* __restrict__ cflag, UmiIds ids,
                                                       UmiCall* __restrict__ call) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint32_t uid = 0, cid = 0, fid = 0;
  const uint8_t st = stage[i];
  if (st == kStCounted) {
    const UmiRec r = rec[i];
    const uint32_t us = uslot[i];
    if (us != kNoIdx) {
      if (us & kWhiteBit) uid = (us & ~kWhiteBit) + 1u;
      else if (!P.umi_table_keys) uid = P.n_known_umis + pu.at(U.first[us]) + 1u;
      else {  // the id the whole file gives this UMI
        uint32_t lo = 0, hi = P.n_umi_table;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (P.umi_table_keys[mid] < r.umi_i) lo = mid + 1;
          else hi = mid;
        }
        if (lo < P.n_umi_table && P.umi_table_keys[lo] == r.umi_i) uid = P.n_known_umis + P.umi_table_ids[lo];
        else atomicOr(&call->table_full, 2u);  // (not in the table: the caller's table does not cover this shard)
      }
    }
    const uint32_t cs = cslot[i];
    if (cs != kNoIdx) cid = pc.at(C.first[cs]) + 1u;
    unsigned long long key = kKeyEmpty;
    auto finding = [&](uint32_t code, unsigned long long) {  // (the host reads the offending id from ids[])
      const unsigned long long k = ((unsigned long long)i << 8) | code;
      if (k < key) {
        key = k;
        atomicMin(&call->first_key, k);
      }
    };
    // :1002-1008 - ids are dense in first-appearance order, so the largest id so far is the number
    // of cells first seen up to here
    if (P.sorted_by_cell && cid != pc.at(i) + cflag[i]) finding(FQG_E_UMI_NOT_SORTED, cid);
    if (r.tok_len > 0 && key == kKeyEmpty) {
      if (r.tok_len + 1 >= (uint32_t)kFeatIdMaxLen) finding(FQG_E_UMI_FEATURE_NAME, r.tok_len);
      else {
        const uint32_t fs = fslot[i];
        if (fs != kNoIdx) fid = pf.at(F.first[fs]) + 1u;
        // process_entry :447-462
        if (uid > kUmisFeature) finding(FQG_E_UMI_TOO_MANY_UMIS, uid);
        else if (!P.sorted_by_cell && cid > P.max_cells && P.max_cells > 1) finding(FQG_E_UMI_TOO_MANY_CELLS, cid);
        else if (fid > P.max_features) finding(FQG_E_UMI_TOO_MANY_FEATURES, fid);
      }
    }
    if (r.incr != 1.0f && fid) atomicAnd(&call->all_unit, 0u);
  }
  ids.umi[i] = uid;
  ids.cell[i] = cid;
  ids.feat[i] = fid;
}

// ---- counting -------------------------------------------------------------------------------------
struct PairTable {
  SlotTable t;           // key = cell << 32 | feature
  uint32_t* reads;       // records of the pair
  uint32_t* umis;        // records that brought a new UMI
};

__device__ __forceinline__ unsigned long long pair_key(uint32_t cell, uint32_t feat) {
  return ((unsigned long long)cell << 32) | feat;
}

// (cell, feature, UMI) set: key = slot of the (cell, feature) pair << 32 | umi id - exact and unbounded
__global__ __launch_bounds__(kBlock) void k_umi_count(uint32_t n, uint32_t limit, UmiIds ids, SlotTable T,
                                                      PairTable Pt, uint32_t* __restrict__ pslot,
                                                      uint32_t* __restrict__ tslot, UmiCall* __restrict__ call) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint32_t ps = kNoIdx, ts = kNoIdx;
  const uint32_t fid = ids.feat[i];
  if (fid && i < limit) {
    ps = slot_insert(Pt.t, pair_key(ids.cell[i], fid), i, call);
    if (ps != kNoIdx) {
      atomicAdd(&Pt.reads[ps], 1u);
      ts = slot_insert(T, ((unsigned long long)ps << 32) | ids.umi[i], i, call);
    }
  }
  pslot[i] = ps;
  tslot[i] = ts;
}

// is_new[i]: record i is the first one of its (cell, feature, UMI); unit-increment counters
__global__ __launch_bounds__(kBlock) void k_umi_new(uint32_t n, const uint32_t* __restrict__ tslot,
                                                    const uint32_t* __restrict__ pslot, SlotTable T, PairTable Pt,
                                                    UmiIds ids, uint8_t* __restrict__ is_new,
                                                    uint32_t* __restrict__ cell_reads, uint32_t* __restrict__ cell_umis,
                                                    UmiCall* __restrict__ call) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t ts = i < n ? tslot[i] : kNoIdx;
  uint8_t nw = 0;
  uint32_t cid = 0;
  if (ts != kNoIdx) {
    nw = T.s[ts].first == i;
    cid = ids.cell[i];
    if (nw) atomicAdd(&Pt.umis[pslot[i]], 1u);
  }
  if (i < n) is_new[i] = nw;
  const unsigned long long cnt = __ballot(ts != kNoIdx), nws = __ballot(nw != 0);
  // the cell counters: neighbouring records mostly share their cell (a file grouped by cell: all 64 lanes one address),
  // so the first lane of every run of equal cells adds the run's counts
  {
    const uint32_t key = ts != kNoIdx ? cid : 0u;  // (cell ids start at 1: 0 = not counted)
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool leader = key != 0u && (lane == 0 || prev != key);
    const unsigned long long lm = __ballot(leader), zm = __ballot(key == 0u);
    if (leader) {
      // my run ends in front of the next leader or the next lane that is not counted
      const unsigned long long stop = (lm | zm) & (lane == 63 ? 0ull : (~0ull << (lane + 1)));
      const int end = stop ? __builtin_ctzll(stop) : 64;
      const unsigned long long run = (end == 64 ? ~0ull : ((1ull << end) - 1ull)) & (~0ull << lane);
      atomicAdd(&cell_reads[cid], (uint32_t)__builtin_popcountll(cnt & run));
      const uint32_t nn = (uint32_t)__builtin_popcountll(nws & run);
      if (nn) atomicAdd(&cell_umis[cid], nn);
    }
  }
  if ((threadIdx.x & 63) == 0) {
    const uint32_t which = (blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) & 63;
    if (cnt) atomicAdd(&call->spread[1][which], (unsigned long long)__builtin_popcountll(cnt));
    if (nws) atomicAdd(&call->spread[2][which], (unsigned long long)__builtin_popcountll(nws));
  }
}

// float32 value of `count` additions of 1.0f starting from 0 (saturates at 2^24)
__device__ __forceinline__ float unit_sum(uint32_t count) { return count > 16777216u ? 16777216.0f : (float)count; }

// General increments: records sorted by group (stable, so record order inside a group); one thread
// walks a group and adds in float32 exactly as process_entry does.
__global__ __launch_bounds__(kBlock) void k_umi_sums_sorted(uint32_t m, const uint32_t* __restrict__ group,
                                                            const uint32_t* __restrict__ order,
                                                            const UmiRec* __restrict__ rec,
                                                            const uint8_t* __restrict__ is_new,
                                                            float* __restrict__ sum_reads, float* __restrict__ sum_umis) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= m) return;
  const uint32_t g = group[j];
  if (g == kNoIdx || (j > 0 && group[j - 1] == g)) return;  // excluded, or not the first record of its group
  float r = 0.0f, u = 0.0f;
  for (uint32_t k = j; k < m && group[k] == g; ++k) {
    const uint32_t i = order[k];
    const float incr = rec[i].incr;
    if (is_new[i]) u += incr;
    r += incr;
  }
  sum_reads[g] = r;
  sum_umis[g] = u;
}

// unit increments: counters -> the float32 values the reference would hold
__global__ __launch_bounds__(kBlock) void k_umi_unit_sums(uint64_t n, const uint32_t* __restrict__ reads,
                                                          const uint32_t* __restrict__ umis,
                                                          float* __restrict__ f_reads, float* __restrict__ f_umis) {
  const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  f_reads[j] = unit_sum(reads[j]);
  f_umis[j] = unit_sum(umis[j]);
}

// sort keys of the general path: the record's group (pair slot / cell id), kNoIdx when not counted
__global__ __launch_bounds__(kBlock) void k_umi_sort_keys(uint32_t n, const uint32_t* __restrict__ tslot,
                                                          const uint32_t* __restrict__ pslot,
                                                          const uint32_t* __restrict__ cell,
                                                          uint32_t* __restrict__ key_pair, uint32_t* __restrict__ key_cell,
                                                          uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const bool counted = tslot[i] != kNoIdx;
  key_pair[i] = counted ? pslot[i] : kNoIdx;
  key_cell[i] = counted ? cell[i] : kNoIdx;
  idx[i] = i;
}

// db->tot_reads_obs / tot_umi_obs: one float32 chain over all records in order
// (a shard of a file continues the chain of the shards before it: start values, and `skip` records in front that are not its own)
__global__ void k_umi_db_totals(uint32_t n, uint32_t limit, const UmiRec* __restrict__ rec,
                                const uint32_t* __restrict__ tslot, const uint8_t* __restrict__ is_new,
                                UmiCall* __restrict__ call, float start_reads, float start_umi, uint32_t skip) {
  if (blockIdx.x || threadIdx.x) return;
  float r = start_reads, u = start_umi;
  for (uint32_t i = skip; i < n && i < limit; ++i) {
    if (tslot[i] == kNoIdx) continue;
    const float incr = rec[i].incr;
    if (is_new[i]) u += incr;
    r += incr;
  }
  call->db_reads = r;
  call->db_umi = u;
}

// features that have a replayed (cell, feature) set (hash path): pair key = cell << 32 | feature
__global__ __launch_bounds__(kBlock) void k_umi_mark_replayed(uint32_t n_flagged, const uint32_t* __restrict__ flagged,
                                                              const uint32_t* __restrict__ run_pslot, PairTable Pt,
                                                              uint8_t* __restrict__ feat_flag) {
  const uint32_t fi = blockIdx.x * kBlock + threadIdx.x;
  if (fi < n_flagged) feat_flag[(uint32_t)Pt.t.s[run_pslot[flagged[fi]]].key] = 1;
}

// ---- output ---------------------------------------------------------------------------------------
// pairs grouped by cell: cell_count[c] pairs, cell_start = exclusive prefix; pair_of[] filled by atomics
__global__ __launch_bounds__(kBlock) void k_umi_pairs_count(uint64_t n_slots, PairTable Pt,
                                                            uint32_t* __restrict__ cell_pairs) {
  const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h >= n_slots) return;
  const unsigned long long k = Pt.t.s[h].key;
  if (k != kKeyEmpty) atomicAdd(&cell_pairs[(uint32_t)(k >> 32)], 1u);
}
__global__ __launch_bounds__(kBlock) void k_umi_pairs_fill(uint64_t n_slots, PairTable Pt, Prefix start,
                                                           uint32_t* __restrict__ cursor,
                                                           uint32_t* __restrict__ pair_of) {
  const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h >= n_slots) return;
  const unsigned long long k = Pt.t.s[h].key;
  if (k == kKeyEmpty) return;
  const uint32_t c = (uint32_t)(k >> 32);
  pair_of[start.at(c) + atomicAdd(&cursor[c], 1u)] = (uint32_t)h;
}

struct UmiEntry {
  uint32_t row, col, val;
};

// One workgroup per cell: rank the cell's features with a bitmap, apply the output rules and write
// every printed line at (first pair of the cell + rank of the feature) into a scratch that has one
// entry per pair and was filled with 0xFF: file order with gaps, closed by k_umi_compact.
constexpr int kUmiBitmapWords = 4096;  // 131072 feature ids per sweep; more features loop over sweeps

struct EmitArgs {
  PairTable Pt;
  Prefix start;                 // first pair of a cell in pair_of[]
  const uint32_t* cell_pairs;
  const uint32_t* pair_of;
  const float* pair_reads;      // float32 counters per pair slot
  const float* pair_umis;
  const float* cell_umis;       // cells[c].tot_umi_obs
  UmiParams P;
  uint32_t n_cells;
  const uint32_t* remap;        // feature id -> id used for the output rules and printed (null: identity);
                                // sharded runs pass the global first-appearance ids here
  uint32_t cell_offset;         // added to the printed cell id (cells of earlier shards)
  int pairs_by_feature;         // pair_of[] lists a cell's pairs in ascending feature id (the per-cell counting path)
  UmiEntry* out_u;              // scratch, one entry per pair
  UmiEntry* out_r;
};

__global__ __launch_bounds__(kBlock) void k_umi_emit(EmitArgs A, UmiCall* __restrict__ call) {
  __shared__ uint32_t s_bits[kUmiBitmapWords];
  __shared__ uint32_t s_pre[kUmiBitmapWords];
  __shared__ uint32_t s_wave[kBlock / kWave];
  __shared__ uint32_t s_base;
  const uint32_t c = blockIdx.x + 1;  // cell id
  if (c > A.n_cells) return;
  const uint32_t np = A.cell_pairs[c];
  const uint32_t p0 = A.start.at(c);
  const float tot = A.cell_umis[c];
  if (threadIdx.x == 0) s_base = 0;
  // unsorted mode never prints cell ids >= max_cells (write2MM loops while cell_id < max_cells, :612)
  const bool cell_printed = A.P.sorted_by_cell || c + A.cell_offset < A.P.max_cells;
  unsigned long long tu = 0, tr = 0;
  uint32_t np_sweep = 1;
  // sweeps over the feature id space, 131072 ids at a time (one sweep unless --max_feat is huge)
  uint32_t max_f = 0;
  auto feat_of = [&](uint32_t slot) {
    const uint32_t f = (uint32_t)A.Pt.t.s[slot].key;
    return A.remap ? A.remap[f] : f;
  };
  for (uint32_t k = threadIdx.x; k < np; k += kBlock) {
    const uint32_t f = feat_of(A.pair_of[p0 + k]);
    max_f = f > max_f ? f : max_f;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t o = __shfl_xor(max_f, d, 64);
    max_f = o > max_f ? o : max_f;
  }
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = max_f;
  __syncthreads();
  max_f = 0;
  for (int w = 0; w < kBlock / kWave; ++w) max_f = s_wave[w] > max_f ? s_wave[w] : max_f;
  __syncthreads();
  // the output rules for one pair of the cell; `rank` = its place among the cell's features
  auto emit_pair = [&](uint32_t slot, uint32_t feat, uint32_t rank) {
    // the walk over cf stops once `pr >= tot_umi_obs` (:697 / :643).  Sorted mode counts every feature
    // seen so far in the file (ids 1..feat-1 precede this one), unsorted mode the cell's own.
    const uint32_t pr_before = A.P.sorted_by_cell ? feat - 1u : rank;
    if (!cell_printed || feat >= A.P.max_features || (float)pr_before >= tot) return;
    const float u = A.pair_umis[slot], r = A.pair_reads[slot];
    if (!(r >= (float)A.P.min_reads && u >= (float)A.P.min_umis)) return;
    // ucounts: UMIs, or the reads when the UMI count truncates to 0 (:685-695)
    uint32_t val_u = 0, val_r = 0;
    bool pu = false, prd = false;
    if ((uint32_t)u >= 1u) {
      pu = true;
      val_u = (uint32_t)roundf(u);
      tu += (uint32_t)u;
    } else if ((uint32_t)r >= 1u) {
      pu = true;
      val_u = (uint32_t)roundf(r);
      tu += (uint32_t)r;
    }
    if ((uint32_t)r >= 1u) {
      prd = true;
      val_r = (uint32_t)roundf(r);
      tr += (uint32_t)r;
    }
    if (pu) A.out_u[p0 + rank] = UmiEntry{A.P.sorted_by_cell ? feat : 0u, c + A.cell_offset, val_u};
    if (prd) A.out_r[p0 + rank] = UmiEntry{A.P.sorted_by_cell ? feat : 0u, c + A.cell_offset, val_r};
  };
  if (A.pairs_by_feature && !A.remap) {
    // the per-cell counting path hands the pairs over in feature order: the rank is the position, and the bitmap
    // (16 KiB zeroed, counted and prefixed per cell for a few hundred pairs) is not needed
    for (uint32_t k = threadIdx.x; k < np; k += kBlock) {
      const uint32_t slot = A.pair_of[p0 + k];
      emit_pair(slot, (uint32_t)A.Pt.t.s[slot].key, k);
    }
    max_f = 0;  // (no sweep below)
    np_sweep = 0;
  }
  for (uint32_t sweep0 = 0; np_sweep && sweep0 <= max_f; sweep0 += kUmiBitmapWords * 32) {
    for (int w = threadIdx.x; w < kUmiBitmapWords; w += kBlock) s_bits[w] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < np; k += kBlock) {
      const uint32_t f = feat_of(A.pair_of[p0 + k]) - sweep0;
      if (f < (uint32_t)kUmiBitmapWords * 32) atomicOr(&s_bits[f >> 5], 1u << (f & 31));
    }
    __syncthreads();
    // exclusive prefix of the word popcounts: 16 words per thread
    uint32_t local[kUmiBitmapWords / kBlock], sum = 0;
#pragma unroll
    for (int q = 0; q < kUmiBitmapWords / kBlock; ++q) {
      local[q] = sum;
      sum += __popc(s_bits[threadIdx.x * (kUmiBitmapWords / kBlock) + q]);
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < kBlock / kWave; ++w) {
      if (w < (int)(threadIdx.x >> 6)) before += s_wave[w];
      all += s_wave[w];
    }
    const uint32_t base = s_base;  // features of earlier sweeps
#pragma unroll
    for (int q = 0; q < kUmiBitmapWords / kBlock; ++q)
      s_pre[threadIdx.x * (kUmiBitmapWords / kBlock) + q] = base + before + incl - sum + local[q];
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < np; k += kBlock) {
      const uint32_t slot = A.pair_of[p0 + k];
      const uint32_t feat = feat_of(slot);
      const uint32_t f = feat - sweep0;
      if (f >= (uint32_t)kUmiBitmapWords * 32) continue;
      const uint32_t rank = s_pre[f >> 5] + __popc(s_bits[f >> 5] & ((1u << (f & 31)) - 1u));  // among the cell's features
      emit_pair(slot, feat, rank);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = base + all;
    __syncthreads();
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    tu += __shfl_down(tu, d, 64);
    tr += __shfl_down(tr, d, 64);
  }
  // one add per WORKGROUP, to one of 64 copies (40 000 wavefronts adding to two addresses were most of this kernel)
  __shared__ unsigned long long s_tot[2][kBlock / kWave];
  if ((threadIdx.x & 63) == 0) {
    s_tot[0][threadIdx.x >> 6] = tu;
    s_tot[1][threadIdx.x >> 6] = tr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, b = 0;
    for (int w = 0; w < kBlock / kWave; ++w) {
      a += s_tot[0][w];
      b += s_tot[1][w];
    }
    if (a) atomicAdd(&call->tot_spread[0][blockIdx.x & 63], a);
    if (b) atomicAdd(&call->tot_spread[1][blockIdx.x & 63], b);
  }
}

// close the gaps of the rank-indexed scratch: flag -> prefix -> scatter
__global__ __launch_bounds__(kBlock) void k_umi_entry_flags(uint32_t n, const UmiEntry* __restrict__ in,
                                                            uint32_t* __restrict__ flag) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j < n) flag[j] = in[j].col != 0xFFFFFFFFu ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void k_umi_compact(uint32_t n, const UmiEntry* __restrict__ in,
                                                        const uint32_t* __restrict__ flag, Prefix pos,
                                                        UmiEntry* __restrict__ out) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j < n && flag[j]) out[pos.at(j)] = in[j];
}

// names / packed barcodes in id order
__global__ __launch_bounds__(kBlock) void k_umi_export_features(const uint8_t* __restrict__ buf, uint64_t n_slots,
                                                                NameTable F, Prefix pf,
                                                                const UmiRec* __restrict__ rec,
                                                                char* __restrict__ names /* 25 bytes each */) {
  const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h >= n_slots) return;
  const uint32_t f = F.first[h];
  if (f == kNoIdx) return;
  const uint32_t id = pf.at(f);  // 0-based
  const UmiRec r = rec[f];
  char* dst = names + (uint64_t)id * kFeatIdMaxLen;
  for (uint32_t k = 0; k < (uint32_t)kFeatIdMaxLen; ++k) dst[k] = k < r.tok_len ? (char)buf[r.tok_off + k] : 0;
}
__global__ __launch_bounds__(kBlock) void k_umi_export_cells(uint64_t n_slots, KeyTable C, Prefix pc,
                                                             unsigned long long* __restrict__ out) {
  const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h >= n_slots) return;
  const uint32_t f = C.first[h];
  if (f == kNoIdx) return;
  out[pc.at(f)] = C.keys[h];
}

}  // namespace fqg
