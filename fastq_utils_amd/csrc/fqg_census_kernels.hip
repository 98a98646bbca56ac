// fqg_census_kernels.hip - FASTQ -> (cell, UMI) without the BAM round trip.
//
// In the reference's pipeline (sh/fastq2bam:116-273) the barcodes fastq_pre_barcodes cuts out of a read travel inside
// its NAME - add_tags2readname, src/fastq_pre_barcodes.c:192-216: STAGS_CELL=.._UMI=.._SAMPLE=.._ETAGS_ - through the
// aligner into a BAM file, where bam_add_tags parses them out again (get_barcodes, src/bam_add_tags.c:43-99) and writes
// them as CR / RX tags, which bam_umi_count reads back (get_tag, src/bam_umi_count.c:513-522) and packs with
// char2uint_64 (:364-382).  What that chain says about cells and UMIs does not depend on the alignment: it is decided
// when fastq_pre_barcodes keeps a read.  k_bc_census takes the iterations fqg_barcodes_transform kept (its status
// bytes), cuts the same characters get_barcode cut (bc_get), and packs them as char2uint_64 packs the tag values - a
// (cell, UMI) pair of two 64-bit words per kept read that has a UMI (bam_umi_count skips an alignment without one,
// :960; a read without a cell barcode has the cell 0, as char2uint_64 of a missing tag gives).  The pairs are sorted by
// (cell, UMI) (rocPRIM radix sort, fqg_abi.hip) and k_census_* turn the sorted pairs into one line per cell: reads and
// distinct UMIs - bam_umi_count's per-cell totals before a gene tag exists.
#include "fqg_device.h"

namespace fqg {

struct CensusCell {
  unsigned long long cell, reads, umis;
};

// char2uint_64 (src/bam_umi_count.c:364-382) on the n characters at s: from the last character backwards, base 10
// digits A C G T N = 1..5 (either case), stopping at the first character that is none of them
__device__ __forceinline__ unsigned long long census_pack(const uint8_t* __restrict__ s, uint32_t n) {
  unsigned long long v = 0;
  for (uint32_t i = n; i-- > 0;) {
    const uint32_t u = (uint32_t)s[i] & 0xDFu;
    const uint32_t base = u == 'A' ? 1u : u == 'C' ? 2u : u == 'G' ? 3u : u == 'T' ? 4u : u == 'N' ? 5u : 0u;
    if (!base) break;
    v = v * 10ull + base;
  }
  return v;
}
// The tag value is a C string inside the read name: it ends at a NUL (images with NUL bytes).  get_barcodes
// (src/bam_add_tags.c:60-70) ends a value at the first '_' and then expects the next key: a cell or UMI value that holds
// a '_' itself (no barcode of bases does) makes it give up - the read gets no tags at all.  Returns the length, or ~0u
// for such a value.
__device__ __forceinline__ uint32_t census_value_len(const uint8_t* __restrict__ s, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    if (s[i] == 0) return i;
    if (s[i] == '_') return ~0u;
  }
  return n;
}

// The same two functions on a value of at most 32 characters whose bytes lie in four registers (w[j] = bytes 8j .. 8j+7;
// bytes at and beyond n are ignored): the barcodes of every protocol the reference's scripts know are shorter than that.
// A lane that walks its value byte by byte waits for one load per character, and the loads of its 63 neighbours go to 63
// other cache lines; four 8-byte loads, all in flight at once, bring the same bytes.
__device__ __forceinline__ uint32_t census_value_len_w(const uint64_t (&w)[4], uint32_t n) {
  uint32_t len = n;
  bool bad = false;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    const uint64_t z = (w[j] - 0x0101010101010101ull) & ~w[j] & 0x8080808080808080ull;  // lowest set bit: the first NUL
    const uint64_t y = w[j] ^ 0x5F5F5F5F5F5F5F5Full;                                     // '_'
    const uint64_t u = (y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull;
    const uint64_t e = z | u;
    if (e) {
      const uint32_t at = 8u * (uint32_t)j + ((uint32_t)__builtin_ctzll(e) >> 3);
      if (at < n) {  // (words are walked from the last to the first: the first stop of the value wins)
        len = at;
        bad = ((u >> (__builtin_ctzll(e) & 63)) & 1ull) != 0;  // (the lowest set bit of a zero-byte mask is exact)
      }
    }
  }
  return bad ? ~0u : len;
}
__device__ __forceinline__ unsigned long long census_pack_w(const uint64_t (&w)[4], uint32_t n) {
  unsigned long long v = 0;
  bool go = true;
#pragma unroll
  for (int i = 31; i >= 0; --i) {
    const uint32_t u = (uint32_t)(w[i >> 3] >> (8 * (i & 7))) & 0xDFu;
    const uint32_t base = u == 'A' ? 1u : u == 'C' ? 2u : u == 'G' ? 3u : u == 'T' ? 4u : u == 'N' ? 5u : 0u;
    const bool in = (uint32_t)i < n;
    go = go && (!in || base != 0u);
    if (in && go) v = v * 10ull + base;
  }
  return v;
}
// the value's bytes: four loads when they may all be made (the last word may reach 7 bytes beyond the value - they have
// to lie inside the image), else byte by byte
__device__ __forceinline__ bool census_fetch(const uint8_t* __restrict__ s, uint32_t n, const uint8_t* __restrict__ img_end,
                                             uint64_t (&w)[4]) {
  if (n > 32u || s + ((n + 7u) & ~7u) > img_end) return false;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = 8u * (uint32_t)j < n ? ld8(s + 8 * j) : 0ull;
  return true;
}
// one tag value -> its length as get_barcodes sees it (~0u: a '_' inside) and, when it has one, its packed form
__device__ __forceinline__ uint32_t census_value(const uint8_t* __restrict__ s, uint32_t n, const uint8_t* __restrict__ img_end,
                                                 unsigned long long* v) {
  uint64_t w[4];
  if (census_fetch(s, n, img_end, w)) {
    n = census_value_len_w(w, n);
    if (n && n != ~0u) *v = census_pack_w(w, n);
    return n;
  }
  n = census_value_len(s, n);
  if (n && n != ~0u) *v = census_pack(s, n);
  return n;
}

// kCensusPer iterations per lane and ONE reservation of output slots per workgroup and round (a counter that every
// wavefront adds to once per 64 reads is the whole kernel's clock: ~90 adds per microsecond to one address).  The order of
// the pairs in the arrays is of no consequence - fqg_census_finish sorts them.
constexpr int kCensusPer = 4;
__global__ __launch_bounds__(kBlock) void k_bc_census(BcParams P, uint64_t n_done, const uint8_t* __restrict__ status,
                                                      unsigned long long* __restrict__ cells, unsigned long long* __restrict__ umis,
                                                      unsigned long long base, unsigned long long cap,
                                                      unsigned long long* __restrict__ count) {
  __shared__ uint32_t s_wave[kBlock / 64];
  __shared__ unsigned long long s_base;
  const uint64_t tile = (uint64_t)kBlock * kCensusPer;
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n_done; t0 += (uint64_t)gridDim.x * tile) {
    unsigned long long cell[kCensusPer], umi[kCensusPer];
    uint32_t at[kCensusPer];  // my slot among the wavefront's pairs of this round; ~0u: no pair
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kCensusPer; ++j) {
      const uint64_t k = t0 + (uint64_t)j * kBlock + threadIdx.x;
      bool have = k < n_done && status[k] == kBcKeep;
      cell[j] = umi[j] = 0;
      if (have) {
        uint32_t n = 0, cn = 0, qn = 0;
        const uint8_t *s = nullptr, *cs = nullptr, *q = nullptr;
        BcLine ln[4];
        if (P.umi_read > 0) {
          const BcFile& f = P.f[P.umi_read];
          bc_lines(f, k, ln);
          if (bc_get<false>(ln, (long)P.umi_off, (long)P.umi_size, P.phred, 0, &n, &qn, &s, &q) != 0) n = 0;
          if (n) n = census_value(s, n, f.fv.img + f.fv.nbytes, &umi[j]);
        }
        if (P.cell_read > 0) {
          const BcFile& f = P.f[P.cell_read];
          if (P.cell_read != P.umi_read) bc_lines(f, k, ln);  // (10x: both from the index read - its lines once)
          if (bc_get<false>(ln, (long)P.cell_off, (long)P.cell_size, P.phred, 0, &cn, &qn, &cs, &q) != 0) cn = 0;
          if (cn) cn = census_value(cs, cn, f.fv.img + f.fv.nbytes, &cell[j]);
        }
        // no UMI tag: bam_umi_count does not count the alignment (src/bam_umi_count.c:960); a value with a '_': no tags
        if (!n || n == ~0u || cn == ~0u) have = false;
      }
      const unsigned long long m = __ballot(have);
      at[j] = have ? mine + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull)) : ~0u;
      mine += (uint32_t)__builtin_popcountll(m);
    }
    if (lane == 0) s_wave[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t sum = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) {
        const uint32_t v = s_wave[w];
        s_wave[w] = sum;
        sum += v;
      }
      s_base = sum ? atomicAdd(count, (unsigned long long)sum) : 0ull;
    }
    __syncthreads();
    const unsigned long long first = base + s_base + s_wave[wave];
#pragma unroll
    for (int j = 0; j < kCensusPer; ++j) {
      const unsigned long long i = first + at[j];
      if (at[j] != ~0u && i < cap) {
        cells[i] = cell[j];
        umis[i] = umi[j];
      }
    }
    __syncthreads();  // (s_wave and s_base are written again in the next round)
  }
}

// sorted by (cell, UMI): flag[i] = 1 where a new cell begins
__global__ __launch_bounds__(kBlock) void k_census_flags(const unsigned long long* __restrict__ cells, uint64_t n,
                                                         uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || cells[i] != cells[i - 1]) ? 1u : 0u;
}

// One line per cell.  cell_of(i) = (new-cell flags in front of i) + flag[i] - 1; inside a wavefront the pairs of one cell
// are neighbours, so every run of a cell adds once, by its first lane.
__global__ __launch_bounds__(kBlock) void k_census_count(const unsigned long long* __restrict__ cells,
                                                         const unsigned long long* __restrict__ umis, uint64_t n,
                                                         const uint32_t* __restrict__ flag, const unsigned long long* __restrict__ local,
                                                         const unsigned long long* __restrict__ span_excl,
                                                         CensusCell* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = lane_id();
  const bool in = i < n;
  const unsigned long long cell = in ? cells[i] : 0ull, umi = in ? umis[i] : 0ull;
  const uint32_t fl = in ? flag[i] : 0u;
  const unsigned long long cid = in ? span_excl[i / kScan64Span] + local[i] + fl - 1ull : 0ull;
  const bool new_pair = in && (fl || umis[i - 1] != umi);  // (fl == 0: i > 0 and the same cell in front)
  const bool head = in && (fl || lane == 0);
  const unsigned long long hm = __ballot(head), pm = __ballot(new_pair), im = __ballot(in);
  if (head) {
    // my run: from this lane to the lane in front of the next head (or the last lane that holds a pair)
    const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1)) << (lane + 1);
    const int end = above ? __builtin_ctzll(above) : 64;  // first lane beyond the run
    const unsigned long long run = (end == 64 ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull) & im;
    if (fl) out[cid].cell = cell;
    atomicAdd(&out[cid].reads, (unsigned long long)__builtin_popcountll(run));
    const unsigned long long u = (unsigned long long)__builtin_popcountll(run & pm);
    if (u) atomicAdd(&out[cid].umis, u);
  }
}

}  // namespace fqg
