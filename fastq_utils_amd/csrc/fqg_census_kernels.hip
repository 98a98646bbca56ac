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

__global__ __launch_bounds__(kBlock) void k_bc_census(BcParams P, uint64_t n_done, const uint8_t* __restrict__ status,
                                                      unsigned long long* __restrict__ cells, unsigned long long* __restrict__ umis,
                                                      unsigned long long base, unsigned long long cap,
                                                      unsigned long long* __restrict__ count) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  const int lane = lane_id();
  for (uint64_t k0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); k0 < n_done; k0 += stride) {
    const uint64_t k = k0 + (uint64_t)lane;
    bool have = k < n_done && status[k] == kBcKeep;
    unsigned long long cell = 0, umi = 0;
    if (have) {
      uint32_t n = 0, cn = 0, qn = 0;
      const uint8_t *s = nullptr, *cs = nullptr, *q = nullptr;
      if (P.umi_read > 0) {
        BcLine ln[4];
        bc_lines(P.f[P.umi_read], k, ln);
        if (bc_get<false>(ln, (long)P.umi_off, (long)P.umi_size, P.phred, 0, &n, &qn, &s, &q) != 0) n = 0;
        if (n) n = census_value_len(s, n);
      }
      if (P.cell_read > 0) {
        BcLine ln[4];
        bc_lines(P.f[P.cell_read], k, ln);
        if (bc_get<false>(ln, (long)P.cell_off, (long)P.cell_size, P.phred, 0, &cn, &qn, &cs, &q) != 0) cn = 0;
        if (cn) cn = census_value_len(cs, cn);
      }
      // no UMI tag: bam_umi_count does not count the alignment (src/bam_umi_count.c:960); a value with a '_': no tags
      if (!n || n == ~0u || cn == ~0u) have = false;
      else {
        umi = census_pack(s, n);
        if (cn) cell = census_pack(cs, cn);
      }
    }
    const unsigned long long m = __ballot(have);
    if (!m) continue;
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(count, (unsigned long long)__builtin_popcountll(m));
    at = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(at >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at);
    if (have) {
      const unsigned long long i = base + at + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (i < cap) {
        cells[i] = cell;
        umis[i] = umi;
      }
    }
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
