// fqg_umi_cell_kernels.hip - process_entry (reference src/bam_umi_count.c:444-509) for CR-sorted input with unit
// increments: one WORKGROUP PER CELL, the cell's (feature, UMI) work done in LDS.
//
// In sorted mode the records of a cell are contiguous (the "sorted by CR" test of :1002-1008 has passed when this
// runs), and a (cell, feature, UMI) never crosses a cell.  The global hash tables of the general path
// (k_umi_count / k_umi_new: two random inserts per record into tables of 2 x N slots, beyond every cache) are
// replaced by a sort of the cell's records inside the workgroup:
//
//   key = feature id : 24 | (UMI id - 1) : 20 | index of the record in the cell : 20      (64 bits)
//
// sorted ascending (bitonic, LDS).  A record is the first of its (feature, UMI) - the one process_entry sees as a
// new UMI (:495-502) under set semantics - iff its key differs from its predecessor's above the index bits; the
// pairs (cell, feature) come out grouped and in feature order, their read / UMI counts are differences of
// prefix counts, and the sorted distinct UMIs of every pair (what fqg_rl_sim.h rebuilds an RL_Tree array from) are
// the run of first records.  Cells that do not fit the LDS buffer sort in a global scratch with the same code.
//
// Where the reference's RL_Tree would overwrite a node (fqg_rl_sim.h: rl_detect_step) is found here too, without
// the arrival order of whole pairs: an overwrite needs two UMIs of one pair in the same 64-block but different
// 16-leaves, so only such (pair, block) groups - adjacent in the sorted order - are walked in arrival order.
#include "fqg_rl_sim.h"

namespace fqg {

constexpr int kCellFeatBits = 24, kCellUmiBits = 20, kCellIdxBits = 20;
constexpr unsigned long long kCellPad = ~0ull;

struct CellArgs {
  uint32_t n;                    // records
  uint32_t n_cells;
  const uint32_t* feat;          // per record ids (0: not counted)
  const uint32_t* umi;
  const uint32_t* cell_first;    // [n_cells + 2]: first record of cell c (1-based); [n_cells + 1] = n
  uint8_t* is_new;               // per record: set semantics (the replay patches it afterwards)
  // per pair, worst-case layout: the pairs of cell c live at slots cell_first[c] .. in feature order
  KeySlot* pair_key;             // key = cell << 32 | feature, first = its first record
  uint32_t* pair_reads;
  uint32_t* pair_umis;
  uint32_t* pair_mem;            // where the pair's sorted distinct (UMI id - 1) start in members[]
  uint32_t* pair_pos;            // scratch: position of the pair's first record in the cell's sorted order
  uint32_t* members;             // [n]
  uint32_t* cell_pairs;          // [n_cells + 2]
  uint32_t* cell_reads;
  uint32_t* cell_umis;
  unsigned long long* big_keys;  // [2 n] sort buffer of cells that do not fit LDS (cell c at 2 * cell_first[c])
  uint32_t lds_keys;             // keys the LDS buffer holds (power of two)
  // RL_Tree overwrites: slot_hit[slot] = the earliest record of the pair whose insert overwrites (all ones: none);
  // flagged_slot[] lists the slots that have one
  uint32_t* slot_hit;
  uint32_t* flagged_slot;
  RlCall* rl;
  UmiCall* call;
};

// first record of every cell; the largest cell; (cells are numbered in order of first appearance, so in sorted
// input cell c owns the records from its first one up to the first one of cell c + 1)
__global__ __launch_bounds__(kBlock) void k_umi_cell_first(uint64_t n_slots, KeyTable C, Prefix pc, uint32_t n,
                                                           uint32_t* __restrict__ cell_first) {
  const uint64_t h = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h == 0) cell_first[0] = 0;
  if (h >= n_slots) return;
  const uint32_t f = C.first[h];
  if (f == kNoIdx) return;
  cell_first[pc.at(f) + 1u] = f;
}
// (n_cells is read from the device - tot[1], the scan's total - so that the kernel can be launched before the host has
// seen it: the grid covers n + 1 threads, the number of cells a call of n alignments can have at most)
__global__ __launch_bounds__(kBlock) void k_umi_cell_sizes(const unsigned long long* __restrict__ tot, uint32_t n,
                                                           uint32_t* __restrict__ cell_first, UmiCall* __restrict__ call) {
  const uint32_t n_cells = (uint32_t)tot[1];
  const uint32_t c = blockIdx.x * kBlock + threadIdx.x + 1u;
  if (c == n_cells + 1u) cell_first[c] = n;
  if (c > n_cells) return;
  const uint32_t end = c == n_cells ? n : cell_first[c + 1];
  atomicMax(&call->max_cell_records, end - cell_first[c]);
}

template <class T>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, T* s_tmp, uint32_t* total) {  // 256 threads
  s_tmp[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < kBlock; d <<= 1) {
    const uint32_t o = threadIdx.x >= (unsigned)d ? s_tmp[threadIdx.x - d] : 0u;
    __syncthreads();
    s_tmp[threadIdx.x] += o;
    __syncthreads();
  }
  const uint32_t incl = s_tmp[threadIdx.x];
  *total = s_tmp[kBlock - 1];
  __syncthreads();
  return incl - v;
}

// the work of one cell on its keys, which are in LDS (KP = an LDS pointer: ds_read / ds_write) or, for a cell that does
// not fit there, in a global scratch (KP = a plain pointer).  One body for a pointer that "may be either" would reach
// LDS through FLAT instructions - 700 of them per wavefront in the sort alone.
template <class KP>
__device__ __forceinline__ void umi_cell_body(const CellArgs& A, KP keys, uint32_t c, uint32_t s0, uint32_t m, uint32_t* s_tmp,
                                              uint32_t& s_n) {
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  // ---- the counted records of the cell ----
  for (uint32_t k = threadIdx.x; k < m; k += kBlock) {
    const uint32_t f = A.feat[s0 + k];
    if (!f) {
      A.is_new[s0 + k] = 0;
      continue;
    }
    const uint32_t at = atomicAdd(&s_n, 1u);
    keys[at] = ((unsigned long long)f << (kCellUmiBits + kCellIdxBits)) |
               ((unsigned long long)(A.umi[s0 + k] - 1u) << kCellIdxBits) | k;
  }
  __syncthreads();
  const uint32_t nc = s_n;
  if (threadIdx.x == 0) {
    A.cell_reads[c] = nc;
    if (!nc) {
      A.cell_pairs[c] = 0;
      A.cell_umis[c] = 0;
    }
  }
  if (!nc) return;
  uint32_t P = 64;
  while (P < nc) P <<= 1;
  for (uint32_t k = nc + threadIdx.x; k < P; k += kBlock) keys[k] = kCellPad;
  __syncthreads();
  // ---- bitonic sort ----
  for (uint32_t kk = 2; kk <= P; kk <<= 1)
    for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < P; i += kBlock) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          if ((a > b) == ((i & kk) == 0)) {
            keys[i] = b;
            keys[l] = a;
          }
        }
      }
      __syncthreads();
    }
  // ---- first records of triples and pairs: a contiguous piece per thread, prefix counts over the block ----
  const uint32_t per = (nc + kBlock - 1) / kBlock;
  const uint32_t a = threadIdx.x * per, b = a + per < nc ? a + per : nc;
  uint32_t my_t = 0, my_p = 0;
  for (uint32_t j = a; j < b; ++j) {
    const unsigned long long k1 = keys[j], k0 = j ? keys[j - 1] : kCellPad;
    my_t += (k1 >> kCellIdxBits) != (k0 >> kCellIdxBits);
    my_p += (k1 >> (kCellUmiBits + kCellIdxBits)) != (k0 >> (kCellUmiBits + kCellIdxBits));
  }
  uint32_t n_trip, n_pair;
  uint32_t t_before = block_scan_excl(my_t, s_tmp, &n_trip);
  uint32_t p_before = block_scan_excl(my_p, s_tmp, &n_pair);
  if (threadIdx.x == 0) {
    A.cell_pairs[c] = n_pair;
    A.cell_umis[c] = n_trip;
    atomicAdd(&A.call->spread[1][blockIdx.x & 63], (unsigned long long)nc);
    atomicAdd(&A.call->spread[2][blockIdx.x & 63], (unsigned long long)n_trip);
  }
  // per record: is_new; per triple start: a member; per pair start: key, first record, where its members start and
  // its position in the sorted order (pair_pos) - reads / UMIs follow below as differences to the next pair
  for (uint32_t j = a; j < b; ++j) {
    const unsigned long long k1 = keys[j], k0 = j ? keys[j - 1] : kCellPad;
    const bool ts = (k1 >> kCellIdxBits) != (k0 >> kCellIdxBits);
    const bool ps = (k1 >> (kCellUmiBits + kCellIdxBits)) != (k0 >> (kCellUmiBits + kCellIdxBits));
    const uint32_t rec = s0 + (uint32_t)(k1 & ((1u << kCellIdxBits) - 1u));
    A.is_new[rec] = ts ? 1 : 0;
    if (ts) A.members[s0 + t_before] = (uint32_t)(k1 >> kCellIdxBits) & ((1u << kCellUmiBits) - 1u);
    if (ps) {
      const uint32_t slot = s0 + p_before;
      KeySlot ks;
      ks.key = ((unsigned long long)c << 32) | (uint32_t)(k1 >> (kCellUmiBits + kCellIdxBits));
      ks.first = rec;  // (smallest index of the pair's smallest UMI: only used as "some record of the pair")
      ks.pad = 0;
      A.pair_key[slot] = ks;
      A.pair_mem[slot] = s0 + t_before;
      A.pair_pos[slot] = j;
      ++p_before;
    }
    t_before += ts;
  }
  __syncthreads();
  // reads / UMIs of a pair = distance to the next pair's first record / first member (the cell's end for the last)
  for (uint32_t k = threadIdx.x; k < n_pair; k += kBlock) {
    const uint32_t slot = s0 + k;
    const bool last = k + 1 == n_pair;
    A.pair_reads[slot] = (last ? nc : A.pair_pos[slot + 1]) - A.pair_pos[slot];
    A.pair_umis[slot] = (last ? s0 + n_trip : A.pair_mem[slot + 1]) - A.pair_mem[slot];
  }
  // ---- RL_Tree: (pair, 64-block) groups with UMIs in more than one 16-leaf -> arrival-order walk ----
  for (uint32_t j = a; j < b; ++j) {
    const unsigned long long k1 = keys[j], k0 = j ? keys[j - 1] : kCellPad;
    // j starts a (pair, block) group among the FIRST records of triples?
    if ((k1 >> kCellIdxBits) == (k0 >> kCellIdxBits)) continue;                    // not a first record
    const unsigned long long grp = k1 >> (kCellIdxBits + 6);                        // feature | block
    // previous DISTINCT member: keys[j - 1] belongs to it (any record of it)
    if (j && (k0 >> (kCellIdxBits + 6)) == grp) continue;                            // the group started earlier
    // members of the group: first records at j, then every later triple start with the same grp
    uint32_t leaves = 0, cnt = 0;
    uint32_t e = j;
    while (e < nc && (keys[e] >> (kCellIdxBits + 6)) == grp) {
      if (e == j || (keys[e] >> kCellIdxBits) != (keys[e - 1] >> kCellIdxBits)) {
        leaves |= 1u << ((uint32_t)(keys[e] >> (kCellIdxBits + 4)) & 3u);
        ++cnt;
      }
      ++e;
    }
    if (cnt < 2 || !(leaves & (leaves - 1u))) continue;  // one member, or all in one leaf: no overwrite possible
    // earliest arrival of a member of the same pair in a HIGHER block: from then on this block is not the
    // block of the largest member and cannot overwrite
    const unsigned long long pair = k1 >> (kCellUmiBits + kCellIdxBits);
    uint32_t hi_arrival = ~0u;
    for (uint32_t q = e; q < nc && (keys[q] >> (kCellUmiBits + kCellIdxBits)) == pair; ++q)
      if ((keys[q] >> kCellIdxBits) != (keys[q - 1] >> kCellIdxBits)) {
        const uint32_t arr = (uint32_t)(keys[q] & ((1u << kCellIdxBits) - 1u));
        hi_arrival = arr < hi_arrival ? arr : hi_arrival;
      }
    // walk the group's members in arrival order (selection by smallest arrival above the last one taken)
    uint32_t occ = 0, last = 0;
    bool first = true, hit = false;
    uint32_t hit_rec = 0;
    for (uint32_t step = 0; step < cnt && !hit; ++step) {
      uint32_t best = ~0u, best_leaf = 0;
      for (uint32_t q = j; q < e; ++q) {
        if (q != j && (keys[q] >> kCellIdxBits) == (keys[q - 1] >> kCellIdxBits)) continue;
        const uint32_t arr = (uint32_t)(keys[q] & ((1u << kCellIdxBits) - 1u));
        if ((first || arr > last) && arr < best) {
          best = arr;
          best_leaf = (uint32_t)(keys[q] >> (kCellIdxBits + 4)) & 3u;
        }
      }
      if (best == ~0u || best > hi_arrival) break;
      first = false;
      last = best;
      if (!occ) occ = 1u << best_leaf;
      else if (!(occ & (1u << best_leaf))) {
        const uint32_t above = occ >> (best_leaf + 1u);
        if (above && !(above & (above - 1u))) {
          hit = true;
          hit_rec = s0 + best;
        } else occ |= 1u << best_leaf;
      }
    }
    if (hit) {
      // the pair's slot: pairs before this one in the cell = pair starts before j
      uint32_t rank = 0;
      for (uint32_t q = 1; q <= j; ++q)
        rank += (keys[q] >> (kCellUmiBits + kCellIdxBits)) != (keys[q - 1] >> (kCellUmiBits + kCellIdxBits));
      const uint32_t old = atomicMin(&A.slot_hit[s0 + rank], hit_rec);  // (several blocks of one pair may hit)
      if (old == ~0u) A.flagged_slot[atomicAdd(&A.rl->n_flagged, 1u)] = s0 + rank;
    }
  }
}

typedef __attribute__((address_space(3))) unsigned long long* CellKeysLds;
__global__ __launch_bounds__(kBlock) void k_umi_cells(CellArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
  __shared__ uint32_t s_tmp[kBlock];
  __shared__ uint32_t s_n;
  const uint32_t c = blockIdx.x + 1u;
  if (c > A.n_cells) return;
  const uint32_t s0 = A.cell_first[c], s1 = A.cell_first[c + 1], m = s1 - s0;
  if (m <= A.lds_keys) umi_cell_body(A, (CellKeysLds)s_keys, c, s0, m, s_tmp, s_n);
  else umi_cell_body(A, A.big_keys + 2ull * s0, c, s0, m, s_tmp, s_n);
}

// pair_of[] (pairs grouped by cell, as the output kernels walk them) for the worst-case pair layout
__global__ __launch_bounds__(kBlock) void k_umi_cell_pairs_fill(uint32_t n_cells, const uint32_t* __restrict__ cell_first,
                                                                const uint32_t* __restrict__ cell_pairs, Prefix start,
                                                                uint32_t* __restrict__ pair_of,
                                                                uint32_t* __restrict__ run_of_slot) {
  const uint32_t c = blockIdx.x + 1u;
  if (c > n_cells) return;
  const uint32_t np = cell_pairs[c], s0 = cell_first[c], p0 = start.at(c);
  for (uint32_t k = threadIdx.x; k < np; k += kBlock) {
    pair_of[p0 + k] = s0 + k;
    run_of_slot[s0 + k] = p0 + k;
  }
}

// unit increments: the counts of a cell's pairs as the float32 values the reference would hold
__global__ __launch_bounds__(kBlock) void k_umi_cell_unit_sums(uint32_t n_cells, const uint32_t* __restrict__ cell_first,
                                                               const uint32_t* __restrict__ cell_pairs,
                                                               const uint32_t* __restrict__ reads, const uint32_t* __restrict__ umis,
                                                               float* __restrict__ f_reads, float* __restrict__ f_umis) {
  const uint32_t c = blockIdx.x + 1u;
  if (c > n_cells) return;
  const uint32_t s0 = cell_first[c], np = cell_pairs[c];
  for (uint32_t k = threadIdx.x; k < np; k += kBlock) {
    f_reads[s0 + k] = unit_sum(reads[s0 + k]);
    f_umis[s0 + k] = unit_sum(umis[s0 + k]);
  }
}

// ---- RL_Tree replay on (cell, feature) runs: what k_rl_replay (by_cell) reads per run ----
// The replay looks back along a feature's earlier (cell, feature) pairs - but only for the features that have a
// flagged pair at all (63 of 20 000 on BASELINE.json configs[3]).  k_rl_cell_flags marks those features,
// k_rl_cell_chain keeps the pairs ("runs") of marked features: their description for the replay (records of the cell,
// members) and an entry (feature << 32 | cell, run) in the list that is then sorted into the chains.
__global__ __launch_bounds__(kBlock) void k_rl_cell_flags(uint32_t n_flagged, const uint32_t* __restrict__ flagged_slot,
                                                          const uint32_t* __restrict__ slot_hit,
                                                          const uint32_t* __restrict__ run_of_slot,
                                                          const KeySlot* __restrict__ pair_key,
                                                          const uint32_t* __restrict__ pair_reads, RlRuns runs,
                                                          uint32_t* __restrict__ flagged, uint32_t* __restrict__ flag_k0,
                                                          uint32_t* __restrict__ flag_len, uint8_t* __restrict__ feat_flag,
                                                          RlCall* __restrict__ call) {
  const uint32_t fi = blockIdx.x * kBlock + threadIdx.x;
  if (fi >= n_flagged) return;
  const uint32_t slot = flagged_slot[fi], run = run_of_slot[slot];
  flagged[fi] = run;
  runs.flag[run] = fi;
  flag_k0[fi] = slot_hit[slot];       // the record whose insert overwrites: records below it precede it
  const uint32_t len = pair_reads[slot];
  flag_len[fi] = len;
  feat_flag[(uint32_t)pair_key[slot].key] = 1;
  atomicMax(&call->max_flagged_len, len);
}
__global__ __launch_bounds__(kBlock) void k_rl_cell_chain(uint32_t n_runs, const uint32_t* __restrict__ pair_of,
                                                          const KeySlot* __restrict__ pair_key,
                                                          const uint8_t* __restrict__ feat_flag,
                                                          const uint32_t* __restrict__ cell_first,
                                                          const uint32_t* __restrict__ pair_mem,
                                                          const uint32_t* __restrict__ pair_umis, RlRuns runs,
                                                          uint32_t* __restrict__ run_feat, uint32_t* __restrict__ run_mem,
                                                          uint32_t* __restrict__ run_nmem,
                                                          unsigned long long* __restrict__ chain_key,
                                                          uint32_t* __restrict__ chain_run, RlCall* __restrict__ call) {
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  bool keep = false;
  unsigned long long key = 0;
  uint32_t slot = 0;
  if (r < n_runs) {
    slot = pair_of[r];
    key = pair_key[slot].key;  // cell << 32 | feature
    keep = feat_flag[(uint32_t)key] != 0;
  }
  // one atomic per WORKGROUP for the place in the list (25 000 wavefronts asking one address for a return value
  // take longer than everything else this kernel does)
  __shared__ uint32_t s_cnt[kBlock / kWave], s_base;
  const unsigned long long m = __ballot(keep);
  const int wv = (int)(threadIdx.x >> 6);
  if ((threadIdx.x & 63) == 0) s_cnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) tot += s_cnt[w];
    s_base = tot ? atomicAdd(&call->n_chain, tot) : 0u;
  }
  __syncthreads();
  if (!keep) return;
  uint32_t at = s_base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
  for (int w = 0; w < wv; ++w) at += s_cnt[w];
  chain_key[at] = (key << 32) | (key >> 32);
  chain_run[at] = r;
  const uint32_t cell = (uint32_t)(key >> 32);
  runs.start[r] = cell_first[cell];
  runs.len[r] = cell_first[cell + 1] - cell_first[cell];
  run_feat[r] = (uint32_t)key;
  run_mem[r] = pair_mem[slot];
  const uint32_t nm = pair_umis[slot];  // (set semantics: distinct members; read before any replay patches it)
  run_nmem[r] = nm;
  if (nm > call->max_len) atomicMax(&call->max_len, nm);
}

}  // namespace fqg
