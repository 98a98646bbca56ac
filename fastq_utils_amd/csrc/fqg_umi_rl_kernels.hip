// fqg_umi_rl_kernels.hip - bam_umi_count's distinct-UMI decision where the reference's RL_Tree is not a set
// (reference src/range_list.c; the algorithm and its derivation are in fqg_rl_sim.h).
//
// The counting kernels (fqg_umi_kernels.hip) decide "new UMI" with set semantics.  That is what the reference
// computes for every (cell, gene) epoch in which its tree never overwrites its last node - on
// BASELINE.json configs[3] all but 63 of 766 569.  This stage finds the others and replays them:
//
//   k_rl_starts / k_rl_detect   records sorted by (pair slot, record index) [rocPRIM]; one thread per run
//                    feeds the run's new UMIs, in arrival order, through rl_detect_step (4 bits of state)
//   k_rl_chain_keys / k_rl_positions / k_rl_item_keys   the runs of one gene in cell order form a chain (the gene's
//                    tree lives for the whole file); flagged runs are ordered by their place in their chain, so that
//                    a run a replay may have to wait for always holds an earlier ticket
//   k_rl_replay      one wavefront per flagged run (tickets in that order): exact replay on the node array (LDS),
//                    earlier cells' arrays rebuilt on demand or taken from the arena where an earlier replay left
//                    them; patches is_new[] and the counters
#include "fqg_rl_sim.h"

namespace fqg {

struct GpuWaveBase {
  static constexpr int lanes = kWave;
  static __device__ __forceinline__ uint32_t lane() { return threadIdx.x; }
  static __device__ __forceinline__ void sync() { __syncthreads(); }
  static __device__ __forceinline__ uint32_t rank(bool b) {
    const unsigned long long m = __ballot(b);
    return (uint32_t)__popcll(m & ((1ull << threadIdx.x) - 1ull));
  }
  static __device__ __forceinline__ uint32_t count(bool b) { return (uint32_t)__popcll(__ballot(b)); }
  // the lowest lane whose b is set, or rl::kNone
  static __device__ __forceinline__ uint32_t find_first(bool b) {
    const unsigned long long m = __ballot(b);
    return m ? (uint32_t)__builtin_ctzll(m) : 0xFFFFFFFFu;
  }
  // v, which is the same in every lane, as a scalar
  static __device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
  // lane `src`'s value of v (src is the same in every lane)
  static __device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)src));
  }
  // the final array of an earlier replayed run of the same chain: its worker took its ticket before ours, so it
  // is running (or done) and this wait ends
  static __device__ __forceinline__ uint32_t wait_nonzero(const uint32_t* p) {
    uint32_t v;
    while (!(v = __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT))) __builtin_amdgcn_s_sleep(8);
    return v;
  }
  static __device__ __forceinline__ void publish(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  static __device__ __forceinline__ void fence() { __threadfence(); }
  static __device__ __forceinline__ unsigned long long clock() { return wall_clock64(); }
};
// where the worker's arrays live: LDS (the usual case: ds_read / ds_write) or a global scratch (flagged sets too
// large for LDS)
struct GpuWaveLds : GpuWaveBase {
  typedef __attribute__((address_space(3))) uint16_t* p16;
  typedef __attribute__((address_space(3))) uint32_t* p32;
};
struct GpuWaveGlobal : GpuWaveBase {
  typedef uint16_t* p16;
  typedef uint32_t* p32;
};

struct RlCall {
  uint32_t n_flagged, next_item, max_len, max_flagged_len;
  uint32_t undefined, overwrites, wild_writes, overflow, changed, lookback_runs;
  uint32_t n_chain, pad_;   // sorted mode by cell: runs of the features that have a flagged run
  unsigned long long clk_replay, clk_lookback, clk_store;  // wall_clock64 ticks (100 MHz) summed over the workers
  unsigned long long clk_build, clk_loop, clk_max, clk_max_wait;
};

struct RlRuns {      // per run (= per (cell, feature) pair), in order of the sorted pair slots
  uint32_t* start;   // first position in the sorted order
  uint32_t* len;
  uint32_t* pslot;
  uint32_t* flag;    // index into the flagged list, or kNone (host fills with 0xFF)
};

__global__ __launch_bounds__(kBlock) void k_rl_starts(uint32_t n, const uint32_t* __restrict__ key,
                                                      uint32_t* __restrict__ flag) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  const uint32_t k = key[j];
  flag[j] = (k != kNoIdx && (j == 0 || key[j - 1] != k)) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void k_rl_detect(uint32_t n, const uint32_t* __restrict__ key,
                                                      const uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ start_flag, Prefix run_of,
                                                      const uint32_t* __restrict__ umi_id,
                                                      const uint8_t* __restrict__ is_new, RlRuns runs,
                                                      uint32_t* __restrict__ flagged, uint32_t* __restrict__ flag_k0,
                                                      RlCall* __restrict__ call) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= n || !start_flag[j]) return;
  const uint32_t r = run_of.at(j), k = key[j];
  uint32_t state = 0, len = 0, k0 = 0;
  bool hit = false;
  for (uint32_t p = j; p < n && key[p] == k; ++p, ++len) {
    const uint32_t rec = order[p];
    if (!hit && is_new[rec] && rl::rl_detect_step(state, umi_id[rec])) {
      hit = true;
      k0 = len;  // records of the run before the one whose insert overwrites
    }
  }
  runs.start[r] = j;
  runs.len[r] = len;
  runs.pslot[r] = k;
  if (len > call->max_len) atomicMax(&call->max_len, len);
  if (hit) {
    const uint32_t fi = atomicAdd(&call->n_flagged, 1u);
    flagged[fi] = r;
    flag_k0[fi] = k0;
    runs.flag[r] = fi;
    atomicMax(&call->max_flagged_len, len);
  }
}

// sorted mode: chain = feature; key = feature << 32 | cell puts a feature's runs in cell order
__global__ __launch_bounds__(kBlock) void k_rl_chain_keys(uint32_t n_runs, const uint32_t* __restrict__ run_pslot,
                                                          PairTable Pt, unsigned long long* __restrict__ key,
                                                          uint32_t* __restrict__ val) {
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n_runs) return;
  const unsigned long long k = Pt.t.s[run_pslot[r]].key;  // cell << 32 | feature
  key[r] = (k << 32) | (k >> 32);
  val[r] = r;
}
__global__ __launch_bounds__(kBlock) void k_rl_positions(uint32_t n_runs, const uint32_t* __restrict__ chain_runs,
                                                         uint32_t* __restrict__ pos_of_run) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p < n_runs) pos_of_run[chain_runs[p]] = p;
}
// replay order: flagged runs by position in the (chain, cell) order, so that a run a replay may have to wait
// for always holds an earlier ticket
__global__ __launch_bounds__(kBlock) void k_rl_item_keys(uint32_t n_flagged, const uint32_t* __restrict__ flagged,
                                                         const uint32_t* __restrict__ pos_of_run,
                                                         uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t fi = blockIdx.x * kBlock + threadIdx.x;
  if (fi >= n_flagged) return;
  key[fi] = pos_of_run[flagged[fi]];
  val[fi] = fi;
}

struct RlReplayArgs {
  const uint32_t* chain_runs;          // sorted mode: run ids by (feature, cell); unsorted mode: null (no history)
  const unsigned long long* chain_key; // feature << 32 | cell per position of chain_runs
  const uint32_t* pos_of_run;
  const uint32_t* items;               // flagged indices in replay order (null: 0, 1, 2, ...)
  const uint32_t* flagged;             // flagged index -> run
  uint32_t n_items;
  RlRuns runs;
  const uint32_t* order;
  const uint32_t* umi_id;
  const uint32_t* cell_id;
  uint8_t* is_new;
  const uint32_t* flag_k0;
  const uint32_t* flag_off;
  uint32_t* flag_ext;
  uint16_t* arena;
  uint32_t cap, mcap;            // nodes of the replayed array / members of a looked-up run
  // by_cell: runs are (cell, feature) pairs of CR-sorted input (fqg_umi_cell_kernels.hip)
  int by_cell;
  const uint32_t* rec_feat;
  const uint32_t* run_feat;
  const uint32_t* run_mem;
  const uint32_t* run_nmem;
  const uint32_t* members;
  int in_lds;                    // the worker's arrays fit the LDS budget
  uint8_t* scratch;              // else: per workgroup node | stale | known | mem | base
  uint64_t scratch_stride;
  // counters to patch
  uint32_t* pair_umis;
  uint32_t* cell_umis;
  unsigned long long* n_new_spread;  // 64 copies
  RlCall* call;
};

template <class W, class P8>
__device__ __forceinline__ void rl_replay_body(const RlReplayArgs& A, P8 g, typename W::p32 scan, uint32_t* s_item) {
  auto carve = [&](uint64_t bytes) {
    P8 p = g;
    g += (bytes + 15) & ~15ull;
    return p;
  };
  rl::WorkT<W> wk;
  wk.cap = A.cap;
  wk.mcap = A.mcap;
  wk.node = (typename W::p16)carve((uint64_t)A.cap * 2);
  wk.stale = (typename W::p16)carve((uint64_t)A.cap * 2);
  wk.known = (typename W::p32)carve((uint64_t)A.cap / 8 + 4);
  wk.mem = (typename W::p32)carve((uint64_t)A.mcap * 4);
  wk.base = (typename W::p32)carve((uint64_t)(A.mcap + 1) * 4);
  wk.memo = (typename W::p32)carve((uint64_t)rl::kMemoWords * 4);
  wk.scratch = scan;
  rl::Stats st{};
  for (;;) {  // tickets in replay order
    if (threadIdx.x == 0) *s_item = atomicAdd(&A.call->next_item, 1u);
    __syncthreads();
    const uint32_t item = *s_item;
    __syncthreads();
    if (item >= A.n_items) break;
    const uint32_t fi = A.items ? A.items[item] : item;
    rl::ChainView cv;
    cv.run = A.flagged[fi];
    cv.chain_runs = A.chain_runs;
    cv.chain_key = A.chain_key;
    cv.pos = A.pos_of_run ? A.pos_of_run[cv.run] : 0u;
    cv.run_start = A.runs.start;
    cv.run_len = A.runs.len;
    cv.order = A.order;
    cv.umi_id = A.umi_id;
    cv.set_new = A.is_new;
    cv.run_flag = A.runs.flag;
    cv.flag_k0 = A.flag_k0;
    cv.flag_off = A.flag_off;
    cv.flag_ext = A.flag_ext;
    cv.arena = A.arena;
    cv.by_cell = A.by_cell;
    cv.rec_feat = A.rec_feat;
    cv.run_feat = A.run_feat;
    cv.run_mem = A.run_mem;
    cv.run_nmem = A.run_nmem;
    cv.members = A.members;
    rl::replay_run<W>(cv, wk, st, A.is_new, [&](uint32_t rec, uint8_t nw, uint32_t run) {
      const uint32_t d = nw ? 1u : 0xFFFFFFFFu;  // +1 / -1
      atomicAdd(&A.pair_umis[A.runs.pslot[run]], d);
      atomicAdd(&A.cell_umis[A.cell_id[rec]], d);
      atomicAdd(&A.n_new_spread[blockIdx.x & 63], nw ? 1ull : ~0ull);
    });
  }
  if (threadIdx.x == 0) {
    if (st.undefined) atomicAdd(&A.call->undefined, st.undefined);
    if (st.overwrites) atomicAdd(&A.call->overwrites, st.overwrites);
    if (st.wild_writes) atomicAdd(&A.call->wild_writes, st.wild_writes);
    if (st.overflow) atomicOr(&A.call->overflow, 1u);
    if (st.changed) atomicAdd(&A.call->changed, st.changed);
    if (st.lookback_runs) atomicAdd(&A.call->lookback_runs, st.lookback_runs);
    atomicAdd(&A.call->clk_replay, st.clk_replay);
    atomicAdd(&A.call->clk_lookback, st.clk_lookback);
    atomicAdd(&A.call->clk_store, st.clk_store);
    atomicAdd(&A.call->clk_build, st.clk_build);
    atomicAdd(&A.call->clk_loop, st.clk_loop);
    atomicMax(&A.call->clk_max, st.clk_max);
    atomicMax(&A.call->clk_max_wait, st.clk_max_wait);
  }
}

__global__ __launch_bounds__(kWave) void k_rl_replay(RlReplayArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  __shared__ uint32_t s_scan[kWave + 1];
  __shared__ uint32_t s_item;
  if (A.in_lds) {
    typedef __attribute__((address_space(3))) uint8_t* lds8;
    rl_replay_body<GpuWaveLds, lds8>(A, (lds8)s_dyn, (GpuWaveLds::p32)s_scan, &s_item);
  } else {
    rl_replay_body<GpuWaveGlobal, uint8_t*>(A, A.scratch + (uint64_t)blockIdx.x * A.scratch_stride, (uint32_t*)s_scan, &s_item);
  }
}

// bytes of a worker's arrays (the carve-up of k_rl_replay)
static inline uint64_t rl_work_bytes(uint32_t cap, uint32_t mcap) {
  auto r = [](uint64_t b) { return (b + 15) & ~15ull; };
  return 2 * r((uint64_t)cap * 2) + r((uint64_t)cap / 8 + 4) + r((uint64_t)mcap * 4) + r((uint64_t)(mcap + 1) * 4) +
         r((uint64_t)rl::kMemoWords * 4);
}

__global__ __launch_bounds__(kBlock) void k_rl_flag_lens(uint32_t n_flagged, const uint32_t* __restrict__ flagged,
                                                         const uint32_t* __restrict__ run_len,
                                                         uint32_t* __restrict__ out) {
  const uint32_t fi = blockIdx.x * kBlock + threadIdx.x;
  if (fi < n_flagged) out[fi] = run_len[flagged[fi]];
}

}  // namespace fqg
