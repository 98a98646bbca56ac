// TEST DRIVER (CPU): runs fastq_utils_amd/csrc/fqg_rl_sim.h - the code the GPU chain kernel executes - with a
// one-lane "wavefront" so that tests/test_rl_sim.py can compare it with the oracle (oracle/rl_oracle.c) here,
// without a GPU.  Not part of the product: libfqgpu.so only instantiates the 64-lane device policy.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "../../fastq_utils_amd/csrc/fqg_rl_sim.h"

namespace {
struct CpuWave {
  typedef uint16_t* p16;
  typedef uint32_t* p32;
  static constexpr int lanes = 1;
  static uint32_t lane() { return 0; }
  static void sync() {}
  static uint32_t rank(bool) { return 0; }
  static uint32_t count(bool b) { return b ? 1u : 0u; }
  static uint32_t bcast(uint32_t v, uint32_t) { return v; }
  static uint32_t uni(uint32_t v) { return v; }
  static uint32_t find_first(bool b) { return b ? 0u : 0xFFFFFFFFu; }
  static uint32_t wait_nonzero(const uint32_t* p) { return *p; }  // (runs are replayed in chain order here)
  static void publish(uint32_t* p, uint32_t v) { *p = v; }
  static void fence() {}
  static unsigned long long clock() { return 0; }
};
}  // namespace

// chain[i]: tree the record touches (gene id in sorted mode, a (cell, gene) number in unsorted mode);
// epoch[i]: the cell (records of one cell are contiguous in sorted mode).  Epochs of a chain in file order.
// out_new[i]: the decision for record i; stats[0..5] = undefined, overwrites, wild writes, overflow, changed,
// flagged runs.  cap / mcap: array sizes of the worker (to exercise the overflow paths).
extern "C" int rl_sim_check(uint32_t n, const uint32_t* chain, const uint32_t* epoch, const uint32_t* umi,
                            uint32_t cap, uint32_t mcap, int history, int from_overwrite, int by_cell, uint8_t* out_new, uint64_t* stats) {
  using namespace fqg::rl;
  // runs = (chain, epoch) groups in order of first appearance; order[] = records grouped by run
  std::map<std::pair<uint32_t, uint32_t>, uint32_t> run_of;
  std::vector<std::vector<uint32_t>> recs;
  std::vector<std::pair<uint32_t, uint32_t>> key;
  for (uint32_t i = 0; i < n; ++i) {
    auto k = std::make_pair(chain[i], epoch[i]);
    auto it = run_of.find(k);
    if (it == run_of.end()) {
      it = run_of.emplace(k, (uint32_t)recs.size()).first;
      recs.emplace_back();
      key.push_back(k);
    }
    recs[it->second].push_back(i);
  }
  const uint32_t n_runs = (uint32_t)recs.size();
  std::vector<uint32_t> order, run_start(n_runs), run_len(n_runs), run_flag(n_runs, kNone);
  // by_cell: runs as fqg_umi_cell_kernels.hip hands them over (records of a cell are contiguous in the input)
  std::vector<uint32_t> run_feat(n_runs), run_mem(n_runs), run_nmem(n_runs), members, cell_lo(n_runs), cell_n(n_runs);
  if (by_cell) {
    std::map<uint32_t, std::pair<uint32_t, uint32_t>> span;  // epoch -> [first record, one past the last]
    for (uint32_t i = 0; i < n; ++i) {
      auto it = span.find(epoch[i]);
      if (it == span.end()) span[epoch[i]] = {i, i + 1};
      else it->second.second = i + 1;
    }
    for (uint32_t r = 0; r < n_runs; ++r) {
      cell_lo[r] = span[key[r].second].first;
      cell_n[r] = span[key[r].second].second - cell_lo[r];
      run_feat[r] = key[r].first;
    }
  }
  std::vector<uint8_t> set_new(n, 0);
  std::vector<uint32_t> flagged, flag_k0;
  for (uint32_t r = 0; r < n_runs; ++r) {
    run_start[r] = (uint32_t)order.size();
    run_len[r] = (uint32_t)recs[r].size();
    std::set<uint32_t> seen;
    uint32_t state = 0, k = 0, k0 = 0;
    bool hit = false;
    for (uint32_t i : recs[r]) {
      order.push_back(i);
      if (seen.insert(umi[i]).second) {
        set_new[i] = 1;
        if (!hit && rl_detect_step(state, umi[i])) {
          hit = true;
          k0 = by_cell ? i : k;
        }
      }
      ++k;
    }
    if (by_cell) {
      run_mem[r] = (uint32_t)members.size();
      run_nmem[r] = (uint32_t)seen.size();
      for (uint32_t u : seen) members.push_back(u - 1);  // (std::set: ascending)
      run_start[r] = cell_lo[r];
      run_len[r] = cell_n[r];
    }
    if (hit) {
      run_flag[r] = (uint32_t)flagged.size();
      flagged.push_back(r);
      flag_k0.push_back(from_overwrite ? k0 : 0);
    }
  }
  memcpy(out_new, set_new.data(), n);
  // all runs ordered by (chain, first appearance): a chain's runs are in cell order for sorted input
  std::vector<uint32_t> chain_runs(n_runs);
  for (uint32_t r = 0; r < n_runs; ++r) chain_runs[r] = r;
  std::stable_sort(chain_runs.begin(), chain_runs.end(), [&](uint32_t a, uint32_t b) { return key[a].first < key[b].first; });
  std::vector<unsigned long long> chain_key(n_runs);
  for (uint32_t p = 0; p < n_runs; ++p) chain_key[p] = ((unsigned long long)key[chain_runs[p]].first << 32) | p;
  std::vector<uint32_t> flag_off(flagged.size()), flag_ext(flagged.size(), 0);
  uint64_t arena_n = 0;
  for (size_t f = 0; f < flagged.size(); ++f) {
    flag_off[f] = (uint32_t)arena_n;
    arena_n += cap;
  }
  std::vector<uint16_t> arena(arena_n + 1);
  std::vector<uint16_t> node(cap), stale(cap);
  std::vector<uint32_t> known(cap / 32 + 1), mem(mcap + 1), base(mcap + 2), scratch(4);
  std::vector<uint32_t> memo(kMemoWords);
  WorkT<CpuWave> wk{node.data(), known.data(), stale.data(), mem.data(), base.data(), scratch.data(), cap, mcap, memo.data()};
  Stats st{};
  for (uint32_t p = 0; p < n_runs; ++p) {  // flagged runs in (chain, cell) order: the order the GPU hands them out
    const uint32_t r = chain_runs[p];
    if (run_flag[r] == kNone) continue;
    ChainView cv{history ? chain_runs.data() : nullptr, chain_key.data(), p, r, run_start.data(), run_len.data(),
                 order.data(), umi, out_new, run_flag.data(), flag_k0.data(), flag_off.data(), flag_ext.data(), arena.data(),
                 by_cell, chain, run_feat.data(), run_mem.data(), run_nmem.data(), members.data()};
    replay_run<CpuWave>(cv, wk, st, out_new, [](uint32_t, uint8_t, uint32_t) {});
  }
  stats[0] = st.undefined;
  stats[1] = st.overwrites;
  stats[2] = st.wild_writes;
  stats[3] = st.overflow;
  stats[4] = st.changed;
  stats[5] = flagged.size();
  stats[6] = st.lookback_runs;
  return 0;
}
