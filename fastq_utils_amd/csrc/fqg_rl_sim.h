// fqg_rl_sim.h - the reference's UMI container as it BEHAVES (reference src/range_list.c, 0.25.3), for the
// few (cell, gene) epochs in which it differs from a set.
//
// bam_umi_count keeps the UMI ids of a (cell, gene) in an RL_Tree: a 4-ary tree over [1..1048576]
// (src/bam_umi_count.c:48,479) stored as ONE array of 16-bit nodes in pre-order (src/range_list.h:35-41,
// 96-103).  An inner node = four 2-bit quadrant states (quadrant 1 in bits 0-1; 0 = empty, 2 = has a child
// node, 3 = full) + an 8-bit count of the nodes of its subtree that saturates at 255; a leaf = a bitmap of
// 16 numbers.  Nine levels: the root (depth 0), inner nodes at depths 1..7 (widths 4^9 .. 4^3 = 64) and
// leaves at depth 8.  In sorted mode the tree of a gene lives for the whole file: rl_all(OUT) between two
// cells (quick_reset_db, src/bam_umi_count.c:418-441 -> src/range_list.c:187-198) resets the root's
// quadrants and size = 1 and leaves every other array slot as it is.
//
// Where it is not a set (all restated exactly below, as in oracle/rl_oracle.c, which is the checker):
//   * new_node (src/range_list.c:325-372) opens a gap with shift_right (:287-301), which moves NOTHING when
//     exactly one node lies at / behind the insertion point: the new node overwrites the array's last
//     node, the array still grows by one, and its new last slot keeps what the memory held - the node
//     some EARLIER cell left there.  From then on membership answers depend on stale bytes.
//   * the count refresh of a saturated node is computed one level short (:485).
//
// While no overwrite has happened in an epoch (one gene in one cell), the array is exactly the pre-order
// trie of the set of members, whatever the array held before.  So:
//   detect  (rl_detect_step)  the first overwrite of an epoch is a function of the arrival order of its
//           members alone: a new member x opens a leaf directly in front of the array's last node iff
//           it falls into the 64-block of the largest member so far, into a leaf of that block that is
//           still empty, and exactly one occupied leaf of that block lies above it.
//   replay  (Sim)  only flagged epochs are replayed on the array, node for node.  What the array held
//           before the epoch (slots the epoch reads before writing them) comes from
//   history (History)  the final arrays of the gene's earlier epochs, newest first: slot k holds what the
//           most recent epoch with more than k nodes left there.  A clean epoch's final array is the
//           pre-order trie of its sorted members and is built on demand, slot by slot (trie_node); a
//           replayed epoch's final array was stored when it was replayed.
// Slots no epoch ever wrote are heap bytes in the reference: they read as 0 here and are counted
// (`undefined`): the reference's own output is then not a function of its input.
//
// Execution model: one wavefront per chain (gene, or (cell, gene) in unsorted mode).  The tree walk is
// scalar work that every lane executes identically on LDS-resident arrays; shifts, sorts, scans and trie
// construction are lane-parallel.  The template parameter W supplies lane id / lane count / barrier, so
// that tests/cxx/rl_sim_check.cpp can run the very same code on the CPU (1 lane) against the oracle.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FQG_HD __host__ __device__ __forceinline__
#define FQG_HD_COLD __host__ __device__ __attribute__((noinline))
// (everything is inlined on the device: a version that kept the large pieces - load_run, extend, subtree_nodes,
// open_node - as functions put the simulator's state in scratch memory and ran 1.3 - 1.9 times slower)
#else
#define FQG_HD inline
#define FQG_HD_COLD inline
#endif

namespace fqg {
namespace rl {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMaxUmi = 1048576u;  // UMIS_FEATURE, src/bam_umi_count.c:48
enum : uint32_t { kQOut = 0, kQPart = 2, kQAll = 3 };

// ---- detection -----------------------------------------------------------------------------------
// state: bits 0..3 occupied leaves of the 64-block of the largest member so far, bits 4.. that block + 1
// (0 = no member yet).  Feed the NEW members of an epoch in arrival order; returns true at the first
// insert that overwrites the array's last node.
FQG_HD bool rl_detect_step(uint32_t& state, uint32_t umi_id) {
  const uint32_t v = umi_id - 1u, blk = (v >> 6) + 1u, leaf = (v >> 4) & 3u;
  const uint32_t cur = state >> 4;
  if (blk > cur) {
    state = (blk << 4) | (1u << leaf);
    return false;
  }
  if (blk < cur) return false;
  const uint32_t occ = state & 15u;
  if (occ & (1u << leaf)) return false;
  const uint32_t above = occ >> (leaf + 1u);
  if (above && !(above & (above - 1u))) return true;  // exactly one occupied leaf above: it is the last node
  state |= 1u << leaf;
  return false;
}

// ---- the pre-order trie of a sorted set ------------------------------------------------------------
// v[0..t): sorted distinct (umi id - 1), 20 bits = 8 two-bit digits (depths 0..7) + 4 leaf bits.
// shared_digits(a, b): leading digits two members have in common (8 = same leaf)
FQG_HD uint32_t shared_digits(uint32_t a, uint32_t b) {
  const uint32_t x = (a ^ b) >> 4;  // 16 bits
  if (!x) return 8;
  uint32_t hi = 15;
  while (!((x >> hi) & 1u)) --hi;   // highest differing bit
  return (15u - hi) >> 1;
}
template <class P32>
FQG_HD uint32_t lower_bound_u32(P32 a, uint32_t n, uint32_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}
// base[i] = array index of the first node member i brings (1 + nodes of earlier members), base[t] = size.
// Node at array index p (1 <= p < base[t]):
template <class P32>
FQG_HD uint16_t trie_node(P32 v, P32 base, uint32_t t, uint32_t p) {
  // the member that brings node p: the largest i with base[i] <= p
  uint32_t lo = 0, hi = t;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (base[mid] <= p) lo = mid;
    else hi = mid;
  }
  const uint32_t i = lo;
  const uint32_t sh = i ? shared_digits(v[i - 1], v[i]) : 0u;
  const uint32_t d = sh + 1u + (p - base[i]);  // depth 1..8
  if (d >= 8) {
    uint32_t bits = 0;
    for (uint32_t j = i; j < t && (v[j] >> 4) == (v[i] >> 4); ++j) bits |= 1u << (v[j] & 15u);
    return (uint16_t)bits;
  }
  const uint32_t sft = 20u - 2u * d;  // bits below the node's d digits
  const uint32_t prefix = v[i] >> sft;
  const uint32_t end = lower_bound_u32(v, t, (prefix + 1u) << sft);  // one past the last member below this node
  uint32_t quads = 0;
  for (uint32_t q = 0; q < 4; ++q) {
    const uint32_t key = ((prefix << 2) | q) << (sft - 2u);
    const uint32_t at = lower_bound_u32(v, t, key);
    if (at < t && (v[at] >> (sft - 2u)) == ((prefix << 2) | q)) quads |= kQPart << (2u * q);
  }
  uint32_t cnt = (9u - d) + (base[end] - base[i + 1]);
  if (cnt > 254u) cnt = 255u;
  return (uint16_t)(quads | (cnt << 8));
}

// ---- what a chain kernel reads ----------------------------------------------------------------------
struct ChainView {
  const uint32_t* chain_runs;            // run ids ordered by (chain, cell); null: the run has no history (unsorted mode)
  const unsigned long long* chain_key;   // per position of chain_runs: chain id << 32 | cell
  uint32_t pos;                          // position of the run to replay in chain_runs
  uint32_t run;                          // the run to replay
  const uint32_t* run_start;    // per run: first position in order[]
  const uint32_t* run_len;
  const uint32_t* order;        // record indices grouped by run, record order inside a run
  const uint32_t* umi_id;       // per record
  const uint8_t* set_new;       // per record: first record of its (cell, gene, UMI) - exact for clean epochs
  const uint32_t* run_flag;     // per run: index into the flagged list, or kNone
  const uint32_t* flag_k0;      // per flagged run: records before the one whose insert overwrites (set semantics hold there)
  const uint32_t* flag_off;     // per flagged run: where its final array is stored in `arena`
  uint32_t* flag_ext;           // per flagged run: nodes stored; 0 until it has been replayed (then >= 1)
  uint16_t* arena;
  // by_cell = 1: a run is a (cell, feature) pair of CR-sorted input (fqg_umi_cell_kernels.hip): its records are those
  // of the cell's record range [run_start, run_start + run_len) whose feature is run_feat (order[] is not used),
  // its sorted distinct members are stored, and flag_k0 is a RECORD INDEX (records below it precede the overwrite)
  int by_cell;
  const uint32_t* rec_feat;     // per record
  const uint32_t* run_feat;     // per run
  const uint32_t* run_mem;      // per run: first of its members in members[]
  const uint32_t* run_nmem;
  const uint32_t* members;      // (UMI id - 1), sorted inside a run
};

struct Stats {
  uint32_t undefined, overwrites, wild_writes, overflow, changed, lookback_runs;
  unsigned long long clk_replay, clk_lookback, clk_store;  // device clock ticks spent per phase (0 on the CPU)
  unsigned long long clk_max_wait;                         // most look-back ticks (incl. waiting for an earlier replay) of one run
  unsigned long long clk_max;                              // the slowest run: loop ticks << 32 | records used << 16 | inserts
  unsigned long long clk_build, clk_loop;                  // inside clk_replay: building the array up to the first overwrite / the inserts after it
};

// LDS (or, on the CPU, heap) arrays of one chain worker.  W::p16 / W::p32 are the pointer types of the memory they
// live in: on the device LDS pointers (ds_read / ds_write instead of flat accesses) or plain global ones.
// tree_size() of a node whose count byte is saturated walks the node's whole subtree (src/range_list.c:566-593) -
// for the root of a set of a hundred UMIs that is a thousand dependent reads per insert, twice.  The walks are
// memoised: an entry (slot, depth, size) stays valid until something is written inside [slot, slot + size), and
// moves along when an insert shifts the nodes behind its place.
constexpr uint32_t kMemo = 64;
constexpr uint32_t kMemoWords = 3 * kMemo;
constexpr uint32_t kMemoMin = 1024;  // subtrees smaller than this are walked again (keeping the memo valid costs more)

template <class W>
struct WorkT {
  typename W::p16 node;     // [cap] the array being replayed
  typename W::p32 known;    // [cap / 32] bit: node[] holds the slot's current content
  typename W::p16 stale;    // [cap] history: stale[1..S) materialised
  typename W::p32 mem;      // [mcap] sorted members of the epoch being looked at
  typename W::p32 base;     // [mcap + 1]
  typename W::p32 scratch;  // [lanes + 1] for the scan
  uint32_t cap, mcap;
  typename W::p32 memo;     // [kMemoWords] sizes of saturated subtrees: slot | depth | size (0: free entry)
};

template <class W>
struct History;
// History::get's slow path - one more run of the past, and another, until slot k is covered - as ONE function the device
// code CALLS (round 6).  Inlined, every read of a node (rd: some fifty places) carried its own copy of extend() and
// load_run() - the sort, the trie, the wait for an earlier replay: ten copies in the instruction stream of a kernel that
// one wavefront per CU executes, several times the instruction cache (46 000 instructions; 50 cycles per instruction by the
// counters).  The state travels BY VALUE (the worker's pointers, pos, S): nothing of the simulator has its address
// taken, so it stays in registers - what the attempts of round 3 (functions taking `this`) lost.
template <class W>
FQG_HD_COLD unsigned long long history_fill(const ChainView* cv, WorkT<W> wk, Stats* st, uint32_t pos, uint32_t S, uint32_t k);

template <class W>
struct History {
  const ChainView* cv;
  WorkT<W>* wk;
  Stats* st;
  uint32_t pos;   // runs [0, pos) of the chain have not been looked at yet
  uint32_t S;     // stale[1..S) valid
  bool count;     // count undefined reads (off while a final array is being stored)

  FQG_HD void reset(const ChainView* c, WorkT<W>* w, Stats* s) {
    cv = c; wk = w; st = s; pos = c->chain_runs ? c->pos : 0; S = 1; count = true;
  }
  // is there an earlier run of the same chain?
  FQG_HD bool more() const {
    return pos > 0 && (uint32_t)(cv->chain_key[pos - 1] >> 32) == (uint32_t)(cv->chain_key[cv->pos] >> 32);
  }

  // bitonic sort of mem[0..P), P a power of two
  FQG_HD void sort_members(uint32_t P) {
    typename W::p32 m = wk->mem;
    for (uint32_t k = 2; k <= P; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = W::lane(); i < P; i += W::lanes) {
          const uint32_t l = i ^ j;
          if (l > i) {
            const uint32_t a = m[i], b = m[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { m[i] = b; m[l] = a; }
          }
        }
        W::sync();
      }
  }

  // members of a clean run -> mem[] sorted, base[]; returns the node count of its trie (its final size)
  // (limit: only the records before it are looked at - a count of records, or a record index when by_cell)
  FQG_HD uint32_t load_run(uint32_t run, uint32_t* t_out, uint32_t limit = kNone) {
    const uint32_t s0 = cv->run_start[run], len = cv->run_len[run];
    uint32_t t = 0;
    if (cv->by_cell && limit == kNone) {
      // a clean (cell, feature) pair: its members were stored sorted
      t = cv->run_nmem[run];
      const uint32_t* src = cv->members + cv->run_mem[run];
      for (uint32_t i = W::lane(); i < t && i < wk->mcap; i += W::lanes) wk->mem[i] = src[i];
    } else {
      // compact the new members (uniform scalar loop per 64 records: every lane counts the same)
      const uint32_t feat = cv->by_cell ? cv->run_feat[run] : 0u;
      for (uint32_t k0 = 0; k0 < len; k0 += W::lanes) {
        const uint32_t k = k0 + W::lane();
        uint32_t rec = 0; bool isn = false;
        if (k < len) {
          if (cv->by_cell) {
            rec = s0 + k;
            isn = rec < limit && cv->rec_feat[rec] == feat && cv->set_new[rec] != 0;
          } else {
            rec = cv->order[s0 + k];
            isn = k < limit && cv->set_new[rec] != 0;
          }
        }
        const uint32_t before = W::rank(isn);   // new members in lower lanes
        const uint32_t total = W::count(isn);
        if (isn && t + before < wk->mcap) wk->mem[t + before] = cv->umi_id[rec] - 1u;
        t += total;
      }
    }
    if (t > wk->mcap) { st->overflow = 1; t = wk->mcap; }
    uint32_t P = 1;
    while (P < t) P <<= 1;
    W::sync();
    for (uint32_t i = t + W::lane(); i < P; i += W::lanes) wk->mem[i] = kNone;
    W::sync();
    if (P > 1 && !(cv->by_cell && limit == kNone)) sort_members(P);
    // new nodes per member -> exclusive prefix (+1): every lane sums a contiguous piece
    const uint32_t per = (t + W::lanes - 1) / W::lanes;
    const uint32_t a = W::lane() * per, b = a + per < t ? a + per : t;
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; ++i) sum += i ? 8u - shared_digits(wk->mem[i - 1], wk->mem[i]) : 8u;
    wk->scratch[W::lane()] = sum;
    W::sync();
    uint32_t off = 1;
    for (uint32_t l = 0; l < W::lane(); ++l) off += wk->scratch[l];
    for (uint32_t i = a; i < b; ++i) {
      wk->base[i] = off;
      off += i ? 8u - shared_digits(wk->mem[i - 1], wk->mem[i]) : 8u;
    }
    W::sync();
    uint32_t size = 1;
    for (uint32_t l = 0; l < (uint32_t)W::lanes; ++l) size += wk->scratch[l];
    if (W::lane() == 0) wk->base[t] = size;
    W::sync();
    *t_out = t;
    return size;
  }

  // one more run of the past: extends stale[] if that run left more nodes than anything newer
  FQG_HD void extend() {
    const uint32_t run = cv->chain_runs[--pos];
    st->lookback_runs++;
    const uint32_t fi = cv->run_flag[run];
    if (fi != kNone) {  // a replayed run: its final array is stored once its worker is done (it started before us)
      uint32_t ext = W::wait_nonzero(&cv->flag_ext[fi]);
      if (ext > wk->cap) ext = wk->cap;
      if (ext > S) {
        const uint16_t* src = cv->arena + cv->flag_off[fi];
        for (uint32_t p = S + W::lane(); p < ext; p += W::lanes) wk->stale[p] = src[p];
        W::sync();
        S = ext;
      }
      return;
    }
    if (1u + 8u * (cv->by_cell ? cv->run_nmem[run] : cv->run_len[run]) <= S) return;  // cannot have more than S nodes
    uint32_t t;
    uint32_t size = load_run(run, &t);
    if (size > wk->cap) size = wk->cap;  // only slots the replayed array can have are ever asked for
    if (size > S) {
      for (uint32_t p = S + W::lane(); p < size; p += W::lanes) wk->stale[p] = trie_node(wk->mem, wk->base, t, p);
      W::sync();
      S = size;
    }
  }

  FQG_HD uint16_t get(uint32_t k) {
    if (S <= k && more()) {
      const unsigned long long t0 = W::clock();
      const unsigned long long r = history_fill<W>(cv, *wk, st, pos, S, k);
      pos = (uint32_t)(r >> 32);
      S = (uint32_t)r;
      st->clk_lookback += W::clock() - t0;
    }
    if (k < S) return wk->stale[k];
    if (count) st->undefined++;
    return 0;
  }
};

template <class W>
FQG_HD_COLD unsigned long long history_fill(const ChainView* cv, WorkT<W> wk, Stats* st, uint32_t pos, uint32_t S, uint32_t k) {
  History<W> h;
  h.cv = cv;
  h.wk = &wk;
  h.st = st;
  h.pos = pos;
  h.S = S;
  h.count = true;
  while (h.S <= k && h.more()) h.extend();
  return ((unsigned long long)h.pos << 32) | h.S;
}

template <class W>
struct Sim {
  WorkT<W>* wk;
  History<W>* hist;
  Stats* st;
  uint32_t size, unknown_live, pending, memo_next, memo_live;  // memo_live: valid entries (the same in every lane)

  FQG_HD uint32_t memo_get(uint32_t idx, uint32_t d) {
    if (!memo_live) return 0;  // (the usual case: sets below kMemoMin nodes never put anything)
    for (uint32_t e0 = 0; e0 < kMemo; e0 += (uint32_t)W::lanes) {
      const uint32_t e = e0 + W::lane();
      const bool hit = e < kMemo && wk->memo[2 * kMemo + e] && wk->memo[e] == idx && wk->memo[kMemo + e] == d;
      const uint32_t f = W::find_first(hit);
      if (f != kNone) return wk->memo[2 * kMemo + e0 + f];
    }
    return 0;
  }
  FQG_HD void memo_put(uint32_t idx, uint32_t d, uint32_t t) {
    uint32_t slot = kNone;
    for (uint32_t e0 = 0; e0 < kMemo && slot == kNone; e0 += (uint32_t)W::lanes) {
      const uint32_t e = e0 + W::lane();
      const uint32_t f = W::find_first(e < kMemo && wk->memo[2 * kMemo + e] == 0);
      if (f != kNone) slot = e0 + f;
    }
    if (slot == kNone) slot = memo_next++ % kMemo;
    else ++memo_live;
    if (W::lane() == 0) {
      wk->memo[slot] = idx;
      wk->memo[kMemo + slot] = d;
      wk->memo[2 * kMemo + slot] = t;
    }
    W::sync();
  }
  // a write to slot idx: entries whose subtree holds it are void
  FQG_HD void memo_touch(uint32_t idx) {
    if (!memo_live) return;
    for (uint32_t e0 = 0; e0 < kMemo; e0 += (uint32_t)W::lanes) {
      const uint32_t e = e0 + W::lane();
      const uint32_t t = e < kMemo ? wk->memo[2 * kMemo + e] : 0u;
      const bool gone = t && idx - wk->memo[e] < t;
      if (gone) wk->memo[2 * kMemo + e] = 0;
      memo_live -= W::count(gone);
    }
    W::sync();
  }
  // nodes [at, size) move m slots up
  FQG_HD void memo_shift(uint32_t at, uint32_t m) {
    if (!memo_live) return;
    for (uint32_t e0 = 0; e0 < kMemo; e0 += (uint32_t)W::lanes) {
      const uint32_t e = e0 + W::lane();
      const uint32_t t = e < kMemo ? wk->memo[2 * kMemo + e] : 0u;
      bool gone = false;
      if (t) {
        const uint32_t ci = wk->memo[e];
        if (ci >= at) {
          if (ci + t > size) gone = true;  // (it read slots behind the last node: they do not move)
          else wk->memo[e] = ci + m;
        } else if (ci + t > at) gone = true;
        if (gone) wk->memo[2 * kMemo + e] = 0;
      }
      memo_live -= W::count(gone);
    }
    W::sync();
  }

  // (what the tree walk reads is the same in every lane: telling the compiler - W::uni = readfirstlane - moves the
  // walk's arithmetic and its branches from the vector unit, masks and all, to the scalar one)
  FQG_HD bool known(uint32_t idx) const { return (W::uni(wk->known[idx >> 5]) >> (idx & 31u)) & 1u; }
  FQG_HD void mark(uint32_t idx) { wk->known[idx >> 5] |= 1u << (idx & 31u); }

  FQG_HD void begin() {
    for (uint32_t i = W::lane(); i < wk->cap / 32u; i += W::lanes) wk->known[i] = 0;
    for (uint32_t e = W::lane(); e < kMemo; e += (uint32_t)W::lanes) wk->memo[2 * kMemo + e] = 0;
    memo_next = 0;
    memo_live = 0;
    W::sync();
    wk->node[0] = 1u << 8;  // rl_all(OUT): quadrants empty (the root's count is never looked at by anyone else)
    mark(0);
    size = 1;
    unknown_live = 0;
    pending = 0;
  }
  FQG_HD uint16_t rd(uint32_t idx) {
    if (!unknown_live && idx < size && idx < wk->cap) return (uint16_t)W::uni(wk->node[idx]);  // the usual case: a live, written slot
    if (idx >= wk->cap) { st->overflow = 1; return 0; }
    if (!known(idx)) {
      wk->node[idx] = hist->get(idx);
      mark(idx);
      if (idx < size) --unknown_live;
    }
    return (uint16_t)W::uni(wk->node[idx]);
  }
  FQG_HD void wr(uint32_t idx, uint16_t v) {
    if (idx >= wk->cap) { st->overflow = 1; return; }
    if (!unknown_live && idx < size) {  // a live slot while every live slot is known: no look at the bitmap
      wk->node[idx] = v;
      memo_touch(idx);
      return;
    }
    if (!known(idx)) {
      mark(idx);
      if (idx < size) --unknown_live;
    }
    wk->node[idx] = v;
    memo_touch(idx);
  }
  FQG_HD uint32_t quad(uint32_t idx, uint32_t q) { return (rd(idx) >> (2u * (q - 1u))) & 3u; }
  FQG_HD uint32_t count(uint32_t idx) { return rd(idx) >> 8; }
  static FQG_HD uint32_t quad_of(uint16_t nv, uint32_t q) { return (nv >> (2u * (q - 1u))) & 3u; }

  // tree_size (src/range_list.c:566-593) of the node at idx taken as a node of depth d
  FQG_HD uint32_t subtree_nodes(uint32_t idx, uint32_t d) {
    if (d >= 8) return 1;
    const uint16_t root = rd(idx);
    const uint32_t c0 = root >> 8;
    if (c0 != 255u) return c0;
    {
      const uint32_t m = memo_get(idx, d);
      if (m) return m;
    }
    // Every node of the walk is READ ONCE (round 6): its value stays in s_nv while its four quadrants are looked at -
    // nothing writes to the node array during the walk, and a read is an LDS round trip plus a readfirstlane, which is
    // what this walk is made of (a node was read five times: four quadrants and its count, 13 us per insert on the
    // longest chain of BASELINE configs[3]).
    uint32_t s_idx[9], s_c[9], s_q[9];
    uint16_t s_nv[9];
    int sp = 0;
    s_idx[0] = idx; s_c[0] = 1; s_q[0] = 1; s_nv[0] = root;
    for (;;) {
      if (s_q[sp] > 4) {
        const uint32_t r = s_c[sp];
        if (r >= kMemoMin) memo_put(s_idx[sp], d + (uint32_t)sp, r);
        if (!sp) return r;
        --sp;
        s_c[sp] += r;
        ++s_q[sp];
        continue;
      }
      if (quad_of(s_nv[sp], s_q[sp]) == kQPart) {
        const uint32_t child = s_idx[sp] + s_c[sp], cd = d + (uint32_t)sp + 1u;
        uint32_t cc = 1;
        if (cd < 8) {
          const uint16_t cv = rd(child);
          cc = cv >> 8;
          if (cc == 255u && sp < 8) {
            const uint32_t m = memo_get(child, cd);
            if (m) cc = m;
            else {
              ++sp;
              s_idx[sp] = child; s_c[sp] = 1; s_q[sp] = 1; s_nv[sp] = cv;
              continue;
            }
          }
        }
        s_c[sp] += cc;
      }
      ++s_q[sp];
    }
  }
  // get_location (src/range_list.c:375-408): node at depth d holding value nv, quadrant q (1..4)
  FQG_HD uint32_t child_offset(uint32_t idx, uint16_t nv, uint32_t q, uint32_t d) {
    if (q == 1) return 1;
    uint32_t c = 1;
    if (d == 7) {
      for (uint32_t i = 1; i < q; ++i) c += quad_of(nv, i) == kQPart;
      return c;
    }
    uint32_t at = idx + 1;
    for (uint32_t i = 1; i < q; ++i)
      if (quad_of(nv, i) == kQPart) {
        const uint32_t s = subtree_nodes(at, d + 1);
        at += s;
        c += s;
      }
    return c;
  }
  // slots [lo, hi] must hold their content before lanes move them
  FQG_HD void ensure_known(uint32_t lo, uint32_t hi) {
    // (a slot of the range may stay unknown for as long as one before it does: the bitmap is looked at a word per
    // lane, not a slot per step)
    for (uint32_t w0 = lo >> 5; w0 <= (hi >> 5) && unknown_live; w0 += (uint32_t)W::lanes) {
      const uint32_t w = w0 + W::lane();
      uint32_t missing = 0;
      if (w <= (hi >> 5)) {
        missing = ~wk->known[w];
        if (w == (lo >> 5)) missing &= ~0u << (lo & 31u);
        if (w == (hi >> 5) && (hi & 31u) != 31u) missing &= (1u << ((hi & 31u) + 1u)) - 1u;
      }
      for (;;) {
        const uint32_t f = W::find_first(missing != 0);
        if (f == kNone) break;
        uint32_t bits = W::bcast(missing, f);
        while (bits) {
          const uint32_t b = (uint32_t)__builtin_ctz(bits);
          bits &= bits - 1u;
          (void)rd((w0 + f) * 32u + b);
        }
        if (W::lane() == f) missing = 0;
      }
    }
  }
  // new_node(.., IN) (src/range_list.c:325-372).  A node that has just been created has no children, so an insert
  // that creates a node at depth d + 1 creates every node below it too, m = 8 - d in all, at consecutive slots
  // at, at + 1, .. - each of the reference's m calls sees the same number of nodes behind its slot.  The m
  // one-slot shifts (or the m non-shifts of the defect) are therefore done as ONE move by m slots.
  FQG_HD uint32_t open_node(uint32_t father, uint16_t fv, uint32_t q, uint32_t d) {
    const uint32_t at = father + child_offset(father, fv, q, d);
    if (pending) --pending;  // the gap was opened by the first new node of this insert
    else {
      const uint32_t m = 8u - d;
      const long behind = (long)size - 1 - (long)at;
      if (behind > 0) {
        if (size - 1u + m >= wk->cap) { st->overflow = 1; }
        else {
          if (unknown_live) ensure_known(at, size - 1u);
          memo_shift(at, m);
          // move node[at .. size - 1] m slots up, top piece first, 8 nodes per lane and piece
          uint32_t hi = size - 1u, remaining = (uint32_t)behind + 1u;
          while (remaining) {
            const uint32_t n = remaining < 8u * W::lanes ? remaining : 8u * (uint32_t)W::lanes;
            const uint32_t lo = hi + 1u - n;
            uint16_t v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
              const uint32_t i = lo + j * W::lanes + W::lane();
              v[j] = i <= hi ? wk->node[i] : (uint16_t)0;
            }
            W::sync();
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
              const uint32_t i = lo + j * W::lanes + W::lane();
              if (i <= hi) wk->node[i + m] = v[j];
            }
            W::sync();
            hi = lo - 1u;
            remaining -= n;
          }
          // slots size .. size + m - 1 now hold moved nodes (not live yet: no unknown_live bookkeeping): at most two
          // words of the bitmap
          {
            const uint32_t lo = size, hi = size + m - 1u;
            for (uint32_t wq = lo >> 5; wq <= (hi >> 5); ++wq) {
              uint32_t mask = ~0u;
              if (wq == (lo >> 5)) mask &= ~0u << (lo & 31u);
              if (wq == (hi >> 5) && (hi & 31u) != 31u) mask &= (1u << ((hi & 31u) + 1u)) - 1u;
              wk->known[wq] |= mask;
            }
          }
        }
      } else if (behind == 0) {
        st->overwrites += m;
      } else if (at > size) {
        st->wild_writes += m;
      }
      pending = m - 1u;
    }
    // set_quadrant(father, q, PART)
    wr(father, (uint16_t)((fv & ~(3u << (2u * (q - 1u)))) | (kQPart << (2u * (q - 1u)))));
    wr(at, d + 1u >= 8u ? (uint16_t)0 : (uint16_t)(1u << 8));
    ++size;
    if (size - 1u < wk->cap && !known(size - 1u)) ++unknown_live;
    return at;
  }
  // in_rl (src/range_list.c:203-207 -> in_tree :664-690)
  FQG_HD bool member(uint32_t umi_id) {
    const uint32_t v = umi_id - 1u;
    uint32_t idx = 0;
    for (uint32_t d = 0; d < 8; ++d) {
      const uint32_t q = ((v >> (18u - 2u * d)) & 3u) + 1u;
      const uint16_t nv = rd(idx);
      const uint32_t s = quad_of(nv, q);
      if (s == kQAll) return true;
      if (s != kQPart) return false;
      idx += child_offset(idx, nv, q, d);
    }
    return (rd(idx) >> (v & 15u)) & 1u;
  }
  // set_in_rl(.., IN) (src/range_list.c:169-183 -> set_in :417-496)
  FQG_HD void insert(uint32_t umi_id) {
    const uint32_t v = umi_id - 1u;
    uint32_t path[8], before[8];
    uint32_t idx = 0, d = 0;
    bool reached_leaf = true;
    for (; d < 8; ++d) {
      path[d] = idx;
      before[d] = size;
      const uint32_t q = ((v >> (18u - 2u * d)) & 3u) + 1u;
      const uint16_t nv = rd(idx);
      const uint32_t s = quad_of(nv, q);
      if (s == kQOut) idx = open_node(idx, nv, q, d);
      else if (s == kQAll) { reached_leaf = false; break; }  // returns 0 at this level: no count refresh here
      else idx += child_offset(idx, nv, q, d);
    }
    if (reached_leaf) wr(idx, (uint16_t)(rd(idx) | (1u << (v & 15u))));
    for (int k = (int)d - 1; k >= 0; --k) {
      const uint32_t node = path[k];
      const uint32_t added = size - before[k];
      const uint16_t nv = rd(node);
      const uint32_t c0 = nv >> 8;
      uint32_t c = c0 == 255u ? subtree_nodes(node, (uint32_t)k + 1u) : added + c0;  // (:485: the child's width)
      if (c > 254u) c = 255u;
      wr(node, (uint16_t)((nv & 0xFFu) | (c << 8)));
    }
  }
  // in_rl and, when it says no, set_in_rl(.., IN) in ONE walk: insert() goes down the path member() has just taken and
  // reads the same values (nothing is written in between), so the second descent is skipped.  True: was a member.
  FQG_HD bool member_or_insert(uint32_t umi_id) {
    const uint32_t v = umi_id - 1u;
    uint32_t path[8], before[8];
    uint16_t pnv[8];  // what the path's nodes hold (round 6: the refresh below does not read them again - a read is an
                      // LDS round trip, and the only writes to them in between are open_node's, mirrored here)
    uint32_t idx = 0, d = 0;
    bool opened = false;
    for (; d < 8; ++d) {
      path[d] = idx;
      before[d] = size;
      const uint32_t q = ((v >> (18u - 2u * d)) & 3u) + 1u;
      const uint16_t nv = rd(idx);
      pnv[d] = nv;
      const uint32_t s = quad_of(nv, q);
      if (s == kQOut) {
        idx = open_node(idx, nv, q, d);
        pnv[d] = (uint16_t)((nv & ~(3u << (2u * (q - 1u)))) | (kQPart << (2u * (q - 1u))));  // set_quadrant(father, q, PART)
        opened = true;
      } else if (s == kQAll) return true;  // (never on a freshly opened path)
      else idx += child_offset(idx, nv, q, d);
    }
    const uint16_t leaf = rd(idx);
    if (!opened && ((leaf >> (v & 15u)) & 1u)) return true;
    wr(idx, (uint16_t)(leaf | (1u << (v & 15u))));
    for (int k = 7; k >= 0; --k) {
      const uint32_t node = path[k];
      const uint32_t added = size - before[k];
      const uint16_t nv = path[k] < wk->cap ? pnv[k] : (uint16_t)rd(node);  // (beyond the array: rd flags the overflow)
      const uint32_t c0 = nv >> 8;
      uint32_t c = c0 == 255u ? subtree_nodes(node, (uint32_t)k + 1u) : added + c0;  // (:485: the child's width)
      if (c > 254u) c = 255u;
      wr(node, (uint16_t)((nv & 0xFFu) | (c << 8)));
    }
    return false;
  }
};

// (Sim::member_or_insert is defined with the struct; see below)
// Replays one flagged run.  new_out[rec] (the storage cv.set_new points to, through a non-const pointer) receives
// the reference's decision for every record of the run whose decision differs from set semantics;
// on_change(record, is_new, run) is called (lane 0) for each of them.
template <class W, class OnChange>
FQG_HD void replay_run(const ChainView& cv, WorkT<W>& wk, Stats& st, uint8_t* new_out, OnChange on_change) {
  History<W> hist;
  Sim<W> sim;
  sim.wk = &wk;
  sim.hist = &hist;
  sim.st = &st;
  const uint32_t run = cv.run, fi = cv.run_flag[run];
  const unsigned long long t0 = W::clock();
  const unsigned long long lb0 = st.clk_lookback;
  hist.reset(&cv, &wk, &st);
  sim.begin();
  const uint32_t s0 = cv.run_start[run], len = cv.run_len[run];
  const uint32_t k0 = cv.flag_k0[fi];
  uint32_t k_first = 0;
  if (k0) {
    // up to the first overwrite the array is the pre-order trie of the members so far: build it, do not replay it
    uint32_t t;
    uint32_t size = hist.load_run(run, &t, k0);
    if (size > wk.cap) st.overflow = 1;
    else {
      for (uint32_t p = 1 + W::lane(); p < size; p += W::lanes) wk.node[p] = trie_node(wk.mem, wk.base, t, p);
      uint32_t quads = 0;
      for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t at = lower_bound_u32(wk.mem, t, q << 18);
        if (at < t && (wk.mem[at] >> 18) == q) quads |= kQPart << (2u * q);
      }
      wk.node[0] = (uint16_t)(quads | (1u << 8));
      for (uint32_t w = W::lane(); w * 32u < size; w += W::lanes)
        wk.known[w] = w * 32u + 32u <= size ? 0xFFFFFFFFu : (1u << (size - w * 32u)) - 1u;
      W::sync();
      sim.size = size;
      k_first = k0;
    }
  }
  const unsigned long long t_built = W::clock();
  uint32_t dbg_used = 0, dbg_ins = 0;
  const uint32_t my_feat = cv.by_cell ? cv.run_feat[run] : 0u;
  // The records of the run, a wavefront's worth at a time: every lane fetches the ids of one record, then the replay
  // takes them lane by lane - one round trip to memory per 64 records instead of three per record.
  for (uint32_t kb = cv.by_cell ? 0u : k_first; kb < len; kb += (uint32_t)W::lanes) {
    const uint32_t kk = kb + W::lane();
    uint32_t m_rec = 0, m_u = 0, m_use = 0, m_set = 0;
    if (kk < len) {
      m_rec = cv.by_cell ? s0 + kk : cv.order[s0 + kk];
      m_use = !(cv.by_cell && (cv.rec_feat[m_rec] != my_feat || m_rec < k_first));
      if (m_use) {
        m_u = cv.umi_id[m_rec];
        m_set = cv.set_new[m_rec];
      }
    }
    const uint32_t nb = len - kb < (uint32_t)W::lanes ? len - kb : (uint32_t)W::lanes;
    for (uint32_t j = 0; j < nb; ++j) {
      if (!W::bcast(m_use, j)) continue;
      const uint32_t rec = W::bcast(m_rec, j), u = W::bcast(m_u, j);
      const bool in = sim.member_or_insert(u);
      ++dbg_used;
      dbg_ins += in ? 0u : 1u;
      const uint8_t nw = in ? 0 : 1;
      if (nw != (uint8_t)W::bcast(m_set, j)) {
        st.changed++;
        if (W::lane() == 0) {
          new_out[rec] = nw;
          on_change(rec, nw, run);
        }
      }
    }
  }
  const unsigned long long t1 = W::clock();
  // keep the final array for the later epochs of this chain
  uint32_t ext = sim.size < wk.cap ? sim.size : wk.cap;
  hist.count = false;
  sim.ensure_known(0, ext - 1u);
  hist.count = true;
  uint16_t* dst = cv.arena + cv.flag_off[fi];
  W::sync();
  for (uint32_t p = W::lane(); p < ext; p += W::lanes) dst[p] = wk.node[p];
  W::fence();
  W::sync();
  if (W::lane() == 0) W::publish(&cv.flag_ext[fi], ext);
  W::sync();
  const unsigned long long t2 = W::clock();
  st.clk_replay += t1 - t0;
  st.clk_store += t2 - t1;
  if (st.clk_lookback - lb0 > st.clk_max_wait) st.clk_max_wait = st.clk_lookback - lb0;
  st.clk_build += t_built - t0;
  st.clk_loop += t1 - t_built;
  {
    const unsigned long long packed = ((t1 - t_built) << 32) | ((unsigned long long)(dbg_used & 0xFFFFu) << 16) | (dbg_ins & 0xFFFFu);
    if (packed > st.clk_max) st.clk_max = packed;
  }
}

}  // namespace rl
}  // namespace fqg
