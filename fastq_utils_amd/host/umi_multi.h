// umi_multi.h - bam_umi_count over the GPUs of ONE process (bin/bam_umi_count with FQGPU_DEVICES=0,1,..): the
// alignment loop of a CR-sorted file (reference src/bam_umi_count.c:942-1060) sharded by CELL RANGE, SURVEY 8e.
// The protocol is the one fastq_utils_amd/dist.py runs between processes (umi_count_sharded) - here the "all-gather" is a
// loop over the contexts of this process:
//   shards   the alignment records cut into one contiguous range per context, every cut where the CR value changes (a
//            cell never spans a cut; a file that is not grouped by cell shows up as a cell on both sides of one: the
//            merge reports it and the caller runs the file on one device, whose finding is the reference's)
//   round 1  every context counts its shard (fqg_umi_count, defer_output): the file's numbering of features, cells and
//            UMIs is their first appearance over the shards in order (label_str2id / blabel2id, :143-260)
//   round 2  every context counts again with the file's UMI numbers (the reference's RL_Tree holds those NUMBERS) and
//            says which features had a set replayed as the tree behaves - a tree carries state from cell to cell
//   round 3  every context puts the alignments of those features from all EARLIER shards in front of its own and counts
//            a last time; with fractional increments (NH > 1, several genes) db->tot_reads_obs / tot_umi_obs are one
//            float32 chain over the file (:490-507): that count runs shard after shard, each from the totals before
//   output   fqg_umi_emit with the file's feature ids and cell offsets; the lines of the shards in order are the file's
// One thread per context inside a round (a context is not thread-safe, different contexts are independent).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fqg.h"

namespace fqhost {

struct UmiMultiResult {
  bool ok = false;            // false: take the one-device path (a finding, a limit, a library error - `why` says)
  std::string why;
  std::vector<std::string> features;   // the file's, in id order
  std::vector<uint64_t> cells;
  std::vector<fqg_umi_entry> entries[2];
  uint64_t total[2] = {0, 0};
  uint64_t n_alignments = 0, n_tags_found = 0, n_umis_discarded = 0, n_cells_discarded = 0;
  float tot_reads = 0, tot_umi = 0;
  uint64_t rl_replayed = 0, rl_changed = 0, rl_undefined = 0;
};

namespace umi_multi_detail {

// the value of a two-character 'Z' tag in the aux fields of the alignment record at rec (its block_size word), or ""
inline std::string aux_z(const uint8_t* rec, const uint8_t* end, const char tag[2]) {
  if (rec + 36 > end) return "";
  int32_t block;
  memcpy(&block, rec, 4);
  const uint8_t* stop = rec + 4 + block;
  if (block < 32 || stop > end) return "";
  const uint32_t l_read_name = rec[4 + 8];
  uint16_t n_cigar;
  int32_t l_seq;
  memcpy(&n_cigar, rec + 4 + 12, 2);
  memcpy(&l_seq, rec + 4 + 16, 4);
  const uint8_t* p = rec + 4 + 32 + l_read_name + 4u * n_cigar + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
  while (p + 3 <= stop) {
    const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
    p += 3;
    size_t len = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'Z': case 'H': {
        const uint8_t* z = p;
        while (z < stop && *z) ++z;
        if (t0 == tag[0] && t1 == tag[1] && ty == 'Z') return std::string((const char*)p, (size_t)(z - p));
        len = (size_t)(z - p) + 1;
        break;
      }
      case 'B': {
        if (p + 5 > stop) return "";
        const char sub = (char)p[0];
        int32_t cnt;
        memcpy(&cnt, p + 1, 4);
        const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        len = 5 + w * (size_t)(cnt < 0 ? 0 : cnt);
        break;
      }
      default: return "";  // (an unknown type: nothing behind it can be located)
    }
    p += len;
  }
  return "";
}

// does the alignment loop get as far as the cell with this record?  (src/bam_umi_count.c:946-960: mapped, a feature tag,
// a UMI tag - the filters that come before the cell is looked at; whitelists come after and only drop cells)
inline bool reaches_the_cell(const uint8_t* rec, const uint8_t* end, const fqg_umi_params& prm) {
  if (rec + 36 > end) return false;
  int32_t tid;
  uint16_t flag;
  memcpy(&tid, rec + 4, 4);
  memcpy(&flag, rec + 4 + 14, 2);
  if (tid < 0 || (flag & 4u)) return false;
  return !aux_z(rec, end, prm.feat_tag).empty() && !aux_z(rec, end, prm.umi_tag).empty();
}

struct Shard {
  fqg_ctx* ctx = nullptr;
  uint64_t lo = 0, hi = 0;               // records [lo, hi) of the file
  std::vector<uint8_t> buf;              // header + (history +) the shard's records
  std::vector<uint64_t> offs;
  fqg_umi_result res{};
  std::vector<std::string> features;     // of the last count, id order
  std::vector<uint64_t> cells, umis;
  std::string err;
};

inline bool index_records(Shard& s) {
  uint64_t n = 0, used = 0;
  if (fqg_bam_index_records(s.buf.data(), s.buf.size(), nullptr, 0, &n, &used) != 0) return false;
  s.offs.assign(n ? n : 1, 0);
  return fqg_bam_index_records(s.buf.data(), s.buf.size(), s.offs.data(), n, &n, &used) == 0 && (s.offs.resize(n), true);
}

inline bool count(Shard& s, const fqg_umi_params& prm, bool want_lists) {
  if (!index_records(s)) {
    s.err = "a shard is not a BAM stream";
    return false;
  }
  if (fqg_umi_count(s.ctx, s.buf.data(), s.buf.size(), FQG_MEM_HOST, s.offs.data(), s.offs.size(), &prm, &s.res) != 0) {
    s.err = fqg_last_error(s.ctx);
    return false;
  }
  s.features.clear();
  s.cells.clear();
  if (s.res.code != FQG_OK || !want_lists) return true;
  std::vector<char> names(s.res.n_features * 25 + 25);
  s.cells.assign(s.res.n_cells, 0);
  if (fqg_umi_features(s.ctx, names.data(), s.res.n_features) != 0 || fqg_umi_cells(s.ctx, s.cells.data(), s.res.n_cells) != 0) {
    s.err = fqg_last_error(s.ctx);
    return false;
  }
  for (uint64_t i = 0; i < s.res.n_features; ++i) s.features.emplace_back(&names[i * 25]);
  return true;
}

template <class F>
inline void per_shard(std::vector<Shard>& S, F f) {
  std::vector<std::thread> th;
  for (size_t i = 1; i < S.size(); ++i) th.emplace_back([&, i] { f(S[i]); });
  f(S[0]);
  for (auto& t : th) t.join();
}

}  // namespace umi_multi_detail

// stream: the inflated BAM file; offsets: of its n_rec alignment records; ctxs: one context per device (>= 2).
// prm: the parameters of the one-device call (sorted_by_cell must be set).
inline UmiMultiResult umi_count_multi(const std::vector<fqg_ctx*>& ctxs, const std::vector<uint8_t>& stream,
                                      const std::vector<uint64_t>& offsets, uint64_t n_rec, uint64_t used, const fqg_umi_params& prm0) {
  using namespace umi_multi_detail;
  UmiMultiResult R;
  auto give_up = [&](const std::string& why) {
    R.ok = false;
    R.why = why;
    return R;
  };
  if (ctxs.size() < 2 || !prm0.sorted_by_cell || n_rec < ctxs.size()) return give_up("nothing to shard");
  const uint8_t* base = stream.data();
  const uint8_t* end = base + used;
  const size_t hdr_len = (size_t)offsets[0];
  auto rec_end = [&](uint64_t i) { return i + 1 < n_rec ? offsets[i + 1] : used; };
  // ---- the cuts: where the CR value changes, at or behind the even split ----
  std::vector<uint64_t> cut{0};
  for (size_t d = 1; d < ctxs.size(); ++d) {
    // (only records the alignment loop takes as far as the cell say where a cell begins: an unmapped read, a read
    // without a feature or a UMI, carries whatever CR it carries; the value in front of a cut is that of the nearest
    // such record before it)
    uint64_t i = std::max<uint64_t>(n_rec * d / ctxs.size(), cut.back() + 1);
    std::string before;
    bool have_before = false;
    // (a record whose CR this walk does not find - aux fields libbam reads differently than this parser, no tag at all -
    // is no place for a cut and says nothing about the cell in front of one: the merge of the shards' cell lists is
    // what vouches for the cuts, whatever they were made from)
    for (uint64_t b = i; b-- > cut.back() && !have_before;)
      if (reaches_the_cell(base + offsets[b], end, prm0)) {
        before = aux_z(base + offsets[b], end, prm0.cell_tag);
        have_before = !before.empty();
      }
    for (; i < n_rec; ++i) {
      if (!reaches_the_cell(base + offsets[i], end, prm0)) continue;
      const std::string here = aux_z(base + offsets[i], end, prm0.cell_tag);
      if (here.empty()) continue;
      if (have_before && here != before) break;
      before = here;
      have_before = true;
    }
    if (i >= n_rec) break;
    cut.push_back(i);
  }
  cut.push_back(n_rec);
  if (cut.size() < 3) return give_up("one cell range");
  std::vector<Shard> S(cut.size() - 1);
  for (size_t r = 0; r < S.size(); ++r) {
    S[r].ctx = ctxs[r];
    S[r].lo = cut[r];
    S[r].hi = cut[r + 1];
  }
  auto own_bytes = [&](const Shard& s, std::vector<uint8_t>& out) {
    out.insert(out.end(), base + offsets[s.lo], base + rec_end(s.hi - 1));
  };
  auto build = [&](Shard& s, const std::vector<uint8_t>* history) {
    s.buf.assign(base, base + hdr_len);
    if (history) s.buf.insert(s.buf.end(), history->begin(), history->end());
    own_bytes(s, s.buf);
  };
  fqg_umi_params prm = prm0;
  prm.defer_output = 1;
  // ---- round 1 ----
  per_shard(S, [&](Shard& s) {
    build(s, nullptr);
    if (!count(s, prm, true)) return;
    if (s.res.code != FQG_OK) return;
    uint64_t n = 0;
    if (fqg_umi_umis(s.ctx, nullptr, 0, &n) != 0) {
      s.err = fqg_last_error(s.ctx);
      return;
    }
    s.umis.assign(n ? n : 1, 0);
    if (fqg_umi_umis(s.ctx, s.umis.data(), n, &n) != 0) s.err = fqg_last_error(s.ctx);
    s.umis.resize(n);
  });
  for (auto& s : S) {
    if (!s.err.empty()) return give_up(s.err);
    if (s.res.code != FQG_OK) return give_up("a finding in a shard");       // (the one-device run words it as the reference does)
    if (s.res.rl_unresolved) return give_up("a set beyond the replay's limits");
  }
  // the file's features and cells: first appearance over the shards in order
  std::map<std::string, uint32_t> gid;
  std::vector<std::vector<uint32_t>> remap(S.size());
  std::vector<uint64_t> cell_offset(S.size());
  std::set<uint64_t> seen;
  bool unit = true;
  for (size_t r = 0; r < S.size(); ++r) {
    remap[r].push_back(0);
    for (const auto& name : S[r].features) {
      auto it = gid.find(name);
      if (it == gid.end()) {
        R.features.push_back(name);
        it = gid.emplace(name, (uint32_t)R.features.size()).first;
      }
      remap[r].push_back(it->second);
    }
    cell_offset[r] = R.cells.size();
    for (uint64_t c : S[r].cells) {
      if (!seen.insert(c).second) return give_up("a cell in two shards: the file is not grouped by cell");  // (:1004-1007)
      R.cells.push_back(c);
    }
    unit = unit && S[r].res.unit_increments;
    R.n_alignments += S[r].res.n_alignments;
    R.n_tags_found += S[r].res.n_tags_found;
    R.n_umis_discarded += S[r].res.n_umis_discarded;
    R.n_cells_discarded += S[r].res.n_cells_discarded;
  }
  // ---- rounds 2 and 3 (the tree is not a set: strict_set skips them) ----
  std::vector<std::vector<uint8_t>> history(S.size());
  std::vector<uint64_t> tk;
  std::vector<uint32_t> ti;
  fqg_umi_params prm2 = prm;
  if (!prm.strict_set) {
    // UMI numbers of the whole file: 1 + how many distinct UMIs the file saw before (the shards' lists in order)
    std::map<uint64_t, uint32_t> first;
    for (auto& s : S)
      for (uint64_t u : s.umis)
        if (!first.count(u)) {
          const uint32_t id = (uint32_t)first.size() + 1;
          first.emplace(u, id);
        }
    for (auto& kv : first) {  // (a std::map walks its keys in ascending order)
      tk.push_back(kv.first);
      ti.push_back(kv.second);
    }
    prm2.umi_table_keys = tk.empty() ? nullptr : tk.data();
    prm2.umi_table_ids = ti.empty() ? nullptr : ti.data();
    prm2.n_umi_table = tk.size();
    std::vector<std::vector<std::string>> replayed(S.size());
    per_shard(S, [&](Shard& s) {
      if (!count(s, prm2, true) || s.res.code != FQG_OK) return;
      std::vector<uint8_t> flags(s.res.n_features + 2, 0);
      if (fqg_umi_replayed_features(s.ctx, flags.data(), s.res.n_features + 1) != 0) {
        s.err = fqg_last_error(s.ctx);
        return;
      }
      const size_t r = (size_t)(&s - &S[0]);
      for (uint64_t f = 1; f <= s.res.n_features; ++f)
        if (flags[f]) replayed[r].push_back(s.features[f - 1]);
    });
    std::set<std::string> carried;
    for (size_t r = 0; r < S.size(); ++r) {
      if (!S[r].err.empty()) return give_up(S[r].err);
      if (S[r].res.code != FQG_OK || S[r].res.rl_unresolved) return give_up("a finding in a shard (second count)");
      carried.insert(replayed[r].begin(), replayed[r].end());
    }
    if (!carried.empty()) {
      // the alignments of those features, per shard, in file order (kilobytes per feature)
      std::vector<std::vector<uint8_t>> blob(S.size());
      per_shard(S, [&](Shard& s) {
        const size_t r = (size_t)(&s - &S[0]);
        std::set<uint32_t> ids;
        for (uint32_t f = 1; f <= s.features.size(); ++f)
          if (carried.count(s.features[f - 1])) ids.insert(f);
        if (ids.empty()) return;
        std::vector<uint32_t> feat(s.offs.size() + 1, 0);
        if (fqg_umi_record_features(s.ctx, feat.data(), s.offs.size()) != 0) {
          s.err = fqg_last_error(s.ctx);
          return;
        }
        for (size_t i = 0; i < s.offs.size(); ++i)
          if (ids.count(feat[i])) {
            const uint64_t b = s.offs[i], e = i + 1 < s.offs.size() ? s.offs[i + 1] : s.buf.size();
            blob[r].insert(blob[r].end(), s.buf.begin() + (long)b, s.buf.begin() + (long)e);
          }
      });
      for (auto& s : S)
        if (!s.err.empty()) return give_up(s.err);
      for (size_t r = 1; r < S.size(); ++r) {
        history[r] = history[r - 1];
        history[r].insert(history[r].end(), blob[r - 1].begin(), blob[r - 1].end());
      }
    }
  }
  // what the history alone counts (its cells come first in an augmented shard; its counters come off the totals)
  std::vector<fqg_umi_result> hist_res(S.size());
  std::vector<uint64_t> n_hist_cells(S.size(), 0);
  for (auto& h : hist_res) memset(&h, 0, sizeof(h));
  std::vector<bool> augmented(S.size(), false);
  // (the header's third field is the sum of the TRUNCATED counts of the lines, src/bam_umi_count.c:1093-1113, which the
  // lines' rounded values do not give back: fqg_umi_emit adds it up, and the history's share - the same alignments
  // counted alone, first, give the same lines - is taken off)
  std::vector<uint64_t> hist_total0(S.size(), 0), hist_total1(S.size(), 0);
  per_shard(S, [&](Shard& s) {
    const size_t r = (size_t)(&s - &S[0]);
    if (history[r].empty()) return;
    Shard h;
    h.ctx = s.ctx;
    h.buf.assign(base, base + hdr_len);
    h.buf.insert(h.buf.end(), history[r].begin(), history[r].end());
    if (!count(h, prm2, true)) {
      s.err = h.err;
      return;
    }
    hist_res[r] = h.res;
    n_hist_cells[r] = h.res.code == FQG_OK ? h.res.n_cells : 0;
    if (h.res.code == FQG_OK) {
      std::vector<uint32_t> rm{0};
      for (const auto& name : h.features) rm.push_back(gid.count(name) ? gid.at(name) : 0u);
      fqg_umi_result he;
      memset(&he, 0, sizeof(he));
      if (fqg_umi_emit(h.ctx, rm.data(), rm.size(), 0, &he) != 0) {
        s.err = fqg_last_error(h.ctx);
        return;
      }
      hist_total0[r] = he.total[0];
      hist_total1[r] = he.total[1];
    }
    build(s, &history[r]);
    augmented[r] = true;
  });
  for (auto& s : S)
    if (!s.err.empty()) return give_up(s.err);
  float chain_reads = 0, chain_umi = 0;
  if (unit) {
    per_shard(S, [&](Shard& s) {
      const size_t r = (size_t)(&s - &S[0]);
      if (augmented[r]) count(s, prm2, true);
    });
  } else {
    for (size_t r = 0; r < S.size(); ++r) {  // shard after shard: the chain of totals
      fqg_umi_params p = prm2;
      p.db_start_reads = chain_reads;
      p.db_start_umi = chain_umi;
      p.db_skip = augmented[r] ? hist_res[r].n_alignments : 0;
      if (!count(S[r], p, true)) return give_up(S[r].err);
      chain_reads = S[r].res.tot_reads;
      chain_umi = S[r].res.tot_umi;
    }
  }
  for (auto& s : S) {
    if (!s.err.empty()) return give_up(s.err);
    if (s.res.code != FQG_OK || s.res.rl_unresolved) return give_up("a finding in a shard (last count)");
  }
  // ---- output: the file's feature ids by name, the history's cells dropped, cell ids continuing ----
  std::vector<std::vector<fqg_umi_entry>> ent[2];
  ent[0].resize(S.size());
  ent[1].resize(S.size());
  std::vector<uint64_t> n_new(S.size()), n_counted(S.size());
  std::vector<uint64_t> shard_total[2] = {std::vector<uint64_t>(S.size(), 0), std::vector<uint64_t>(S.size(), 0)};
  per_shard(S, [&](Shard& s) {
    const size_t r = (size_t)(&s - &S[0]);
    std::vector<uint32_t> rm{0};
    for (const auto& name : s.features) rm.push_back(gid.at(name));
    fqg_umi_result e;
    memset(&e, 0, sizeof(e));
    if (fqg_umi_emit(s.ctx, rm.data(), rm.size(), (uint32_t)(cell_offset[r] - n_hist_cells[r]), &e) != 0) {
      s.err = fqg_last_error(s.ctx);
      return;
    }
    for (int w = 0; w < 2; ++w) {
      std::vector<fqg_umi_entry> all(e.n_entries[w]);
      if (!all.empty() && fqg_umi_entries(s.ctx, w, all.data(), all.size()) != 0) {
        s.err = fqg_last_error(s.ctx);
        return;
      }
      for (const auto& t : all)
        if (t.col > cell_offset[r]) ent[w][r].push_back(t);
      shard_total[w][r] = e.total[w] - (w ? hist_total1[r] : hist_total0[r]);
    }
    n_new[r] = s.res.n_new - hist_res[r].n_new;
    n_counted[r] = s.res.n_counted - hist_res[r].n_counted;
  });
  uint64_t sum_new = 0, sum_counted = 0;
  for (size_t r = 0; r < S.size(); ++r) {
    if (!S[r].err.empty()) return give_up(S[r].err);
    for (int w = 0; w < 2; ++w) {
      R.total[w] += shard_total[w][r];
      R.entries[w].insert(R.entries[w].end(), ent[w][r].begin(), ent[w][r].end());
    }
    sum_new += n_new[r];
    sum_counted += n_counted[r];
    R.rl_replayed += S[r].res.rl_replayed;
    R.rl_changed += S[r].res.rl_changed;
    R.rl_undefined += S[r].res.rl_undefined - hist_res[r].rl_undefined;
  }
  auto unit_float = [](uint64_t n) { return (float)std::min<uint64_t>(n, 1ull << 24); };  // additions of 1.0f saturate at 2^24
  R.tot_reads = unit ? unit_float(sum_counted) : chain_reads;
  R.tot_umi = unit ? unit_float(sum_new) : chain_umi;
  R.ok = true;
  return R;
}

}  // namespace fqhost
