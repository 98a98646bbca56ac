// fq_names_multi.h - read names across the GPUs of ONE process (the drop-in programs with FQGPU_DEVICES=0,1,..):
// the unique-name test of fastq_index_readnames (reference src/fastq.c:396-439) and the file-2 loop of fastq_info
// (src/fastq_info.c:333-362) when the records of a file are spread over several contexts.  The protocol is
// SURVEY 8e's, the same that fastq_utils_amd/dist.py runs between processes over RCCL - here the "all-to-all" is a
// set of peer copies (fqg_device_copy: xGMI between two GPUs), since all contexts live in this process:
//   1. every context turns the canonical names of its frames into 16-byte (fingerprint, global record index) pairs,
//      bucketed by owner (fqg_names_fingerprints_acct; owner = high bits of the fingerprint);
//   2. bucket (d -> o) is copied into owner o's receive buffer;
//   3. every owner sorts what it received and classifies runs of equal fingerprints on its device
//      (fqg_fpset_candidates / fqg_fpset_pair_runs);
//   4. what a fingerprint cannot decide is decided on the name BYTES, fetched from the context that holds the record
//      (fqg_frame_name) - a hash collision can neither fake nor hide a finding.  In the pairing the names themselves
//      travel beside the pairs (64-byte records, fqg_names_fingerprints_named / fqg_fpset_insert_named): one holder and
//      one asker are a pair only when their names are the same bytes, as the strcmp behind the key match says
//      (src/fastq.c:577-587).
// One thread per context for steps 1 and 3 (a context is not thread-safe, different contexts are independent).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fqg.h"

namespace fqhost {

struct NameShard {  // what one context holds of one file
  fqg_ctx* ctx = nullptr;
  struct Piece {
    const fqg_frame* frame;
    uint64_t first_record, n_records;  // global index of the frame's first record, records in it
  };
  std::vector<Piece> pieces;
};

struct NamesOfFile {
  std::vector<NameShard> shards;  // one per context, in any order
  fqg_file_state st{};
  uint64_t flag = 0;  // FQG_FP_FILE2 for the asking file of a pairing
};

struct PairingOutcome {
  uint64_t matched = 0, leftover = 0, unpaired = 0;
  bool has_first = false;
  uint64_t first_unpaired = 0;  // record index in file 2
  std::string first_name;
};

class NamesExchange {
 public:
  // error text of the last failure ("" when none); every call returns false on failure
  std::string error;

  // smallest global record index whose name already occurred at a smaller index, if any; name_bytes = the sum the
  // reference accounts for the names (src/fastq.c:609)
  bool first_duplicate(const NamesOfFile& f, bool* found, uint64_t* record, std::string* name, uint64_t* name_bytes) {
    *found = false;
    *name_bytes = 0;
    std::vector<const NamesOfFile*> files{&f};
    Exchanged x;
    if (!exchange(files, x)) return false;
    *name_bytes = x.name_bytes;
    // candidates of every owner
    std::vector<std::pair<uint64_t, uint64_t>> cand;
    std::vector<std::string> errs(x.owners.size());
    std::vector<std::vector<uint64_t>> per(x.owners.size());
    run_per_owner(x, [&](size_t o) {
      Owner& ow = x.owners[o];
      if (!ow.n) return;
      fqg_fpset* set = nullptr;
      if (fqg_fpset_create(ow.ctx, ow.n, &set) != 0 || fqg_fpset_insert(ow.ctx, set, ow.recv, ow.n) != 0) {
        errs[o] = fqg_last_error(ow.ctx);
        if (set) fqg_fpset_destroy(set);
        return;
      }
      uint64_t cap = 1 << 16, n_found = 0;
      std::vector<uint64_t> pairs(2 * cap);
      int rc = fqg_fpset_candidates(ow.ctx, set, pairs.data(), cap, &n_found);
      if (rc == 0 && n_found > cap) {
        cap = n_found;
        pairs.resize(2 * cap);
        rc = fqg_fpset_candidates(ow.ctx, set, pairs.data(), cap, &n_found);
      }
      if (rc != 0) errs[o] = fqg_last_error(ow.ctx);
      else per[o].assign(pairs.begin(), pairs.begin() + 2 * n_found);
      fqg_fpset_destroy(set);
    });
    release(x);
    for (auto& e : errs)
      if (!e.empty()) return fail(e);
    for (auto& p : per)
      for (size_t k = 0; k + 1 < p.size(); k += 2) cand.emplace_back(p[k], p[k + 1]);
    if (cand.empty()) return true;
    // all holders of one fingerprint are compared with each other: the smallest index whose name equals the name
    // of an earlier holder is the record the serial loop stops at
    std::map<uint64_t, std::set<uint64_t>> groups;
    for (auto& c : cand) groups[c.first].insert(c.second);
    bool have = false;
    uint64_t best = 0;
    std::string best_name;
    for (auto& g : groups) {
      std::string nm;
      if (!name_of(f, g.first, &nm)) return false;
      std::set<std::string> seen{nm};
      for (uint64_t later : g.second) {
        if (have && later >= best) break;
        if (!name_of(f, later, &nm)) return false;
        if (seen.count(nm)) {
          have = true;
          best = later;
          best_name = nm;
          break;
        }
        seen.insert(nm);
      }
    }
    *found = have;
    *record = best;
    *name = best_name;
    return true;
  }

  // the file-2 loop: every name of f2 (flag FQG_FP_FILE2) looks for its holder in f1
  // Mate files in one order (the usual case): is the name of EVERY record of f2 the name of the record of f1 with the
  // same number?  (fqg_frame_name_records / fqg_frame_names_equal, include/fqg.h: a contiguous copy of name records
  // per piece instead of the exchange by hash of pairing() below.)  *all = true: yes, and the files hold the same
  // number of records - with f1 free of repeated names (first_duplicate has said so) that is every read paired.
  // Anything else - a file that is shorter, one record out of place, a name beyond the 56 bytes a record holds - is
  // *all = false and left to pairing().  One thread drives every context here: the pieces are at rest, the work per
  // piece is two small kernels and one copy.
  bool paired_by_position(const NamesOfFile& f1, const NamesOfFile& f2, bool* all) {
    *all = false;
    struct Held {
      fqg_ctx* ctx;
      NameShard::Piece p;
    };
    std::vector<Held> one, two;
    uint64_t n1 = 0, n2 = 0, longest = 0;
    for (auto& sh : f1.shards)
      for (auto& p : sh.pieces) one.push_back(Held{sh.ctx, p}), n1 += p.n_records, longest = std::max(longest, p.n_records);
    for (auto& sh : f2.shards)
      for (auto& p : sh.pieces) two.push_back(Held{sh.ctx, p}), n2 += p.n_records, longest = std::max(longest, p.n_records);
    if (n1 != n2 || !n1) {
      why = "the files hold " + std::to_string(n1) + " and " + std::to_string(n2) + " records";
      return true;
    }
    auto by_first = [](const Held& a, const Held& b) { return a.p.first_record < b.p.first_record; };
    std::sort(one.begin(), one.end(), by_first);
    std::sort(two.begin(), two.end(), by_first);
    std::map<fqg_ctx*, void*> scratch;  // per context: room for the name records of the longest piece
    auto room = [&](fqg_ctx* c) -> void* {
      auto it = scratch.find(c);
      if (it != scratch.end()) return it->second;
      void* p = fqg_device_alloc(c, longest * FQG_NAME_REC_BYTES);
      scratch[c] = p;
      return p;
    };
    bool ok = true, same = true;
    size_t q = 0;
    for (size_t t = 0; t < two.size() && ok && same; ++t) {
      const Held& b = two[t];
      const uint64_t b_lo = b.p.first_record, b_hi = b_lo + b.p.n_records;
      while (q < one.size() && one[q].p.first_record + one[q].p.n_records <= b_lo) ++q;
      for (size_t k = q; k < one.size() && one[k].p.first_record < b_hi && ok && same; ++k) {
        const Held& a = one[k];
        const uint64_t lo = std::max(a.p.first_record, b_lo), hi = std::min(a.p.first_record + a.p.n_records, b_hi);
        if (lo >= hi) continue;
        void* at_a = room(a.ctx);
        void* at_b = a.ctx == b.ctx ? at_a : room(b.ctx);
        if (!at_a || !at_b) {
          ok = fail("device allocation failed");
          break;
        }
        if (fqg_frame_name_records(a.ctx, a.p.frame, &f1.st, lo - a.p.first_record, hi - lo, at_a) != 0) {
          ok = fail(fqg_last_error(a.ctx));
          break;
        }
        if (a.ctx != b.ctx && fqg_device_copy(b.ctx, at_b, a.ctx, at_a, (hi - lo) * FQG_NAME_REC_BYTES) != 0) {
          ok = fail(fqg_last_error(b.ctx));
          break;
        }
        uint64_t eq = 0, undecided = 0;
        if (fqg_frame_names_equal(b.ctx, b.p.frame, &f2.st, lo - b_lo, hi - lo, at_b, &eq, &undecided) != 0) {
          ok = fail(fqg_last_error(b.ctx));
          break;
        }
        if (eq != hi - lo) {
          same = false;
          why = "records " + std::to_string(lo) + " .. " + std::to_string(hi - 1) + ": " + std::to_string(eq) + " names at their place, " +
                std::to_string(undecided) + " beyond what a name record holds";
        }
      }
    }
    for (auto& kv : scratch)
      if (kv.second) fqg_device_free(kv.first, kv.second);
    if (ok && same) {
      *all = true;
      why = "every name at its place";
    }
    return ok;
  }
  std::string why;  // what paired_by_position found (FQGPU_TIMING prints it)

  static double t_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  static void t_say(const char* what, double since) {
    if (getenv("FQGPU_TIMING")) fprintf(diag(), "fqgpu timing: name exchange: %s %.3f s\n", what, t_now() - since);
  }
  bool pairing(const NamesOfFile& f1, const NamesOfFile& f2, PairingOutcome* out) {
    *out = PairingOutcome();
    std::vector<const NamesOfFile*> files{&f1, &f2};
    Exchanged x;
    double t0 = t_now();
    if (!exchange(files, x, true)) return false;
    t_say("fingerprints + name records made and sent to their owners", t0);
    t0 = t_now();
    struct Part {
      fqg_pair_summary s{};
      std::vector<uint64_t> entries;  // (run, index) pairs
      std::string err;
    };
    std::vector<Part> parts(x.owners.size());
    run_per_owner(x, [&](size_t o) {
      Owner& ow = x.owners[o];
      parts[o].s.first_unpaired = ~0ull;
      if (!ow.n) return;
      fqg_fpset* set = nullptr;
      if (fqg_fpset_create(ow.ctx, ow.n, &set) != 0 || fqg_fpset_insert_named(ow.ctx, set, ow.recv, ow.recv_names, ow.n) != 0) {
        parts[o].err = fqg_last_error(ow.ctx);
        if (set) fqg_fpset_destroy(set);
        return;
      }
      uint64_t cap = 1 << 16;
      parts[o].entries.resize(2 * cap);
      int rc = fqg_fpset_pair_runs(ow.ctx, set, &parts[o].s, parts[o].entries.data(), cap);
      if (rc == 0 && parts[o].s.n_complex > cap) {
        cap = parts[o].s.n_complex;
        parts[o].entries.resize(2 * cap);
        rc = fqg_fpset_pair_runs(ow.ctx, set, &parts[o].s, parts[o].entries.data(), cap);
      }
      if (rc != 0) parts[o].err = fqg_last_error(ow.ctx);
      else parts[o].entries.resize(2 * parts[o].s.n_complex);
      fqg_fpset_destroy(set);
    });
    t_say("owners: set built, runs paired", t0);
    t0 = t_now();
    release(x);
    t_say("buffers freed", t0);
    bool have = false;
    uint64_t first = 0;
    auto offer_first = [&](uint64_t idx) {
      if (!have || idx < first) {
        have = true;
        first = idx;
      }
    };
    for (auto& p : parts) {
      if (!p.err.empty()) return fail(p.err);
      out->matched += p.s.matched;
      out->leftover += p.s.leftover;
      out->unpaired += p.s.unpaired;
      if (p.s.first_unpaired != ~0ull) offer_first(p.s.first_unpaired);
      // runs the device could not classify (a name asked for twice, a collision): per NAME the holder pairs with the
      // smallest asker; later askers, and askers of a name without holder, are unpaired; holders nobody asked for
      // are left over.  Run ids are owner-local.
      std::map<uint64_t, std::vector<uint64_t>> runs;
      for (size_t k = 0; k + 1 < p.entries.size(); k += 2) runs[p.entries[k]].push_back(p.entries[k + 1]);
      for (auto& r : runs) {
        std::map<std::string, std::pair<std::vector<uint64_t>, std::vector<uint64_t>>> by_name;
        for (uint64_t g : r.second) {
          std::string nm;
          const bool asker = (g & FQG_FP_FILE2) != 0;
          if (!name_of(asker ? f2 : f1, g & ~FQG_FP_FILE2, &nm)) return false;
          auto& e = by_name[nm];
          (asker ? e.second : e.first).push_back(g & ~FQG_FP_FILE2);
        }
        for (auto& e : by_name) {
          auto& holders = e.second.first;
          auto& askers = e.second.second;
          std::sort(askers.begin(), askers.end());
          size_t bad_from = 0;
          if (!holders.empty()) {
            if (!askers.empty()) {
              out->matched += 1;
              out->leftover += holders.size() - 1;
              bad_from = 1;
            } else {
              out->leftover += holders.size();
              bad_from = askers.size();
            }
          }
          out->unpaired += askers.size() - bad_from;
          if (bad_from < askers.size()) offer_first(askers[bad_from]);
        }
      }
    }
    if (have) {
      out->has_first = true;
      out->first_unpaired = first;
      if (!name_of(f2, first, &out->first_name)) return false;
    }
    return true;
  }

 private:
  struct Owner {
    fqg_ctx* ctx = nullptr;
    void* recv = nullptr;
    void* recv_names = nullptr;  // (pairing) the name record of every received pair, in the same order
    uint64_t n = 0;
  };
  struct Exchanged {
    std::vector<Owner> owners;
    uint64_t name_bytes = 0;
  };
  bool fail(const std::string& e) {
    error = e;
    return false;
  }
  template <class F>
  static void run_per_owner(Exchanged& x, F fn) {
    std::vector<std::thread> th;
    for (size_t o = 1; o < x.owners.size(); ++o) th.emplace_back(fn, o);
    if (!x.owners.empty()) fn(0);
    for (auto& t : th) t.join();
  }
  static void release(Exchanged& x) {
    for (auto& o : x.owners) {
      if (o.recv) fqg_device_free(o.ctx, o.recv);
      if (o.recv_names) fqg_device_free(o.ctx, o.recv_names);
    }
    x.owners.clear();
  }

  // canonical name of global record `idx` of file f, from the context that holds it
  bool name_of(const NamesOfFile& f, uint64_t idx, std::string* out) {
    for (auto& sh : f.shards)
      for (auto& p : sh.pieces)
        if (idx >= p.first_record && idx < p.first_record + p.n_records) {
          char buf[FQG_MAX_LABEL_LENGTH + 8];
          const int64_t n = fqg_frame_name(sh.ctx, p.frame, &f.st, idx - p.first_record, buf, sizeof buf);
          if (n < 0) return fail(fqg_last_error(sh.ctx));
          out->assign(buf, (size_t)n);
          return true;
        }
    return fail("a record index that no context holds");
  }

  // steps 1 and 2: fingerprints of all files on all contexts, every bucket copied to its owner.  The owners are the
  // contexts of files[0] (every file has a shard on every context, possibly without pieces).
  bool exchange(const std::vector<const NamesOfFile*>& files, Exchanged& x, bool named = false) {
    const size_t D = files[0]->shards.size();
    if (D == 0 || D > FQG_MAX_OWNERS) return fail("between 1 and 64 contexts");
    for (auto* f : files)
      if (f->shards.size() != D) return fail("every file needs a shard per context");
    struct Bucket {
      const void* src;
      uint64_t n;
      const void* nsrc;  // (named) the bucket's name records
    };
    struct Local {
      void* buf = nullptr;
      void* nbuf = nullptr;
      std::vector<std::vector<Bucket>> to;  // per owner
      uint64_t name_bytes = 0;
      std::string err;
    };
    std::vector<Local> loc(D);
    auto produce = [&](size_t d) {
      Local& L = loc[d];
      L.to.resize(D);
      fqg_ctx* ctx = files[0]->shards[d].ctx;
      uint64_t total = 0;
      for (auto* f : files)
        for (auto& p : f->shards[d].pieces) total += p.n_records;
      if (!total) return;
      L.buf = fqg_device_alloc(ctx, total * sizeof(fqg_fp));
      if (named) L.nbuf = fqg_device_alloc(ctx, total * FQG_NAME_REC_BYTES);
      if (!L.buf || (named && !L.nbuf)) {
        L.err = "device allocation failed";
        return;
      }
      uint64_t off = 0;
      for (auto* f : files)
        for (auto& p : f->shards[d].pieces) {
          if (f->shards[d].ctx != ctx) {
            L.err = "the shards of a context must belong to it";
            return;
          }
          uint64_t counts[FQG_MAX_OWNERS], nb = 0;
          if (fqg_names_fingerprints_named(ctx, p.frame, &f->st, p.first_record | f->flag, (uint32_t)D,
                                           (char*)L.buf + off * sizeof(fqg_fp),
                                           named ? (char*)L.nbuf + off * FQG_NAME_REC_BYTES : nullptr, counts, &nb) != 0) {
            L.err = fqg_last_error(ctx);
            return;
          }
          if (!f->flag) L.name_bytes += nb;
          for (size_t o = 0; o < D; ++o) {
            if (counts[o])
              L.to[o].push_back(Bucket{(char*)L.buf + off * sizeof(fqg_fp), counts[o],
                                       named ? (char*)L.nbuf + off * FQG_NAME_REC_BYTES : nullptr});
            off += counts[o];
          }
        }
    };
    const double t_made = t_now();
    {
      std::vector<std::thread> th;
      for (size_t d = 1; d < D; ++d) th.emplace_back(produce, d);
      produce(0);
      for (auto& t : th) t.join();
    }
    t_say("... fingerprints of every piece", t_made);
    bool ok = true;
    for (auto& L : loc)
      if (!L.err.empty()) ok = fail(L.err);
    x.owners.assign(D, Owner());
    for (size_t o = 0; o < D && ok; ++o) {
      Owner& ow = x.owners[o];
      ow.ctx = files[0]->shards[o].ctx;
      for (size_t d = 0; d < D; ++d)
        for (auto& b : loc[d].to[o]) ow.n += b.n;
      if (!ow.n) continue;
      ow.recv = fqg_device_alloc(ow.ctx, ow.n * sizeof(fqg_fp));
      if (named) ow.recv_names = fqg_device_alloc(ow.ctx, ow.n * FQG_NAME_REC_BYTES);
      if (!ow.recv || (named && !ow.recv_names)) {
        ok = fail("device allocation failed");
        break;
      }
      uint64_t at = 0;
      for (size_t d = 0; d < D && ok; ++d)
        for (auto& b : loc[d].to[o]) {
          if (fqg_device_copy(ow.ctx, (char*)ow.recv + at * sizeof(fqg_fp), files[0]->shards[d].ctx, b.src, b.n * sizeof(fqg_fp)) != 0 ||
              (named && fqg_device_copy(ow.ctx, (char*)ow.recv_names + at * FQG_NAME_REC_BYTES, files[0]->shards[d].ctx, b.nsrc,
                                        b.n * FQG_NAME_REC_BYTES) != 0)) {
            ok = fail(fqg_last_error(ow.ctx));
            break;
          }
          at += b.n;
        }
    }
    for (size_t d = 0; d < D; ++d) {
      x.name_bytes += loc[d].name_bytes;
      if (loc[d].buf) fqg_device_free(files[0]->shards[d].ctx, loc[d].buf);
      if (loc[d].nbuf) fqg_device_free(files[0]->shards[d].ctx, loc[d].nbuf);
    }
    if (!ok) release(x);
    return ok;
  }
};

}  // namespace fqhost
