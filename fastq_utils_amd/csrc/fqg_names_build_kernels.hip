// fqg_names_build_kernels.hip - the read-name table built PART BY PART in LDS (round 6).
//
// k_names_pass claims one slot per name with a CAS somewhere in a table of gigabytes: a hundred million scattered
// atomics cost 5.1 - 5.9 ms on this GPU wherever the table lives (tools/kbench/rmwbench.hip), and each one moves a
// 64-byte line in and out of the HBM for the 8 bytes it changes.  When a frame brings a sizeable part of what the table
// will hold (fastq_info on a file that is resident at once: the bench's 100 M names, a piece of gigabytes) the table is
// built the other way round - the KEYS go to where their slots are:
//
//   k_build_scatter<0>  the name digests of the streaming pass (16 bytes per header: hash, lengths, '@'; fqg_device.h)
//                       -> keys {hash, global record} scattered into <= 256 buckets by the top bits of their home slot.
//                       A workgroup bins a tile of 4096 digest slots in LDS and writes every bucket's share as one run,
//                       so the stores are runs of hundreds of bytes, not 16-byte scatters; it also does what
//                       k_names_pass does beside the table access (which chunks are trusted, which slots go to
//                       k_names_rest, the counts)
//   k_build_scatter<1>  every bucket again, by the next bits: one sub-bucket per 64 KiB PART of the table (8192 slots)
//   k_build_parts       one workgroup per part: the part's slots in LDS (empty, or as the table holds them when the index
//                       is not empty), its ~3000 keys inserted there with LDS atomics - the same linear probing, the same
//                       tag | record words, the same "every thread that meets its own name does atomicMin and reports
//                       max(previous owner, me)" as insert_name - and the part written out whole.  A key whose probe
//                       sequence runs past the end of its part (a few per thousand parts) is spilled
//   k_build_spill       the spilled keys: the ordinary CAS insert on the finished table (the slots between a key's home
//                       and the end of its part are all taken, which is why it spilled: the probing invariant holds)
//
// Buckets have a fixed capacity (mean + 8 standard deviations of a uniform hash); a bucket that overflows - names that
// share a hash, whatever makes a hash not uniform - raises a flag that k_build_parts looks at before it touches the table,
// and the caller takes k_names_pass instead.  Equality is decided on the name BYTES, as everywhere: two keys of one tag
// are compared through the line index.
#include "fqg_device.h"

namespace fqg {

// slots per part: 4096 (32 KiB of LDS: four workgroups of k_build_parts on a CU hide each other's phases - clear, insert,
// write out) up to a table of 2^28 slots, 8192 for one of 2^29
constexpr uint32_t kPartLogMin = 12, kPartLogMax = 13;
constexpr uint32_t kBuildTile = 4096;                   // keys a workgroup bins at once (64 KiB of staging)
constexpr uint32_t kBuildMaxBuckets = 256;              // per level
// the scatter kernels run 512 threads per workgroup: the 64 KiB staging area allows two workgroups per CU, and eight
// keys per thread instead of sixteen halve the registers (16 per thread in 256 threads: 257 registers, ONE workgroup of
// four wavefronts per CU - 2.0 ms for level 0)
constexpr int kBuildThreads = 512;

struct BuildKey {
  unsigned long long h, g;  // hash of the canonical name, global record index
};

struct BuildLevel {
  BuildKey* out;            // n_buckets_out * cap keys
  unsigned int* cursor;     // keys written per output bucket
  unsigned long long cap;   // room per output bucket
  uint32_t shift, bits;     // output bucket of a key: ((h & mask) >> shift) & ((1 << bits) - 1), under its input bucket
};

// exclusive prefix over one value per bucket (threads 0 .. 255 bring one, the others 0); every thread of the workgroup calls
__device__ __forceinline__ uint32_t build_scan_excl(uint32_t v, uint32_t* s_wave /*[kBuildThreads / kWave]*/, uint32_t* total) {
  const uint32_t incl = wave_scan_incl(v);
  if (lane_id() == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBuildThreads / kWave; ++w) {
    const uint32_t t = s_wave[w];
    if (w < (int)(threadIdx.x >> 6)) before += t;
    all += t;
  }
  __syncthreads();
  *total = all;
  return before + incl - v;
}
// tally_flush for a workgroup of kBuildThreads
__device__ __forceinline__ void build_tally_flush(IndexTally& t, IndexCall* __restrict__ call) {
  if (t.first_dup != kNoRecord) atomicMin(&call->first_dup, t.first_dup);
  if (t.first_wrong != kNoRecord) atomicMin(&call->first_wrong, t.first_wrong);
  if (t.first_missing != kNoRecord) atomicMin(&call->first_missing, t.first_missing);
  unsigned long long v[5] = {t.inserted, t.matched, t.name_bytes, t.seen, t.captured};
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] += __shfl_down(v[i], d, 64);
  __shared__ unsigned long long s_sum[kBuildThreads / kWave][5];
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 5; ++i) s_sum[threadIdx.x >> 6][i] = v[i];
  __syncthreads();
  if (threadIdx.x < 5) {
    unsigned long long a = 0;
    for (int w = 0; w < kBuildThreads / kWave; ++w) a += s_sum[w][threadIdx.x];
    unsigned long long* dst = threadIdx.x == 0   ? &call->inserted
                              : threadIdx.x == 1 ? &call->matched
                              : threadIdx.x == 2 ? &call->name_bytes
                              : threadIdx.x == 3 ? &call->seen
                                                 : &call->captured;
    if (a) atomicAdd(dst, a);
  }
}

// One tile: every thread brings up to kBuildTile / kBlock keys (live[i]: it has one), the workgroup writes them grouped
// by bucket.  `prefix` = the input bucket (level 1; 0 at level 0): output bucket = prefix << bits | own bits.
template <int PER>
__device__ __forceinline__ void build_scatter_tile(const BuildKey (&key)[PER], const bool (&live)[PER], uint64_t mask,
                                                   const BuildLevel& L, uint32_t prefix, IndexCall* __restrict__ call,
                                                   BuildKey* s_stage, uint8_t* s_bkt, uint32_t* s_cnt, uint32_t* s_ofs,
                                                   uint32_t* s_gbase, uint32_t* s_wave) {
  const uint32_t nb = 1u << L.bits;
  for (uint32_t i = threadIdx.x; i < kBuildMaxBuckets; i += kBuildThreads) s_cnt[i] = 0;
  __syncthreads();
  uint16_t bk[PER], rk[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    bk[i] = rk[i] = 0;
    if (live[i]) {
      bk[i] = (uint16_t)(((key[i].h & mask) >> L.shift) & (nb - 1u));
      rk[i] = (uint16_t)atomicAdd(&s_cnt[bk[i]], 1u);
    }
  }
  __syncthreads();
  // one reservation per bucket and tile; the buckets' places in the staging area
  const uint32_t mine = threadIdx.x < nb ? s_cnt[threadIdx.x] : 0u;
  uint32_t total;
  const uint32_t ofs = build_scan_excl(mine, s_wave, &total);
  if (threadIdx.x < nb) {
    s_ofs[threadIdx.x] = ofs;
    uint32_t gb = 0;
    if (mine) {
      gb = atomicAdd(&L.cursor[((uint64_t)prefix << L.bits) + threadIdx.x], mine);
      if ((unsigned long long)gb + mine > L.cap) atomicOr(&call->build_overflow, 1u);
    }
    s_gbase[threadIdx.x] = gb;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (live[i]) {
      const uint32_t e = s_ofs[bk[i]] + rk[i];
      s_stage[e] = key[i];
      s_bkt[e] = (uint8_t)bk[i];
    }
  __syncthreads();
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  for (uint32_t e = threadIdx.x; e < total; e += kBuildThreads) {
    const uint32_t b = s_bkt[e];
    const unsigned long long at = (unsigned long long)s_gbase[b] + (e - s_ofs[b]);
    if (at < L.cap) {
      u64x2 x;
      x.x = s_stage[e].h;
      x.y = s_stage[e].g;
      *reinterpret_cast<u64x2*>(L.out + (((uint64_t)prefix << L.bits) + b) * L.cap + at) = x;
    }
  }
  __syncthreads();
}

constexpr int kBuildPer = kBuildTile / kBuildThreads;  // 8

// LEVEL 0: from the digests of the streaming pass (the logic of k_names_pass<insert, DIGEST>); LEVEL 2: the same from its
// 64-byte capture RECORDS - an index that keeps name records (file 1 of a pair): the name is decoded here
// (name_from_record) and its record written to names[g] as insert_name writes it; LEVEL 1: from the buckets level 0 / 2
// wrote (in: n_in buckets of `in_cap`, in_count[] keys each; blockIdx.y = input bucket)
template <int LEVEL>
__global__ __launch_bounds__(kBuildThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_build_scatter(FrameView f, NamesView nv, uint64_t base, uint64_t mask, BuildLevel L,
                                                          const BuildKey* __restrict__ in, const unsigned int* __restrict__ in_count,
                                                          unsigned long long in_cap, IndexCall* __restrict__ call,
                                                          NameRec* __restrict__ names_out = nullptr, int fmt = 0, int is_pe = 0) {
  __shared__ BuildKey s_stage[kBuildTile];
  __shared__ uint8_t s_bkt[kBuildTile];  // (at most kBuildMaxBuckets = 256 buckets per level)
  __shared__ uint32_t s_cnt[kBuildMaxBuckets], s_ofs[kBuildMaxBuckets], s_gbase[kBuildMaxBuckets], s_wave[kBuildThreads / kWave];
  IndexTally t;
  if (LEVEL == 0 || LEVEL == 2) {
    constexpr bool RECORDS = LEVEL == 2;
    // what a slot's digest is worth depends on its CHUNK (was the speculated line type the true one, how many headers
    // did it see, its first rank): a tile's chunks - 4096 / K of them - are looked up once, into LDS, and the tile's 16
    // digest loads per thread are requested together, without a condition (slots nobody wrote hold whatever the
    // allocation held: they are not looked at)
    __shared__ uint64_t s_rank0[kBuildTile / 8];
    __shared__ uint32_t s_hc[kBuildTile / 8];  // hcount, or kNoCapture for a chunk that is not trusted
    const uint64_t n_slots = (uint64_t)nv.cr.n_chunks << nv.k_shift;
    const uint64_t n_tiles = (n_slots + kBuildTile - 1) / kBuildTile;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunks_per_tile = kBuildTile >> nv.k_shift;  // (K >= 8)
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      u64x2_t dg[kBuildPer];
      if (!RECORDS) {
#pragma unroll
        for (int i = 0; i < kBuildPer; ++i) {
          const uint64_t s = tile * kBuildTile + (uint64_t)i * kBuildThreads + threadIdx.x;
          dg[i] = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(nv.recs + (s < n_slots ? s : n_slots - 1) * kDigestWords));
        }
      }
      for (uint32_t ci = threadIdx.x; ci < chunks_per_tile; ci += kBuildThreads) {
        const uint64_t c64 = tile * chunks_per_tile + ci;
        uint32_t hcv = kNoCapture;
        uint64_t r0 = 0;
        if (c64 < nv.cr.n_chunks) {
          const uint32_t c = (uint32_t)c64;
          const uint32_t hc = nv.hcount[c], info = nv.cinfo[c];
          r0 = nv.cr.rank0(c);
          const bool trusted = hc != kNoCapture && hc <= nv.K && !(info & (kInfoUnknown | kInfoOneLine)) && (info & 3u) == ((uint32_t)r0 & 3u);
          nv.chunk_redo[c] = trusted ? 0 : 1;
          if (trusted) hcv = hc;
        }
        s_hc[ci] = hcv;
        s_rank0[ci] = r0;
      }
      __syncthreads();
      BuildKey key[kBuildPer];
      bool live[kBuildPer];
#pragma unroll
      for (int i = 0; i < kBuildPer; ++i) {
        const uint32_t in_tile = (uint32_t)i * kBuildThreads + threadIdx.x;
        const uint64_t s0 = tile * kBuildTile + (uint64_t)i * kBuildThreads + (threadIdx.x & ~63u);
        const uint64_t s = s0 + lane;
        const uint32_t ci = in_tile >> nv.k_shift, j = in_tile & (nv.K - 1u);
        const uint32_t hc = s_hc[ci];
        bool lv = s < n_slots && hc != kNoCapture && j < hc, redo = false;
        key[i].h = key[i].g = 0;
        if (lv && RECORDS) {
          unsigned long long w[kNameRecWords];
          const u64x2_t* src = reinterpret_cast<const u64x2_t*>(nv.recs + s * kNameRecWords);
#pragma unroll
          for (uint32_t q = 0; q < kNameRecWords / 2; ++q) {
            const u64x2_t x = __builtin_nontemporal_load(src + q);
            w[2 * q] = x.x;
            w[2 * q + 1] = x.y;
          }
          NameKey k;
          bool at_sign;
          uint32_t v;
          const bool ok = name_from_record(w, fmt, is_pe, k, &at_sign, &v);
          const uint64_t r = (s_rank0[ci] + v) >> 2;
          if (r >= f.n_records) lv = false;  // a header of the incomplete tail
          else if (!ok) {
            redo = true;
            lv = false;
          } else {
            ++t.captured;
            ++t.seen;
            if (names_out) {  // (a record without '@' or with a repeated name gets one too: nothing will ever point at it)
              u64x2_t* dst = reinterpret_cast<u64x2_t*>(names_out + (base + r));
              u64x2_t x;
              x.x = k.n;
              x.y = k.nm[0];
              __builtin_nontemporal_store(x, dst);
#pragma unroll
              for (uint32_t q = 1; q < 4; ++q) {
                x.x = k.nm[2 * q - 1];
                x.y = k.nm[2 * q];
                __builtin_nontemporal_store(x, dst + q);
              }
            }
            if (!at_sign) {  // fastq_get_readname refuses it (src/fastq.c:448)
              t.first_wrong = r < t.first_wrong ? r : t.first_wrong;
              lv = false;
            } else {
              key[i].h = k.h;
              key[i].g = base + r;
              ++t.inserted;
              t.name_bytes += k.acct;
            }
          }
        } else if (lv) {
          const uint32_t meta = (uint32_t)dg[i].y;
          const uint64_t r = (s_rank0[ci] + ((meta >> 20) & 511u)) >> 2;
          if (r >= f.n_records) lv = false;  // a header of the incomplete tail
          else if (!(meta & kDigestOk)) {
            redo = true;
            lv = false;
          } else {
            ++t.captured;
            ++t.seen;
            if (!(meta & kDigestAt)) {  // fastq_get_readname refuses it (src/fastq.c:448)
              t.first_wrong = r < t.first_wrong ? r : t.first_wrong;
              lv = false;
            } else {
              key[i].h = dg[i].x;
              key[i].g = base + r;
              // (counted here, once per persistent workgroup: a repeated name takes itself and its bytes back in
              // k_build_parts - 65 536 workgroups adding to one counter there would be most of that kernel)
              ++t.inserted;
              t.name_bytes += (meta >> 10) & 1023u;
            }
          }
        }
        const unsigned long long rm = __ballot(redo);
        if (lane == 0 && s0 < n_slots) nv.redo_bits[s0 >> 6] = rm;
        live[i] = lv;
      }
      build_scatter_tile<kBuildPer>(key, live, mask, L, 0u, call, s_stage, s_bkt, s_cnt, s_ofs, s_gbase, s_wave);
    }
    build_tally_flush(t, call);
  } else {
    const uint32_t b = blockIdx.y;
    const unsigned long long n = in_count[b] < in_cap ? in_count[b] : in_cap;
    const BuildKey* src = in + (uint64_t)b * in_cap;
    for (unsigned long long t0 = (unsigned long long)blockIdx.x * kBuildTile; t0 < n; t0 += (unsigned long long)gridDim.x * kBuildTile) {
      BuildKey key[kBuildPer];
      bool live[kBuildPer];
#pragma unroll
      for (int i = 0; i < kBuildPer; ++i) {
        const unsigned long long e = t0 + (unsigned long long)i * kBuildThreads + threadIdx.x;
        live[i] = e < n;
        key[i].h = key[i].g = 0;
        if (live[i]) {
          const u64x2_t x = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(src + e));
          key[i].h = x.x;
          key[i].g = x.y;
        }
      }
      build_scatter_tile<kBuildPer>(key, live, mask, L, b, call, s_stage, s_bkt, s_cnt, s_ofs, s_gbase, s_wave);
    }
  }
}

// rare: two keys of one tag - is the name of this frame's record r the stored record's?  (not inlined: the header in
// registers, the byte-wise fallbacks ... would sit in the register budget of a loop that never needs them - 169
// registers instead of 81 for the 8192-slot kernel.)  *acct = what the reference accounts for the name.
__device__ __attribute__((noinline)) bool build_same_name(const FrameView& f, const IndexView& ix, uint64_t r, unsigned long long other,
                                                           uint32_t* acct) {
  NameKey k;
  bool at_sign;
  name_from_image(f, r, ix.fmt, ix.is_pe, ix.may_have_nul, k, &at_sign);
  *acct = k.acct;
  return stored_name_is(ix, other, NameAt{f, r}(), k.n);
}

// One workgroup per part of the table.  keys: n_parts buckets of `cap`; count[p] keys in part p.
template <uint32_t LOG>
__global__ __launch_bounds__(kBlock) void k_build_parts(FrameView f, IndexView ix, uint64_t record_base, const BuildKey* __restrict__ keys,
                                                        const unsigned int* __restrict__ count, unsigned long long cap,
                                                        int load_existing, BuildKey* __restrict__ spill, unsigned long long spill_cap,
                                                        IndexCall* __restrict__ call) {
  constexpr uint32_t kPartSlots = 1u << LOG;
  __shared__ unsigned long long s_slots[kPartSlots];
  if (__hip_atomic_load(&call->build_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (the table is untouched)
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const uint64_t p = blockIdx.x;
  u64x2* const gpart = reinterpret_cast<u64x2*>(ix.slots + p * kPartSlots);
  if (load_existing) {
    for (uint32_t i = threadIdx.x; i < kPartSlots / 2; i += kBlock) reinterpret_cast<u64x2*>(s_slots)[i] = gpart[i];
  } else {
    u64x2 e;
    e.x = e.y = kSlotEmpty;
    for (uint32_t i = threadIdx.x; i < kPartSlots / 2; i += kBlock) reinterpret_cast<u64x2*>(s_slots)[i] = e;
  }
  __syncthreads();
  const unsigned long long n = count[p] < cap ? count[p] : cap;
  const BuildKey* src = keys + p * cap;
  for (unsigned long long e = threadIdx.x; e < n; e += kBlock) {
    const u64x2_t x = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(src + e));
    const unsigned long long h = x.x, g = x.y;
    const unsigned long long me = ((h >> 40) << 40) | g;
    uint32_t at = (uint32_t)h & (kPartSlots - 1u);
    bool done = false;
    for (; at < kPartSlots; ++at) {
      const unsigned long long cur = atomicCAS(&s_slots[at], kSlotEmpty, me);
      if (cur == kSlotEmpty) {
        done = true;
        break;
      }
      if ((cur >> 40) == (me >> 40)) {
        // rare: the tags agree - the bytes decide, through the line index (this frame's record, the other one wherever
        // it lives)
        uint32_t acct;
        if (build_same_name(f, ix, g - record_base, cur & kIdxMask, &acct)) {
          const unsigned long long prev = atomicMin(&s_slots[at], me);
          const unsigned long long late = (prev & kIdxMask) > g ? (prev & kIdxMask) : g;
          atomicMin(&call->first_dup, late);
          atomicAdd(&call->inserted, ~0ull);                              // (k_build_scatter counted it,
          atomicAdd(&call->name_bytes, 0ull - (unsigned long long)acct);  //  and its bytes)
          done = true;
          break;
        }
      }
    }
    if (!done) {
      const unsigned long long si = atomicAdd(&call->spilled, 1ull);
      if (si < spill_cap) {
        spill[si].h = h;
        spill[si].g = g;
      } else atomicOr(&call->table_full, 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kPartSlots / 2; i += kBlock)
    __builtin_nontemporal_store(reinterpret_cast<const u64x2*>(s_slots)[i], gpart + i);
}

// the keys that ran past the end of their part: insert_name's loop on the finished table, from the key's home slot
__global__ __launch_bounds__(kBlock) void k_build_spill(FrameView f, IndexView ix, uint64_t record_base, const BuildKey* __restrict__ spill,
                                                        unsigned long long spill_cap, IndexCall* __restrict__ call) {
  unsigned long long n = call->spilled;
  if (n > spill_cap) n = spill_cap;
  if (__hip_atomic_load(&call->build_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) n = 0;
  for (unsigned long long e = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (unsigned long long)gridDim.x * kBlock) {
    const unsigned long long h = spill[e].h, g = spill[e].g;
    const unsigned long long me = ((h >> 40) << 40) | g;
    uint64_t at = h & ix.mask;
    bool done = false;
    for (uint64_t probes = 0; probes <= ix.mask && !done; ++probes, at = (at + 1) & ix.mask) {
      const unsigned long long cur = atomicCAS(&ix.slots[at], kSlotEmpty, me);
      if (cur == kSlotEmpty) done = true;
      else if ((cur >> 40) == (me >> 40)) {
        uint32_t acct;
        if (build_same_name(f, ix, g - record_base, cur & kIdxMask, &acct)) {
          const unsigned long long prev = atomicMin(&ix.slots[at], me);
          const unsigned long long late = (prev & kIdxMask) > g ? (prev & kIdxMask) : g;
          atomicMin(&call->first_dup, late);
          atomicAdd(&call->inserted, ~0ull);
          atomicAdd(&call->name_bytes, 0ull - (unsigned long long)acct);
          done = true;
        }
      }
    }
    if (!done) atomicOr(&call->table_full, 1u);
  }
}

}  // namespace fqg
