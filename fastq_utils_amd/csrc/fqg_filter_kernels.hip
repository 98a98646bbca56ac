// fqg_filter_kernels.hip - per-record filters of the reference's small FASTQ tools on the GPU:
//   FQG_FILTER_N        fastq_filter_n        (reference src/fastq_filter_n.c:75-91): a record is
//                       dropped when its sequence holds more N/n than max_n % of read_len allow
//   FQG_FILTER_POLY_AT  fastq_trim_poly_at    (reference src/fastq_trim_poly_at.c:77-119, :214-222):
//                       a poly-A (3') or else poly-T (5') run is cut off, records that end up
//                       shorter than min_len are dropped
// Same three launches and the same tiles as the barcode transform (fqg_barcode_kernels.hip): the
// plan kernel decides and measures one record per lane from LDS, a 64-bit scan places the records,
// the emit kernel writes one record per lane into the tile's LDS output area and flushes it with
// 16-byte stores.  The decision is recomputed by the emit kernel (a few LDS reads) rather than
// stored.  Tiles that do not fit LDS (long reads) go through the *_direct paths.
#include "fqg_device.h"

namespace fqg {

struct RfParams {
  int32_t mode;
  uint32_t max_n;        // FILTER_N: percent, already clamped to 100
  int64_t min_poly;      // POLY_AT: min_poly_at_len
  uint64_t min_len;      // POLY_AT: min_len, compared unsigned like the reference's `read_len >= min_len`
};

enum : uint8_t { kRfDiscard = 1, kRfTrimmed = 2 };

// what is printed for one record: hdr1, seq[s_from, s_from + s_n) (+'\n'), hdr2,
// qual[q_from, q_from + q_n) followed by qual[q2_from, q2_from + q2_n) (+'\n')
struct RfCut {
  uint32_t s_from, s_n, s_nl;
  uint32_t q_from, q_n, q2_from, q2_n, q_nl;
  uint8_t flags;
};

template <bool WIDE>
__device__ __forceinline__ uint32_t rf_count_n(const BcLine& seq) {
  // N or n among the characters before the line's end (src/fastq_filter_n.c:79-85)
  const uint32_t n = seq.len;
  uint32_t cnt = 0;
  if (WIDE) {
    for (uint32_t i = 0; i < n; i += 8) {
      uint64_t m = bytes_eq(ld8(seq.p + i) | 0x2020202020202020ull, (uint8_t)'n');
      if (n - i < 8) m &= (1ull << (8 * (n - i))) - 1ull;
      cnt += (uint32_t)__builtin_popcountll(m);
    }
  } else {
    // the line where it lies in the image: 16-byte loads on the image's own alignment, the two ends masked (the image
    // is 16-byte aligned and readable up to the next multiple of 16 behind its last byte, as for the span copies)
    const uintptr_t a0 = (uintptr_t)seq.p & ~(uintptr_t)15;
    const uint32_t pre = (uint32_t)((uintptr_t)seq.p - a0), total = pre + n;
    for (uint32_t at = 0; at < total; at += 16) {
      const uint4 v = *reinterpret_cast<const uint4*>(a0 + at);
      uint64_t m0 = bytes_eq(((uint64_t)v.x | ((uint64_t)v.y << 32)) | 0x2020202020202020ull, (uint8_t)'n');
      uint64_t m1 = bytes_eq(((uint64_t)v.z | ((uint64_t)v.w << 32)) | 0x2020202020202020ull, (uint8_t)'n');
      const uint32_t lo = at == 0 ? pre : 0u, hi = total - at < 16u ? total - at : 16u;  // bytes [lo, hi) of the piece count
      if (lo | (hi ^ 16u)) {
        auto keep = [](uint32_t from, uint32_t to) {  // bytes [from, to) of an 8-byte word, 0 <= from, to <= 8
          const uint64_t below_to = to >= 8 ? ~0ull : (1ull << (8 * to)) - 1ull;
          const uint64_t below_from = from >= 8 ? ~0ull : (1ull << (8 * from)) - 1ull;
          return below_to & ~below_from;
        };
        m0 &= keep(lo < 8 ? lo : 8, hi < 8 ? hi : 8);
        m1 &= keep(lo > 8 ? lo - 8 : 0, hi > 8 ? hi - 8 : 0);
      }
      cnt += (uint32_t)__builtin_popcountll(m0) + (uint32_t)__builtin_popcountll(m1);
    }
  }
  return cnt;
}

template <bool WIDE>
__device__ __forceinline__ RfCut rf_decide(const RfParams& P, const BcLine (&ln)[4]) {
  const uint32_t L = ln[1].len + ln[1].nl;   // read_len = strlen(seq), the '\n' included
  const uint32_t Lq = ln[3].len + ln[3].nl;  // strlen(qual)
  RfCut c{0, L, 0, 0, Lq, 0, 0, 0, 0};
  if (P.mode == FQG_FILTER_N) {
    const uint32_t max_num_n = (uint32_t)((unsigned long)L * P.max_n / 100ul);
    if (rf_count_n<WIDE>(ln[1]) > max_num_n) c.flags = kRfDiscard;
    return c;
  }
  unsigned long read_len = L;
  if (P.min_poly > 0) {
    // 3' end: from seq[read_len - 2] backwards over N A n a
    long x = (long)((unsigned long)L - 2ul), matched1 = 0;
    for (; x >= 0; --x) {
      const uint8_t ch = ln[1].p[x];
      if (ch != 'N' && ch != 'A' && ch != 'n' && ch != 'a') break;
      ++matched1;
    }
    if (matched1 >= P.min_poly) {
      // seq[x+1] = '\n', seq[x+2] = 0; the same two stores into qual, whatever its length
      const uint32_t keep = (uint32_t)(x + 1);
      c.s_n = keep;
      c.s_nl = 1;
      c.q_n = keep < Lq ? keep : Lq;
      c.q_nl = keep <= Lq ? 1u : 0u;
      c.flags |= kRfTrimmed;
      read_len = L - (unsigned long)matched1;
    } else {
      // 5' end: N T n t from the start (the '\n' stops the scan)
      uint32_t matched2 = 0;
      for (uint32_t i = 0; i < L; ++i) {
        const uint8_t ch = ln[1].p[i];
        if (ch != 'N' && ch != 'T' && ch != 'n' && ch != 't') break;
        ++matched2;
      }
      if ((long)matched2 >= P.min_poly) {
        // seq[i] = seq[i + matched2], qual[i] = qual[i + matched2] for i <= read_len - matched2
        c.s_from = matched2;
        c.s_n = L - matched2;
        if (Lq >= matched2) {
          c.q_from = matched2;
          if (Lq <= L) c.q_n = Lq - matched2;
          else {  // a quality line longer than the sequence keeps its unshifted end
            c.q_n = L + 1 - matched2;
            c.q2_from = L - matched2 + 1;
            c.q2_n = Lq - c.q2_from;
          }
        } else {
          c.q_n = 0;  // (the reference prints stale buffer content here: undefined)
        }
        c.flags |= kRfTrimmed;
        read_len = L - matched2;
      }
    }
  }
  if (!(read_len >= P.min_len)) c.flags |= kRfDiscard;
  return c;
}

__device__ __forceinline__ uint32_t rf_out_len(const BcLine (&ln)[4], const RfCut& c) {
  return ln[0].len + ln[0].nl + c.s_n + c.s_nl + ln[2].len + ln[2].nl + c.q_n + c.q2_n + c.q_nl;
}

template <class W>
__device__ __forceinline__ void rf_emit(const BcLine (&ln)[4], const RfCut& c, W& w) {
  w.bytes(ln[0].p, ln[0].len + ln[0].nl);
  w.bytes(ln[1].p + c.s_from, c.s_n);
  if (c.s_nl) w.ch('\n');
  w.bytes(ln[2].p, ln[2].len + ln[2].nl);
  w.bytes(ln[3].p + c.q_from, c.q_n);
  if (c.q2_n) w.bytes(ln[3].p + c.q2_from, c.q2_n);
  if (c.q_nl) w.ch('\n');
}

// The plan: ONE THREAD PER RECORD, straight from the image.  What it decides needs the four line lengths (the line
// index: 40 bytes per record, neighbouring lanes neighbouring words) and, of the record's bytes, only the sequence line -
// all of it for fastq_filter_n (counted in 16-byte pieces), its two ends for fastq_trim_poly_at (the scans stop at the
// first character that is no A / T / N: one or two bytes per record on ordinary reads).  The first form of this kernel
// staged whole tiles in LDS the way the emit kernel does and read every record twice (39 GB for the 35 GB image of
// the bench, 8.2 ms of 22.5); a record's header and quality bytes are nothing the decision looks at.
// the discarded / trimmed records of a plan kernel's thread -> the call's totals: one add per wavefront, at its end
__device__ __forceinline__ void rf_count_flush(uint32_t d, uint32_t t, BcCall* __restrict__ call) {
  d = wave_sum32(d);
  t = wave_sum32(t);
  if ((threadIdx.x & 63) == 0) {
    if (d) atomicAdd(&call->discarded, (unsigned long long)d);
    if (t) atomicAdd(&call->short_warnings, (unsigned long long)t);  // (the field holds the trimmed count here)
  }
}
__global__ __launch_bounds__(kBlock) void k_rf_plan_records(BcParams F, RfParams P, uint64_t n_rec, uint8_t* __restrict__ status,
                                                            uint32_t* __restrict__ len1, BcCall* __restrict__ call) {
  uint32_t n_d = 0, n_t = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < n_rec; k += (uint64_t)gridDim.x * kBlock) {
    BcLine G[4];
    bc_lines(F.f[1], k, G);
    const RfCut c = rf_decide<false>(P, G);
    status[k] = c.flags;
    len1[k] = (c.flags & kRfDiscard) ? 0u : rf_out_len(G, c);
    n_d += (c.flags & kRfDiscard) ? 1u : 0u;
    n_t += (c.flags & kRfTrimmed) ? 1u : 0u;
  }
  rf_count_flush(n_d, n_t, call);
}

// fastq_filter_n looks at every character of the sequence line: EIGHT LANES PER RECORD read it where it lies, 32 bytes
// a lane and round on the image's own 16-byte alignment - an instruction of the wavefront covers eight neighbouring
// records' sequence lines, a handful of cache lines, where one lane per record asked for 64 lines at once (10.2 ms for
// the 100 M reads of the bench, more than the tile-staging plan it replaced).  The counts meet in the group's first lane.
__global__ __launch_bounds__(kBlock) void k_rf_plan_n(BcParams F, RfParams P, uint64_t n_rec, uint8_t* __restrict__ status,
                                                      uint32_t* __restrict__ len1, BcCall* __restrict__ call) {
  uint32_t n_d = 0;
  const int lane = (int)(threadIdx.x & 63), sub = lane & 7;
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  for (uint64_t k0 = wave * 8; k0 < n_rec; k0 += n_waves * 8) {
    const uint64_t k = k0 + (uint64_t)(lane >> 3);
    const bool valid = k < n_rec;
    BcLine G[4];
    bc_lines(F.f[1], valid ? k : n_rec - 1, G);
    const uint32_t n = G[1].len;
    const uintptr_t a0 = (uintptr_t)G[1].p & ~(uintptr_t)15;
    const uint32_t pre = (uint32_t)((uintptr_t)G[1].p - a0), total = pre + n;
    uint32_t cnt = 0;
    for (uint32_t at = (uint32_t)sub * 32u; __ballot(at < total) != 0; at += 256u) {
      if (at >= total) continue;
#pragma unroll
      for (uint32_t h = 0; h < 32u; h += 16u) {
        const uint32_t b = at + h;
        if (b >= total) break;
        const uint4 v = *reinterpret_cast<const uint4*>(a0 + b);
        uint64_t m0 = bytes_eq(((uint64_t)v.x | ((uint64_t)v.y << 32)) | 0x2020202020202020ull, (uint8_t)'n');
        uint64_t m1 = bytes_eq(((uint64_t)v.z | ((uint64_t)v.w << 32)) | 0x2020202020202020ull, (uint8_t)'n');
        const uint32_t lo = b == 0 ? pre : 0u, hi = total - b < 16u ? total - b : 16u;  // bytes [lo, hi) of the piece count
        if (lo | (hi ^ 16u)) {
          auto keep = [](uint32_t from, uint32_t to) {
            const uint64_t below_to = to >= 8 ? ~0ull : (1ull << (8 * to)) - 1ull;
            const uint64_t below_from = from >= 8 ? ~0ull : (1ull << (8 * from)) - 1ull;
            return below_to & ~below_from;
          };
          m0 &= keep(lo < 8 ? lo : 8, hi < 8 ? hi : 8);
          m1 &= keep(lo > 8 ? lo - 8 : 0, hi > 8 ? hi - 8 : 0);
        }
        cnt += (uint32_t)__builtin_popcountll(m0) + (uint32_t)__builtin_popcountll(m1);
      }
    }
    cnt += (uint32_t)__shfl_xor((int)cnt, 1, 64);
    cnt += (uint32_t)__shfl_xor((int)cnt, 2, 64);
    cnt += (uint32_t)__shfl_xor((int)cnt, 4, 64);
    if (valid && sub == 0) {
      // rf_decide for FQG_FILTER_N: nothing is cut, the record is dropped when it holds more N/n than max_n % of
      // read_len allow (src/fastq_filter_n.c:78-86; read_len = strlen(seq), the '\n' included)
      const uint32_t L = G[1].len + G[1].nl, Lq = G[3].len + G[3].nl;
      const uint32_t max_num_n = (uint32_t)((unsigned long)L * P.max_n / 100ul);
      const bool drop = cnt > max_num_n;
      status[k] = drop ? kRfDiscard : 0;
      len1[k] = drop ? 0u : G[0].len + G[0].nl + L + G[2].len + G[2].nl + Lq;
      n_d += drop ? 1u : 0u;
    }
  }
  rf_count_flush(n_d, 0u, call);
}

// ... and what it decides per EMIT tile - do the tile's input span and its output fit the emit kernel's LDS areas
// (bc_emit_tile_fits for the one input file): one thread per tile
// (behind the scan of the lengths: a tile's output bytes are the difference of two offsets - the sum over its T lengths,
// read one by one at a stride of T words from lane to lane, was most of this kernel's millisecond per 100 M records)
__global__ __launch_bounds__(kBlock) void k_rf_tile_flags(BcParams F, BcTile tc, uint64_t n_rec, const uint32_t* __restrict__ len1,
                                                          const unsigned long long* __restrict__ off,
                                                          const unsigned long long* __restrict__ span_excl,
                                                          uint8_t* __restrict__ tile_big, BcCall* __restrict__ call) {
  const uint64_t n_tiles = (n_rec + tc.T - 1) / tc.T;
  const BcFile& f = F.f[1];
  bool big = false;
  const uint64_t tile = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (tile < n_tiles) {
    const uint64_t k0 = tile * tc.T, k1 = (k0 + tc.T < n_rec ? k0 + tc.T : n_rec) - 1;
    const uint64_t sum = (off[k1] + span_excl[k1 / kScan64Span] + len1[k1]) - (off[k0] + span_excl[k0 / kScan64Span]);
    const uint64_t r0 = f.first + k0 * f.step + f.add, r1 = f.first + k1 * f.step + f.add;
    const uint64_t s0 = r0 == 0 ? 0 : f.fv.line_end[4 * r0 - 1] + 1, e3l = f.fv.line_end[4 * r1 + 3];
    const uint64_t n = (e3l < f.fv.nbytes ? e3l + 1 : e3l) - s0;
    const uint32_t skew = (uint32_t)((uintptr_t)(f.fv.img + s0) & 15u);
    bool fit = n <= (uint64_t)tc.in_cap;
    if (fit) fit = (uint64_t)((skew + (uint32_t)n + 15u) >> 4) * 16u + 32u <= (uint64_t)tc.in_cap;
    big = F.has_nul || !fit || sum + 32 > tc.out_cap;  // (NUL bytes: lines are C strings - the record-by-record kernel, bc_lines)
    tile_big[tile] = big ? 1 : 0;
  }
  const unsigned long long m = __ballot(big);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(&call->big, (unsigned long long)__builtin_popcountll(m));
}

__global__ __launch_bounds__(kWave, 2) void k_rf_emit_tile(BcParams F, RfParams P, BcTile tc, uint64_t n_rec,
                                                        const uint8_t* __restrict__ status,
                                                        const uint8_t* __restrict__ tile_big, EmitOut o) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
  uint8_t* s_in = s_lds;
  uint8_t* s_out = s_lds + tc.in_cap;
  const int lane = (int)threadIdx.x;
  const uint64_t n_tiles = (n_rec + tc.T - 1) / tc.T;
  auto tile_size = [&](uint64_t tile) {
    const uint64_t left = n_rec - tile * tc.T;
    return (uint32_t)(left < (uint64_t)tc.T ? left : (uint64_t)tc.T);
  };
  auto geo_of = [&](uint64_t tile, TileGeo& tg) {
    const uint32_t Tn = tile_size(tile);
    const uint64_t k = tile * tc.T + ((uint32_t)lane < Tn ? (uint32_t)lane : Tn - 1);
    tg.big = tile_big[tile];
    tg.st = status[k];
    bc_geo_load(F.f[1], k, tg.f[1]);
    tg.off[1] = o.off[k], tg.sum[1] = o.sum[k / kScan64Span];  // (added where it is used, see TileGeo)
  };
  // three tiles under way per wavefront, every request without a branch (see k_bc_emit_tile)
  const uint64_t stride = gridDim.x;
  auto clamp_tile = [&](uint64_t t) { return t < n_tiles ? t : n_tiles - 1; };
  TileGeo cur, nxt, nx2;
  bc_u32x4 pf[kSpanPf];
  if (blockIdx.x < n_tiles) {
    geo_of(blockIdx.x, cur);
    geo_of(clamp_tile(blockIdx.x + stride), nxt);
    SpanPlan sp;
    bc_span_plan<false, 0x02>(F, cur, (int)tile_size(blockIdx.x) - 1, tc.in_cap, sp);
    bc_span_fetch(sp, lane, pf);
  }
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += stride, cur = nxt, nxt = nx2) {
    const uint32_t Tn = tile_size(tile);
    const bool valid = (uint32_t)lane < Tn;
    const bool big = __builtin_amdgcn_readfirstlane((int)cur.big) != 0;
    BcLine L[kBcFiles][4];
    {
      SpanPlan sp;
      bc_span_plan<false, 0x02>(F, cur, (int)Tn - 1, tc.in_cap, sp);
      if (!big) bc_span_land(sp, lane, pf, s_in);  // (fits: the plan checked)
      bc_span_lines<false, 0x02>(F, cur, sp, s_in, L);
    }
    {
      const uint64_t tn = clamp_tile(tile + stride);
      SpanPlan sp;
      bc_span_plan<false, 0x02>(F, nxt, (int)tile_size(tn) - 1, tc.in_cap, sp);
      bc_span_fetch(sp, lane, pf);
    }
    geo_of(clamp_tile(tile + 2 * stride), nx2);
    if (big) continue;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool keep = valid && !(cur.st & kRfDiscard);
    const RfCut c = rf_decide<true>(P, L[1]);
    const uint32_t my_len = keep ? rf_out_len(L[1], c) : 0u;
    const unsigned long long where = cur.off[1] + cur.sum[1];
    const unsigned long long tile_at = rfl64(where);
    const uint32_t start = (uint32_t)(where - tile_at);
    const uint32_t total = wave_max32(keep ? start + my_len : 0u);
    uint8_t* dst = o.out + tile_at;
    const uint32_t skew = (uint32_t)((uintptr_t)dst & 15u);
    if (keep) {
      LaneWriter w{s_out + skew + start};
      rf_emit(L[1], c, w);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    emit_flush(s_out, skew, total, dst, lane);
    __builtin_amdgcn_wave_barrier();
  }
}

// records of the tiles that do not fit LDS: one wavefront per record, straight from image to image
__global__ __launch_bounds__(kBlock) void k_rf_emit_direct(BcParams F, RfParams P, BcTile tc, uint64_t n_rec,
                                                           const uint8_t* __restrict__ status,
                                                           const uint8_t* __restrict__ tile_big, EmitOut o) {
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const int lane = (int)(threadIdx.x & 63), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (uint64_t k = (uint64_t)blockIdx.x * (kBlock / kWave) + wv; k < n_rec; k += n_waves) {
    if (!tile_big[k / tc.T] || (status[k] & kRfDiscard)) continue;
    BcLine G[4];
    bc_lines(F.f[1], k, G);
    const RfCut c = rf_decide<false>(P, G);
    Writer w{o.out + o.off[k] + o.sum[k / kScan64Span], lane};
    rf_emit(G, c, w);
  }
}

// ---- ordered gather of whole records (fastq_filterpair: the paired / unpaired partitions) ----------
// What fastq_write_entry / fastq_quick_copy_entry write per record (src/fastq.c:125-157, 265-272): the four lines
// as they stand in the input.  list[k] = record of the frame that comes k-th in the output.
__global__ __launch_bounds__(kBlock) void k_gather_lens(FrameView f, const unsigned long long* __restrict__ list,
                                                        uint64_t n, uint32_t* __restrict__ lens) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const uint64_t r = list[k];
  const uint64_t b = r == 0 ? 0 : f.line_end[4 * r - 1] + 1;
  const uint64_t e = f.line_end[4 * r + 3];
  lens[k] = (uint32_t)(e - b) + (e < f.nbytes ? 1u : 0u);
}
// One wavefront per 64 consecutive entries of the list: every lane fetches where ONE record lies, where it goes and how
// long it is (one round trip to memory for 64 records - per record these were three dependent ones), then the records
// are copied one after the other by all lanes, 16-byte pieces at any alignment.
__global__ __launch_bounds__(kBlock) void k_gather_copy(FrameView f, const unsigned long long* __restrict__ list, uint64_t n,
                                                        const unsigned long long* __restrict__ off_local,
                                                        const unsigned long long* __restrict__ off_span,
                                                        const uint32_t* __restrict__ lens, uint8_t* __restrict__ out) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));
  const int lane = (int)(threadIdx.x & 63);
  const uint64_t stride = (uint64_t)gridDim.x * (kBlock / kWave) * kWave;
  for (uint64_t k0 = ((uint64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kWave; k0 < n; k0 += stride) {
    const uint64_t k = k0 + (uint64_t)lane;
    uint64_t s_off = 0, d_off = 0;
    uint32_t len = 0;
    if (k < n) {
      const uint64_t r = list[k];
      s_off = r == 0 ? 0 : f.line_end[4 * r - 1] + 1;
      d_off = off_local[k] + off_span[k / kScan64Span];
      len = lens[k];
    }
    const int cnt = n - k0 < (uint64_t)kWave ? (int)(n - k0) : kWave;
    // Neighbours in the list are usually neighbours in the image (the mates of two files in the same order, the runs
    // of reads without a mate): 64 records that follow each other are ONE span and are copied as one, all lanes busy.
    {
      const uint64_t next_s = __shfl_down(s_off, 1, 64);
      const bool breaks = lane + 1 < cnt && s_off + len != next_s;
      if (__ballot(breaks) == 0) {
        const uint8_t* __restrict__ src = f.img + rfl64(s_off);
        uint8_t* __restrict__ dst = out + rfl64(d_off);
        const uint64_t total = rl64(d_off + len, cnt - 1) - rfl64(d_off);
        // four pieces per lane in flight (a loop of one load and one store waits for every piece on its own)
        uint64_t o = (uint64_t)lane * 16u;
        for (; o + 3u * 16u * kWave + 16u <= total; o += 4u * 16u * kWave) {
          u32x4 v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4*>(src + o + (uint64_t)q * 16u * kWave);
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(dst + o + (uint64_t)q * 16u * kWave) = v[q];
        }
        for (; o < total; o += 16u * kWave) {
          if (o + 16u <= total) *reinterpret_cast<u32x4*>(dst + o) = *reinterpret_cast<const u32x4*>(src + o);
          else
            for (uint64_t q = o; q < total; ++q) dst[q] = src[q];
        }
        continue;
      }
    }
    // four records at a time: their loads are all in flight before the first store (records of at most 1 KiB - one
    // 16-byte piece per lane; longer ones take the loop below)
    const bool small = __ballot(len > 16u * kWave) == 0;
    int j = 0;
    if (small) {
      for (; j + 4 <= cnt; j += 4) {
        u32x4 v[4];
        uint32_t L[4];
        uint8_t* dst[4];
        const uint32_t o = (uint32_t)lane * 16u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint8_t* __restrict__ src = f.img + rl64(s_off, j + q);
          dst[q] = out + rl64(d_off, j + q);
          L[q] = (uint32_t)__builtin_amdgcn_readlane((int)len, j + q);
          if (o + 16u <= L[q]) v[q] = *reinterpret_cast<const u32x4*>(src + o);
          else if (o < L[q])
            for (uint32_t t = o; t < L[q]; ++t) dst[q][t] = src[t];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (o + 16u <= L[q]) *reinterpret_cast<u32x4*>(dst[q] + o) = v[q];
      }
    }
    for (; j < cnt; ++j) {
      const uint8_t* __restrict__ src = f.img + rl64(s_off, j);
      uint8_t* __restrict__ dst = out + rl64(d_off, j);
      const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, j);
      for (uint32_t o = (uint32_t)lane * 16u; o < L; o += 16u * kWave) {
        if (o + 16u <= L) *reinterpret_cast<u32x4*>(dst + o) = *reinterpret_cast<const u32x4*>(src + o);
        else
          for (uint32_t q = o; q < L; ++q) dst[q] = src[q];
      }
    }
  }
}

// ... of an image with NUL bytes: what GZ_WRITE / gzputs write of a line is the C string (bc_clip_nul) - four pieces per
// record.  Such images are rare and small: one wavefront per record, byte by byte.
__device__ __forceinline__ void gather_lines_nul(const FrameView& f, uint64_t r, BcLine ln[4]) {
  BcFile bf;
  bf.fv = f;
  bf.first = 0;
  bf.step = 1;
  bf.add = 0;
  bf.has_nul = 1;
  bc_lines(bf, r, ln);
}
__global__ __launch_bounds__(kBlock) void k_gather_lens_nul(FrameView f, const unsigned long long* __restrict__ list,
                                                            uint64_t n, uint32_t* __restrict__ lens) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  BcLine ln[4];
  gather_lines_nul(f, list[k], ln);
  lens[k] = ln[0].len + ln[0].nl + ln[1].len + ln[1].nl + ln[2].len + ln[2].nl + ln[3].len + ln[3].nl;
}
__global__ __launch_bounds__(kBlock) void k_gather_copy_nul(FrameView f, const unsigned long long* __restrict__ list, uint64_t n,
                                                            const unsigned long long* __restrict__ off_local,
                                                            const unsigned long long* __restrict__ off_span,
                                                            uint8_t* __restrict__ out) {
  const int lane = (int)(threadIdx.x & 63);
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  for (uint64_t k = (uint64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); k < n; k += n_waves) {
    BcLine ln[4];
    gather_lines_nul(f, list[k], ln);
    uint8_t* dst = out + off_local[k] + off_span[k / kScan64Span];
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const uint32_t m = ln[i].len + ln[i].nl;  // (a line that kept its '\n' holds no NUL: the newline is the byte behind its text)
      for (uint32_t o = (uint32_t)lane; o < m; o += kWave) dst[o] = ln[i].p[o];
      dst += m;
    }
  }
}

}  // namespace fqg
