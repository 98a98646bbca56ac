// fqg_kernels.hip - HIP kernels for gfx950 (MI355X): FASTQ framing and record validation.
//
// Pipeline of one fqg_validate() call:
//   k_count_nl   one wavefront per 4 KiB chunk: count '\n', raise NUL / CR flags
//   k_scan_a/b   exclusive prefix over the chunk counts (two small launches)
//   k_frame_fast_t  one wavefront per chunk: offsets of every '\n' into line_end[], and (tiled
//                path) the byte-class checks + quality range; <7> builds the line index only
//   k_records_fast  one thread per record: lengths, statistics, queue of suspect records
//   k_validate_exact  one wavefront per record: the complete check sequence of
//                fastq_validate_entry (reference src/fastq.c:300-392) with ballots over 64-byte
//                slices of each line; statistics kept in registers and flushed once per wave
//
// Everything here is byte / integer work bounded by HBM bandwidth; there is no MFMA use.
#include "fqg_device.h"

namespace fqg {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// 0x80 in every byte of w that equals the corresponding byte of pat (exact, no borrow leaks)
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t pat) {
  const uint32_t x = w ^ pat;
  const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}
// gather the four 0x80 marks of a word into bits 0..3
__device__ __forceinline__ uint32_t mark_bits(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;  // valid in lane 0
}
// inclusive scan across the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane_id() >= d) v += o;
  }
  return v;
}

// the alphabet of reference src/fastq.c:319-322: ACGTUacgtu0123nN.
__device__ __forceinline__ bool is_base(uint32_t c) {
  // bits for '.'(46) '0'..'3'(48..51) in the low word; letters in the high word (c-64)
  constexpr uint64_t lo = (1ull << 46) | (0xFull << 48);
  constexpr uint64_t hi = (1ull << ('A' - 64)) | (1ull << ('C' - 64)) | (1ull << ('G' - 64)) |
                          (1ull << ('T' - 64)) | (1ull << ('U' - 64)) | (1ull << ('N' - 64)) |
                          (1ull << ('a' - 64)) | (1ull << ('c' - 64)) | (1ull << ('g' - 64)) |
                          (1ull << ('t' - 64)) | (1ull << ('u' - 64)) | (1ull << ('n' - 64));
  const uint64_t w = (c & 64u) ? hi : lo;
  return c < 128u && ((w >> (c & 63u)) & 1ull);
}

// ------------------------------------------------------------------------------------------
// framing: newline census per tile
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_count_nl(const uint8_t* __restrict__ img, uint64_t n,
                                                     uint32_t n_chunks,
                                                     uint32_t* __restrict__ chunk_counts,
                                                     CallState* __restrict__ cs) {
  // one wavefront per 4 KiB chunk, no workgroup-level cooperation
  const uint32_t chunk = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (chunk >= n_chunks) return;
  const uint64_t base = (uint64_t)chunk * kChunkBytes + (uint64_t)lane_id() * 16;
  uint32_t cnt = 0, nul = 0, cr = 0;
  uint4 v[kSlices];
#pragma unroll
  for (int k = 0; k < kSlices; ++k) {
    const uint64_t off = base + (uint64_t)k * (kWave * 16);
    v[k] = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
    if (off + 16 <= n) v[k] = *reinterpret_cast<const uint4*>(img + off);
    else if (off < n) {
      for (uint64_t i = off; i < n; ++i) {
        const uint32_t c = img[i];
        cnt += (c == '\n');
        nul |= (c == 0);
        cr |= (c == '\r');
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kSlices; ++k) {
    const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cnt += __popc(eq_bytes(w[j], 0x0A0A0A0Au));
      nul |= eq_bytes(w[j], 0u);
      cr |= eq_bytes(w[j], 0x0D0D0D0Du);
    }
  }
  cnt = wave_sum(cnt);
  const uint64_t anyn = __ballot(nul != 0), anyc = __ballot(cr != 0);
  if (lane_id() == 0) {
    chunk_counts[chunk] = cnt;
    const uint32_t f = (anyn ? kFlagNul : 0u) | (anyc ? kFlagCr : 0u);
    if (f) atomicOr(&cs->flags, f);
  }
}

// block-wide exclusive scan of one value per thread (kBlock threads); returns the exclusive
// prefix, *total gets the block sum.  Uses 2 barriers.
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* s_wave /*[4]*/,
                                                    uint32_t* total) {
  const uint32_t incl = wave_scan_incl(v);
  if (lane_id() == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) {
    const uint32_t t = s_wave[w];
    if (w < (int)(threadIdx.x >> 6)) before += t;
    all += t;
  }
  __syncthreads();
  *total = all;
  return before + incl - v;
}

// phase A: each workgroup scans kScanSpan tile counts (16 per thread)
// (first_block: the launch covers spans first_block .. first_block + gridDim.x - 1 - the streaming pass in parts)
__global__ __launch_bounds__(kBlock) void k_scan_a(const uint32_t* __restrict__ counts,
                                                   uint32_t n_tiles,
                                                   uint32_t* __restrict__ local_excl,
                                                   unsigned long long* __restrict__ block_sums, uint32_t first_block = 0) {
  __shared__ uint32_t s_wave[kBlock / kWave];
  const uint32_t first = (blockIdx.x + first_block) * kScanSpan + threadIdx.x * 16;
  uint32_t v[16];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    v[k] = (first + k < n_tiles) ? counts[first + k] : 0u;
    sum += v[k];
  }
  uint32_t total;
  uint32_t run = block_scan_excl(sum, s_wave, &total);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (first + k < n_tiles) local_excl[first + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x + first_block] = total;
}

// phase B: one workgroup turns the span sums into exclusive prefixes in place
// (first_block, part: spans [first_block, n_blocks) continue from what part - 1 left in the call state, and leave the
// newline count up to here as part_newlines[part])
__device__ __forceinline__ void scan_b_body(unsigned long long* __restrict__ block_sums, uint32_t n_blocks, const uint8_t* __restrict__ img,
                                            uint64_t n, CallState* __restrict__ cs, uint32_t first_block, uint32_t part,
                                            unsigned long long* s_part, unsigned long long* s_carry) {
  if (threadIdx.x == 0) *s_carry = part ? cs->part_newlines[part - 1] : 0ull;
  __syncthreads();
  for (uint32_t base = first_block; base < n_blocks; base += kBlock) {
    const uint32_t i = base + threadIdx.x;
    const unsigned long long v = (i < n_blocks) ? block_sums[i] : 0ull;
    s_part[threadIdx.x] = v;
    __syncthreads();
    // Hillis-Steele over 256 entries; cheap, runs once per call
    for (int d = 1; d < kBlock; d <<= 1) {
      unsigned long long o = (threadIdx.x >= (unsigned)d) ? s_part[threadIdx.x - d] : 0ull;
      __syncthreads();
      s_part[threadIdx.x] += o;
      __syncthreads();
    }
    const unsigned long long carry = *s_carry;
    if (i < n_blocks) block_sums[i] = carry + s_part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) *s_carry = carry + s_part[kBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    cs->n_newlines = *s_carry;
    cs->part_newlines[part < 4u ? part : 3u] = *s_carry;
    cs->last_byte_is_nl = (n > 0 && img[n - 1] == '\n') ? 1u : 0u;
  }
}
// (first_block, part: spans [first_block, n_blocks) continue from what part - 1 left in the call state, and leave the
// newline count up to here as part_newlines[part])
__global__ __launch_bounds__(kBlock) void k_scan_b(unsigned long long* __restrict__ block_sums,
                                                   uint32_t n_blocks, const uint8_t* __restrict__ img,
                                                   uint64_t n, CallState* __restrict__ cs, uint32_t first_block = 0, uint32_t part = 0) {
  __shared__ unsigned long long s_part[kBlock];
  __shared__ unsigned long long s_carry;
  scan_b_body(block_sums, n_blocks, img, n, cs, first_block, part, s_part, &s_carry);
}

// ------------------------------------------------------------------------------------------
// exact validator: one wavefront per record
// ------------------------------------------------------------------------------------------
template <class Pred>
__device__ __forceinline__ uint32_t wave_find(const uint8_t* __restrict__ p, uint32_t from,
                                              uint32_t to, Pred pred) {
  for (uint32_t b = from; b < to; b += kWave) {
    const uint32_t i = b + lane_id();
    const bool hit = (i < to) && pred((uint32_t)p[i]);
    const uint64_t m = __ballot(hit);
    if (m) return b + (uint32_t)__builtin_ctzll(m);
  }
  return to;
}

struct Line {
  const uint8_t* p;  // first byte
  uint32_t len;      // bytes before the '\n' (or before the end of the image)
  uint32_t nl;       // 1 if terminated by '\n'
};

// Length and start (relative to p) of the canonical read name of a header line, following
// fastq_get_readname (reference src/fastq.c:488-512).  cstr = C-string length of the line.
__device__ __forceinline__ uint32_t canonical_name_len(const Line& h, uint32_t cstr, int fmt, int is_pe) {
  // S = line[1 .. cstr): what strncpy(rn, &hdr[1], ...) copies
  const uint32_t L = cstr > 0 ? cstr - 1 : 0;
  if (fmt == FQG_NAME_CASAVA18) {
    const uint32_t sp = wave_find(h.p + 1, 0, L, [](uint32_t c) { return c == ' '; });
    if (sp >= 2 && h.p[1 + sp - 2] == '/') return sp - 2;
    return sp;
  }
  long len = (long)L;
  if (fmt == FQG_NAME_DEFAULT && is_pe) len--;
  return len >= 1 ? (uint32_t)(len - 1) : L;  // rn[len-1]='\0' lands outside the string when len<1
}

// compare_headers (reference src/fastq.c:543-566) on two NUL-free byte ranges
__device__ __forceinline__ bool same_names(const uint8_t* a, uint32_t na, const uint8_t* b, uint32_t nb) {
  if (nb == 0) return true;
  const uint32_t b0 = b[0];
  if (b0 == '\n' || b0 == '\r') return true;
  const uint32_t m = na < nb ? na : nb;
  uint32_t cp = m;
  for (uint32_t base = 0; base < m; base += kWave) {
    const uint32_t i = base + lane_id();
    const bool diff = (i < m) && (a[i] != b[i]);
    const uint64_t mm = __ballot(diff);
    if (mm) {
      cp = base + (uint32_t)__builtin_ctzll(mm);
      break;
    }
  }
  auto not_eol = [](uint32_t c) { return c != '\r' && c != '\n'; };
  if (wave_find(a, cp, na, not_eol) != na) return false;
  if (wave_find(b, cp, nb, not_eol) != nb) return false;
  return true;
}

struct RecOut {
  uint32_t code;
  uint64_t aux0, aux1;
  uint32_t read_len;  // strlen(seq)
  uint32_t qmin, qmax;
};

__device__ __forceinline__ void load_lines(const FrameView& f, uint64_t r, Line ln[4]) {
  // lanes 0..4 fetch line_end[4r-1 .. 4r+3]
  const int l = lane_id();
  uint64_t e = 0;
  if (l < 5) {
    const uint64_t idx = 4 * r + (uint64_t)l;
    e = (idx == 0) ? ~0ull : f.line_end[idx - 1];
  }
  uint64_t ends[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) ends[k] = __shfl(e, k, 64);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint64_t s = ends[k] + 1;  // ~0ull + 1 == 0 for the very first line
    ln[k].p = f.img + s;
    ln[k].len = (uint32_t)(ends[k + 1] - s);
    ln[k].nl = ends[k + 1] < f.nbytes ? 1u : 0u;
  }
}

// The whole check sequence for one record, executed by one wavefront (all lanes take the same
// path).  Order of checks = reference src/fastq.c:245-261 then :300-392.
__device__ __forceinline__ RecOut validate_record(const FrameView& f, uint64_t r, int is_pe, int fmt,
                                                  int space) {
  RecOut o;
  o.code = FQG_OK;
  o.aux0 = o.aux1 = 0;
  o.read_len = 0;
  o.qmin = 255;
  o.qmax = 0;
  Line ln[4];
  load_lines(f, r, ln);
  const Line &h1 = ln[0], &sq = ln[1], &h2 = ln[2], &ql = ln[3];

  // gzgets limits (src/fastq.c:249-253): a longer line would have been split by the reference
  if (!f.reframed &&
      (h1.len + h1.nl > FQG_MAX_LABEL_LENGTH - 1 || h2.len + h2.nl > FQG_MAX_LABEL_LENGTH - 1 ||
       sq.len + sq.nl > FQG_MAX_READ_LENGTH - 1 || ql.len + ql.nl > FQG_MAX_READ_LENGTH - 1)) {
    o.code = FQG_E_LINE_TOO_LONG;
    return o;
  }
  const uint32_t b0 = h1.p[0];
  // src/fastq.c:254: a 2nd/3rd/4th line that starts with NUL reads as missing
  if (sq.p[0] == 0 || h2.p[0] == 0 || ql.p[0] == 0) {
    o.code = FQG_E_TRUNCATED;
    return o;
  }
  if (b0 != '@') {
    o.code = FQG_E_HDR1_AT;
    return o;
  }
  const uint32_t h1_total = h1.len + h1.nl;
  const uint32_t b1 = h1_total > 1 ? (uint32_t)h1.p[1] : 0u;
  if (b1 == 0 || b1 == '\n' || b1 == '\r') {
    o.code = FQG_E_HDR1_SHORT;
    return o;
  }

  // ---- sequence line ----
  uint32_t slen = sq.len;
  uint32_t term = 256;  // byte that stopped the scan, 256 = the line end
  uint32_t p_inv = ~0u, p_t = ~0u, p_u = ~0u, inv_char = 0;
  for (uint32_t base = 0; base < sq.len; base += kWave) {
    const uint32_t i = base + lane_id();
    const bool in = i < sq.len;
    const uint32_t c = in ? (uint32_t)sq.p[i] : (uint32_t)'A';
    const uint64_t m_term = __ballot(in && (c == 0 || c == '\r'));
    const uint64_t below = m_term ? ((1ull << __builtin_ctzll(m_term)) - 1ull) : ~0ull;
    const uint64_t m_inv = __ballot(in && !is_base(c)) & below;
    const uint64_t m_t = __ballot(in && (c == 'T' || c == 't')) & below;
    const uint64_t m_u = __ballot(in && (c == 'U' || c == 'u')) & below;
    if (m_inv && p_inv == ~0u) {
      const int k = __builtin_ctzll(m_inv);
      p_inv = base + k;
      inv_char = __shfl(c, k, 64);
    }
    if (m_t && p_t == ~0u) p_t = base + __builtin_ctzll(m_t);
    if (m_u && p_u == ~0u) p_u = base + __builtin_ctzll(m_u);
    if (m_term) {
      const int k = __builtin_ctzll(m_term);
      slen = base + k;
      term = __shfl(c, k, 64);
      break;
    }
    if (p_inv != ~0u || (p_t != ~0u && p_u != ~0u)) break;  // outcome already decided
  }
  {
    const uint32_t p_ut = (p_t != ~0u && p_u != ~0u) ? (p_t > p_u ? p_t : p_u) : ~0u;
    if (p_inv < p_ut) {
      o.code = FQG_E_SEQ_CHAR;
      o.aux0 = inv_char;
      return o;
    }
    if (p_ut != ~0u) {
      o.code = FQG_E_SEQ_UT;
      return o;
    }
  }
  // FASTQ_ENTRY.read_len = strlen(seq) (src/fastq.c:259): up to the first NUL, '\n' included
  if (term == 0) o.read_len = slen;
  else if (term == '\r') {
    const uint32_t z = wave_find(sq.p, slen + 1, sq.len, [](uint32_t c) { return c == 0; });
    o.read_len = z < sq.len ? z : sq.len + sq.nl;
  } else o.read_len = sq.len + sq.nl;

  if (slen < 1) {
    o.code = FQG_E_LEN_SMALL;
    o.aux0 = slen;
    return o;
  }
  if (h2.p[0] != '+') {
    o.code = FQG_E_HDR2_PLUS;
    return o;
  }
  // ---- header 2 against header 1 (src/fastq.c:363-370) ----
  {
    const uint32_t h2_total = h2.len + h2.nl;
    const uint32_t c1 = h2_total > 1 ? (uint32_t)h2.p[1] : 0u;
    if (!(c1 == 0 || c1 == '\n' || c1 == '\r')) {
      auto is_nul = [](uint32_t c) { return c == 0; };
      const uint32_t z1 = wave_find(h1.p, 0, h1.len, is_nul);
      const uint32_t z2 = wave_find(h2.p, 0, h2.len, is_nul);
      const uint32_t cs1 = z1 < h1.len ? z1 : h1_total;
      const uint32_t cs2 = z2 < h2.len ? z2 : h2_total;
      const uint32_t n1 = canonical_name_len(h1, cs1, fmt, is_pe);
      const uint32_t n2 = canonical_name_len(h2, cs2, fmt, is_pe);
      if (!same_names(h1.p + 1, n1, h2.p + 1, n2)) {
        o.code = FQG_E_HDR2_DIFF;
        return o;
      }
    }
  }
  // ---- quality line ----
  uint32_t qlen = ql.len;
  uint32_t qmin = 255, qmax = 0;
  for (uint32_t base = 0; base < ql.len; base += kWave) {
    const uint32_t i = base + lane_id();
    const bool in = i < ql.len;
    const uint32_t c = in ? (uint32_t)ql.p[i] : 0xFFu;
    const uint64_t m_term = __ballot(in && (c == 0 || c == '\r'));
    const uint32_t stop = m_term ? (uint32_t)__builtin_ctzll(m_term) : 64u;
    const bool use = in && (uint32_t)lane_id() < stop;
    qmin = use && c < qmin ? c : qmin;
    qmax = use && c > qmax ? c : qmax;
    if (m_term) {
      qlen = base + stop;
      break;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
    qmin = a < qmin ? a : qmin;
    qmax = b > qmax ? b : qmax;
  }
  o.qmin = qmin;
  o.qmax = qmax;
  if (space == FQG_SPACE_SEQ && qlen != slen) {
    o.code = FQG_E_QLEN;
    o.aux0 = slen;
    o.aux1 = qlen;
    return o;
  }
  if (space == FQG_SPACE_COLOUR && !(qlen == slen - 1 || qlen == slen)) {
    o.code = FQG_E_QLEN_CS;
    o.aux0 = slen;
    o.aux1 = qlen;
    return o;
  }
  return o;
}

// Persistent wavefronts: wave w handles records w, w + W, w + 2W, ...  Statistics stay in
// registers (min/max, and a run-length cache for the length histogram) until the wave is done.
__global__ __launch_bounds__(kBlock) void k_validate_exact(FrameView f, int is_pe, int fmt, int space,
                                                           uint32_t weight, AccState* __restrict__ acc,
                                                           unsigned long long* __restrict__ hist,
                                                           CallState* __restrict__ cs,
                                                           uint64_t explain_record,
                                                           const unsigned long long* __restrict__ list,
                                                           const unsigned long long* __restrict__ list_count) {
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (explain_record != kNoRecord) {
    if (wave != 0) return;
    const RecOut o = validate_record(f, explain_record, is_pe, fmt, space);
    if (lane_id() == 0) {
      cs->aux0 = o.aux0;
      cs->aux1 = o.aux1;
    }
    return;
  }
  uint64_t n_ok = 0, min_rl = ~0ull, max_rl = 0;
  uint32_t qmin = 255, qmax = 0;
  uint32_t run_len = 0;
  uint64_t run_cnt = 0;
  unsigned long long key = ~0ull;
  // (an overflowing queue is noticed by the host, which then re-runs over every record)
  const uint64_t n_todo = list ? (uint64_t)*list_count : f.n_records;
  for (uint64_t i = wave; i < n_todo; i += n_waves) {
    const uint64_t r = list ? (uint64_t)list[i] : i;
    const RecOut o = validate_record(f, r, is_pe, fmt, space);
    if (o.code != FQG_OK) {
      const unsigned long long k = (r << 8) | o.code;
      key = k < key ? k : key;
      continue;
    }
    ++n_ok;
    min_rl = o.read_len < min_rl ? o.read_len : min_rl;
    max_rl = o.read_len > max_rl ? o.read_len : max_rl;
    qmin = o.qmin < qmin ? o.qmin : qmin;
    qmax = o.qmax > qmax ? o.qmax : qmax;
    if (o.read_len == run_len) ++run_cnt;
    else {
      if (run_cnt && acc && lane_id() == 0) atomicAdd(&hist[run_len], run_cnt * weight);
      run_len = o.read_len;
      run_cnt = 1;
    }
  }
  if (lane_id() == 0) {
    if (key != ~0ull) atomicMin(&cs->first_key, key);
    if (acc && n_ok) {
      if (run_cnt) atomicAdd(&hist[run_len], run_cnt * weight);
      atomicAdd(&acc->num_rds, n_ok * weight);
      atomicMin(&acc->min_rl, min_rl);
      atomicMax(&acc->max_rl, max_rl);
      if (qmin <= qmax) {
        atomicMin(&acc->min_qbyte, qmin);
        atomicMax(&acc->max_qbyte, qmax);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// tiled fast path
//
// For images without NUL / CR bytes the statistics do not need per-record work on the bytes:
//   - read lengths come from the newline positions alone (k_records_fast),
//   - the quality range is the min/max over every byte of every 4th line,
// and a record is certainly valid when its header starts with '@' and is longer than 1, its
// bases are all in "ACGTN", its third line is exactly "+\n" and the 2nd and 4th line have the
// same non-zero length.  k_frame_fast checks the byte-level conditions while it builds the line
// index, 16 bytes per lane with SWAR compares, and marks every record it cannot vouch for in a
// bitmap; k_records_fast checks the lengths and queues marked records for the exact
// wave-per-record validator, which alone decides error codes.  Over-marking is harmless.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t prefix_xor16(uint32_t x) {
  x ^= x << 1;
  x ^= x << 2;
  x ^= x << 4;
  x ^= x << 8;
  return x & 0xFFFFu;
}

// 0x80 in every byte of w that is NOT one of A C G T N.  The five letters have distinct low
// three bits (1,3,7,4,6), so an 8-entry byte table indexed by (c & 7) names the only letter each
// byte could be; entries 0,2,5 hold 0xFF which no byte with those low bits can equal.
__device__ __forceinline__ uint32_t not_acgtn(uint32_t w) {
  // table bytes, index 0..7: FF 'A' FF 'C' 'T' FF 'N' 'G'
  const uint32_t lut_lo = 0x43FF41FFu;  // idx 3..0
  const uint32_t lut_hi = 0x474EFF54u;  // idx 7..4
  const uint32_t want = __builtin_amdgcn_perm(lut_hi, lut_lo, w & 0x07070707u);
  const uint32_t x = want ^ w;
  return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}

// spread the low 4 bits of m into four 0x00/0xFF bytes
__device__ __forceinline__ uint32_t nibble_to_bytes(uint32_t m) {
  return (((m & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

__device__ __forceinline__ void mark_suspect(const SuspectMap& sm, uint64_t record) {
  if (record < sm.cap) atomicOr(&sm.bits[record >> 5], 1u << (record & 31u));
  else atomicOr(sm.flags, kFlagSuspectOverflow);
}

// Generic byte-at-a-time version of the per-piece work, used for the one or two tiles at the
// end of the image where look-ahead bytes or lines beyond the last complete record exist.
__device__ __forceinline__ void piece_generic(const uint8_t* __restrict__ img, uint64_t n, uint64_t off,
                                              uint64_t line, uint64_t limit, const SuspectMap& suspect,
                                              uint32_t& qmin, uint32_t& qmax) {
  for (int j = 0; j < 16 && off + j < n; ++j) {
    const uint64_t pos = off + j;
    const uint32_t c = img[pos];
    if (line < limit) {
      const uint32_t t = (uint32_t)line & 3u;
      const bool start = pos == 0 || img[pos - 1] == '\n';
      const bool second = !start && (pos == 1 || img[pos - 2] == '\n');
      bool bad = false;
      if (t == 0) bad = (start && c != '@') || (second && c == '\n');
      else if (t == 2) bad = (start && c != '+') || (second && c != '\n');
      else if (t == 1) bad = c != '\n' && !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N');
      else if (c != '\n') {
        qmin = c < qmin ? c : qmin;
        qmax = c > qmax ? c : qmax;
      }
      if (bad) mark_suspect(suspect, line >> 2);
    }
    if (c == '\n') ++line;
  }
}

// DPP lane permutes (gfx9 family): row_shr:n = 0x110+n, row_bcast15 = 0x142, row_bcast31 = 0x143,
// wave_shl1 = 0x130.  Masked-off or out-of-range lanes contribute 0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, true);
}
// inclusive scan over the 64 lanes in 7 DPP adds (no LDS traffic)
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t v) {
  uint32_t x = v + dpp0<0x111, 0xf, 0xf>(v);
  x += dpp0<0x112, 0xf, 0xf>(v);
  x += dpp0<0x113, 0xf, 0xf>(v);
  x += dpp0<0x114, 0xf, 0xe>(x);
  x += dpp0<0x118, 0xf, 0xc>(x);
  x += dpp0<0x142, 0xa, 0xf>(x);
  x += dpp0<0x143, 0xc, 0xf>(x);
  return x;
}

__device__ __forceinline__ uint32_t nl_mask16(const uint4& v) {
  return mark_bits(eq_bytes(v.x, 0x0A0A0A0Au)) | (mark_bits(eq_bytes(v.y, 0x0A0A0A0Au)) << 4) |
         (mark_bits(eq_bytes(v.z, 0x0A0A0A0Au)) << 8) | (mark_bits(eq_bytes(v.w, 0x0A0A0A0Au)) << 12);
}

struct QRange {
  uint32_t mn_e, mn_o, mx_e, mx_o;  // packed u16 pairs: even / odd bytes
};

// exact min / max over the bytes of a 16-byte piece selected by the 16-bit mask qm
__device__ __forceinline__ void qrange_accum(const uint4& v, uint32_t qm, QRange& q) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t bm = nibble_to_bytes(qm >> (4 * k));
    const uint32_t lo = w[k] | ~bm, hi = w[k] & bm;
    q.mn_e = pk_min_u16(q.mn_e, __builtin_amdgcn_perm(0u, lo, 0x0C020C00u));
    q.mn_o = pk_min_u16(q.mn_o, __builtin_amdgcn_perm(0u, lo, 0x0C030C01u));
    q.mx_e = pk_max_u16(q.mx_e, __builtin_amdgcn_perm(0u, hi, 0x0C020C00u));
    q.mx_o = pk_max_u16(q.mx_o, __builtin_amdgcn_perm(0u, hi, 0x0C030C01u));
  }
}

// The SWAR checks on one 16-byte piece whose bytes, two look-ahead bytes and lines all exist.
// ABL is an ablation mask for tools/kbench (product code instantiates 0).
template <uint32_t ABL>
__device__ __forceinline__ void piece_fast(const uint4& v, uint32_t nl, uint64_t line0, uint64_t off,
                                           uint32_t nxt, const SuspectMap& suspect, QRange& q) {
  // type (line index mod 4) of every byte: 2-bit running count of the newlines before it
  const uint32_t e = (nl << 1) & 0xFFFFu;
  const uint32_t P = prefix_xor16(e);       // count bit 0
  const uint32_t Q = prefix_xor16(e & ~P);  // count bit 1 (carry when bit 0 wraps)
  const uint32_t t0 = (uint32_t)line0 & 3u;
  const uint32_t a0 = (t0 & 1u) ? 0xFFFFu : 0u, a1 = (t0 & 2u) ? 0xFFFFu : 0u;
  const uint32_t L = P ^ a0;
  const uint32_t H = Q ^ a1 ^ (P & a0);
  const uint32_t M1 = ~H & L & 0xFFFFu, M3 = H & L;
  uint32_t bad = 0, badstart = 0;
  if (!(ABL & 2u)) {
    // bases: anything outside ACGTN on a sequence line
    const uint32_t inv = mark_bits(not_acgtn(v.x)) | (mark_bits(not_acgtn(v.y)) << 4) |
                         (mark_bits(not_acgtn(v.z)) << 8) | (mark_bits(not_acgtn(v.w)) << 12);
    bad = inv & M1 & ~nl;
  }
  if (!(ABL & 4u)) {
    // first two bytes of header lines, anchored at the newline in front of them
    uint32_t cand = nl & (M1 | M3);  // newline ends a sequence / quality line
    if (cand) {
      const uint64_t q0 = (uint64_t)v.x | ((uint64_t)v.y << 32), q1 = (uint64_t)v.z | ((uint64_t)v.w << 32),
                     q2 = nxt;
      // newline bits of this piece followed by those of the next 2 bytes
      const uint32_t nl18 = nl | ((mark_bits(eq_bytes(nxt, 0x0A0A0A0Au)) & 3u) << 16);
      do {  // one trip for ordinary data: a 16-byte piece rarely holds two such newlines
        const int j = __builtin_ctz(cand);
        cand &= cand - 1;
        const int p = j + 1;
        const uint64_t lo = p < 8 ? q0 : (p < 16 ? q1 : q2);
        const uint32_t c1 = (uint32_t)(lo >> ((p & 7) * 8)) & 0xFFu;
        const bool c2_nl = (nl18 >> (p + 1)) & 1u;
        const bool after_qual = (M3 >> j) & 1u;  // next line is a header 1, else a header 2
        const bool ok = after_qual ? (c1 == '@' && !c2_nl) : (c1 == '+' && c2_nl);
        badstart |= ok ? 0u : (1u << j);
      } while (cand);
    }
    if (off == 0 && ((v.x & 0xFFu) != '@' || ((v.x >> 8) & 0xFFu) == '\n')) mark_suspect(suspect, 0);
  }
  if (bad | badstart) {
    uint32_t m = bad;
    while (m) {
      const int j = __builtin_ctz(m);
      m &= m - 1;
      mark_suspect(suspect, (line0 + __popc(nl & ((1u << j) - 1u))) >> 2);
    }
    m = badstart;
    while (m) {
      const int j = __builtin_ctz(m);
      m &= m - 1;
      mark_suspect(suspect, (line0 + __popc(nl & ((1u << j) - 1u)) + 1) >> 2);
    }
  }
  if (!(ABL & 1u)) qrange_accum(v, M3 & ~nl, q);  // quality range: bytes of 4th lines, newline excluded
}

// Work decomposition: ONE WAVEFRONT owns a contiguous 4 KiB chunk per step and walks it as 4
// slices of 1 KiB (64 lanes x 16 B, fully coalesced).  There is no workgroup-level cooperation
// at all - no LDS, no barriers: the newline ranks inside a chunk come from two packed DPP scans,
// the rank of the chunk itself from the prefix over the per-chunk counts.  Wavefronts are
// persistent (chunk = wave, wave + W, ...), and the loads of the next chunk are issued before
// the current one is processed.
template <uint32_t ABL>
__global__ __launch_bounds__(kBlock) void k_frame_fast_t(const uint8_t* __restrict__ img, uint64_t n,
                                                         uint32_t n_chunks,
                                                         const uint32_t* __restrict__ chunk_local,
                                                         const unsigned long long* __restrict__ span_excl,
                                                         uint64_t* __restrict__ line_end, uint64_t line_cap,
                                                         uint64_t limit, SuspectMap suspect,
                                                         CallState* __restrict__ cs,
                                                         const uint32_t* __restrict__ todo,
                                                         const uint32_t* __restrict__ todo_count) {
  static_assert(kSlices == 4, "packed scans below assume 4 slices per chunk");
  const int lane = lane_id();
  const uint32_t n_waves = gridDim.x * (kBlock / kWave);
  const uint32_t wave0 = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  // todo != null: only the listed chunks (the streaming path's redo list), in any order
  const uint32_t n_items = todo ? *todo_count : n_chunks;
  QRange q{0x00FF00FFu, 0x00FF00FFu, 0u, 0u};
  uint32_t gq_min = 255, gq_max = 0;  // from the generic tail path
  uint4 vn[kSlices];
  auto load_chunk = [&](uint32_t c, uint4 (&dst)[kSlices]) {
    const uint64_t wb = (uint64_t)c * kChunkBytes + (uint64_t)lane * 16;
#pragma unroll
    for (int k = 0; k < kSlices; ++k) {
      const uint64_t off = wb + (uint64_t)k * (kWave * 16);
      dst[k] = make_uint4(0, 0, 0, 0);
      if (off + 16 <= n) dst[k] = *reinterpret_cast<const uint4*>(img + off);
    }
  };
  if (wave0 < n_items) load_chunk(todo ? todo[wave0] : wave0, vn);
  for (uint32_t item = wave0; item < n_items; item += n_waves) {
    const uint32_t chunk = todo ? todo[item] : item;
    const uint64_t rank0 = span_excl[chunk / kScanSpan] + chunk_local[chunk];
    const uint64_t cbase = (uint64_t)chunk * kChunkBytes;
    const uint64_t wbase = cbase + (uint64_t)lane * 16;
    uint4 v[kSlices];
    uint32_t nl[kSlices];
#pragma unroll
    for (int k = 0; k < kSlices; ++k) v[k] = vn[k];
    if (item + n_waves < n_items) load_chunk(todo ? todo[item + n_waves] : item + n_waves, vn);
#pragma unroll
    for (int k = 0; k < kSlices; ++k) {
      const uint64_t off = wbase + (uint64_t)k * (kWave * 16);
      nl[k] = nl_mask16(v[k]);
      if (off + 16 > n) {
        nl[k] = 0;
        for (uint64_t i = off; i < n; ++i) nl[k] |= (img[i] == '\n') ? (1u << (i - off)) : 0u;
      }
    }
    const uint32_t c01 = __popc(nl[0]) | (__popc(nl[1]) << 16), c23 = __popc(nl[2]) | (__popc(nl[3]) << 16);
    const uint32_t s01 = wave_scan_incl_dpp(c01), s23 = wave_scan_incl_dpp(c23);
    const uint32_t t01 = __builtin_amdgcn_readlane(s01, 63), t23 = __builtin_amdgcn_readlane(s23, 63);
    const uint32_t total = (t01 & 0xFFFFu) + (t01 >> 16) + (t23 & 0xFFFFu) + (t23 >> 16);
    const bool interior = cbase + kChunkBytes + 4 <= n && rank0 + total + 1 <= limit;
    // exclusive rank of this lane's slice k inside the chunk
    uint32_t ex[kSlices];
    ex[0] = (s01 & 0xFFFFu) - (c01 & 0xFFFFu);
    ex[1] = (t01 & 0xFFFFu) + (s01 >> 16) - (c01 >> 16);
    ex[2] = (t01 & 0xFFFFu) + (t01 >> 16) + (s23 & 0xFFFFu) - (c23 & 0xFFFFu);
    ex[3] = (t01 & 0xFFFFu) + (t01 >> 16) + (t23 & 0xFFFFu) + (s23 >> 16) - (c23 >> 16);
    uint32_t tail4 = 0;  // the 4 bytes after this chunk, for lane 63 of the last slice
    if (interior && lane == 63) tail4 = *reinterpret_cast<const uint32_t*>(img + cbase + kChunkBytes);
#pragma unroll
    for (int k = 0; k < kSlices; ++k) {
      const uint64_t off = wbase + (uint64_t)k * (kWave * 16);
      const uint64_t line0 = rank0 + ex[k];
      if (!(ABL & 8u)) {
        uint32_t m = nl[k];
        if (m && line0 + 16 <= line_cap) {  // (the host notices an undersized index by the line count)
          uint64_t* dst = line_end + line0;
          dst[0] = off + __builtin_ctz(m);
          m &= m - 1;
          if (m) {
            dst[1] = off + __builtin_ctz(m);
            m &= m - 1;
            for (int r = 2; m; ++r, m &= m - 1) dst[r] = off + __builtin_ctz(m);
          }
        }
      }
      if ((ABL & 7u) == 7u) continue;  // line index only (exact path)
      if (interior) {
        uint32_t nxt = dpp0<0x130, 0xf, 0xf>(v[k].x);  // lane i <- lane i+1
        const uint32_t first_next = (uint32_t)__builtin_amdgcn_readlane(v[k + 1 < kSlices ? k + 1 : k].x, 0);
        if (lane == 63) nxt = k + 1 < kSlices ? first_next : tail4;
        piece_fast<ABL>(v[k], nl[k], line0, off, nxt, suspect, q);
      } else if (off < n) {
        piece_generic(img, n, off, line0, limit, suspect, gq_min, gq_max);
      }
    }
  }
  if (!(ABL & 8u) && blockIdx.x == 0 && threadIdx.x == 0 && n > 0 && !cs->last_byte_is_nl && cs->n_newlines < line_cap)
    line_end[cs->n_newlines] = n;
  if ((ABL & 7u) == 7u) return;
  // ---- fold the quality range: lanes -> wave -> device ----
  uint32_t qmin = pk_min_u16(q.mn_e, q.mn_o), qmax = pk_max_u16(q.mx_e, q.mx_o);
  qmin = (qmin & 0xFFFFu) < (qmin >> 16) ? (qmin & 0xFFFFu) : (qmin >> 16);
  qmax = (qmax & 0xFFFFu) > (qmax >> 16) ? (qmax & 0xFFFFu) : (qmax >> 16);
  qmin = gq_min < qmin ? gq_min : qmin;
  qmax = gq_max > qmax ? gq_max : qmax;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
    qmin = a < qmin ? a : qmin;
    qmax = b > qmax ? b : qmax;
  }
  if (lane == 0 && qmin <= qmax) {
    // plain reads first: after the first few wavefronts nobody improves the range any more
    if (qmin < cs->qmin_byte) atomicMin(&cs->qmin_byte, qmin);
    if (qmax > cs->qmax_byte) atomicMax(&cs->qmax_byte, qmax);
  }
}

// One thread per record: lengths from the line index, statistics, and the list of records the
// exact validator has to look at.
constexpr int kHistLds = 4096;
__global__ __launch_bounds__(kBlock) void k_records_fast(FrameView f, int space, uint32_t weight,
                                                         SuspectMap suspect,
                                                         unsigned long long* __restrict__ list,
                                                         unsigned long long list_cap,
                                                         unsigned long long* __restrict__ list_count,
                                                         AccState* __restrict__ acc,
                                                         unsigned long long* __restrict__ hist,
                                                         const CallState* __restrict__ cs) {
  __shared__ uint32_t s_hist[kHistLds];
  __shared__ unsigned long long s_red[3][kBlock / kWave];
  for (int i = threadIdx.x; i < kHistLds; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  unsigned long long n_ok = 0, min_rl = ~0ull, max_rl = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t r0 = (uint64_t)blockIdx.x * kBlock; r0 < f.n_records; r0 += stride) {
    const uint64_t r = r0 + threadIdx.x;
    if (r < f.n_records) {
      uint64_t e[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const uint64_t idx = 4 * r + k;
        e[k] = idx == 0 ? ~0ull : f.line_end[idx - 1];
      }
      const uint64_t l0 = e[1] - e[0] - 1, l1 = e[2] - e[1] - 1, l2 = e[3] - e[2] - 1, l3 = e[4] - e[3] - 1;
      const uint32_t has_nl = e[4] < f.nbytes ? 1u : 0u;  // only the very last line can lack it
      bool sus = r < suspect.cap ? ((suspect.bits[r >> 5] >> (r & 31u)) & 1u) : true;
      sus |= l1 < 1 || l1 != l3 || space != FQG_SPACE_SEQ;
      sus |= l0 + 1 > FQG_MAX_LABEL_LENGTH - 1 || l2 + 1 > FQG_MAX_LABEL_LENGTH - 1 ||
             l1 + 1 > FQG_MAX_READ_LENGTH - 1 || l3 + has_nl > FQG_MAX_READ_LENGTH - 1;
      if (sus) {
        const unsigned long long at = atomicAdd(list_count, 1ull);
        if (at < list_cap) list[at] = r;
      }
      if (acc && l1 + 1 <= FQG_MAX_READ_LENGTH - 1) {
        const uint64_t rl = l1 + 1;  // strlen(seq): the sequence line always ends in '\n' here
        ++n_ok;
        min_rl = rl < min_rl ? rl : min_rl;
        max_rl = rl > max_rl ? rl : max_rl;
        if (rl < (uint64_t)kHistLds) atomicAdd(&s_hist[rl], 1u);
        else atomicAdd(&hist[rl], (unsigned long long)weight);
      }
    }
  }
  if (!acc) return;
  if (blockIdx.x == 0 && threadIdx.x == 0 && cs->qmin_byte <= cs->qmax_byte) {
    // quality range found by k_frame_fast for this image
    atomicMin(&acc->min_qbyte, cs->qmin_byte);
    atomicMax(&acc->max_qbyte, cs->qmax_byte);
  }
  // block reduction of the scalars
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    n_ok += __shfl_down(n_ok, d, 64);
    const unsigned long long a = __shfl_down(min_rl, d, 64), b = __shfl_down(max_rl, d, 64);
    min_rl = a < min_rl ? a : min_rl;
    max_rl = b > max_rl ? b : max_rl;
  }
  if (lane_id() == 0) {
    s_red[0][threadIdx.x >> 6] = n_ok;
    s_red[1][threadIdx.x >> 6] = min_rl;
    s_red[2][threadIdx.x >> 6] = max_rl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / kWave; ++w) {
      n_ok += s_red[0][w];
      min_rl = s_red[1][w] < min_rl ? s_red[1][w] : min_rl;
      max_rl = s_red[2][w] > max_rl ? s_red[2][w] : max_rl;
    }
    if (n_ok) {
      atomicAdd(&acc->num_rds, n_ok * weight);
      if (min_rl < acc->min_rl) atomicMin(&acc->min_rl, min_rl);
      if (max_rl > acc->max_rl) atomicMax(&acc->max_rl, max_rl);
    }
  }
  for (int i = threadIdx.x; i < kHistLds; i += kBlock) {
    const uint32_t c = s_hist[i];
    if (c) atomicAdd(&hist[i], (unsigned long long)c * weight);
  }
}

// first record whose first line starts with a NUL byte (src/fastq.c:250): only launched for
// images that contain NUL bytes at all
// also_lines (images that are only framed: nobody else looks at the records): a sequence, second header or quality line
// that starts with NUL is an empty string in the reference's buffer - "file truncated" at that record (src/fastq.c:254)
__global__ __launch_bounds__(kBlock) void k_find_stop(FrameView f, CallState* __restrict__ cs, int also_lines) {
  const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= f.n_records) return;
  const uint64_t s = r == 0 ? 0 : f.line_end[4 * r - 1] + 1;
  if (f.img[s] == 0) atomicMin(&cs->stop_record, (unsigned long long)r);
  else if (also_lines) {
    bool cut = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const uint64_t b = f.line_end[4 * r + i] + 1;  // (an unterminated last line is line 4: b < nbytes here)
      cut |= b < f.nbytes && f.img[b] == 0;
    }
    if (cut) atomicMin(&cs->trunc_record, (unsigned long long)r);
  }
}

// lines [first, f.n_lines) against the gzgets limits of the reference (src/fastq.c:249-253): the record of the first
// line that gzgets would not return whole.  For images nobody validates (FQG_VALIDATE_FRAME_ONLY) and for the lines of
// an incomplete last record, which the record-wise checks never see.
__global__ __launch_bounds__(kBlock) void k_overlong(FrameView f, uint64_t first, CallState* __restrict__ cs) {
  const uint64_t i = first + (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool bad = false;
  if (i < f.n_lines) {
    const uint64_t e = f.line_end[i], s = i == 0 ? 0 : f.line_end[i - 1] + 1;
    const uint64_t total = e - s + (e < f.nbytes ? 1u : 0u);
    bad = total > ((i & 1u) ? (uint64_t)FQG_MAX_READ_LENGTH - 1 : (uint64_t)FQG_MAX_LABEL_LENGTH - 1);
  }
  const unsigned long long m = __ballot(bad);
  if (m && lane_id() == __builtin_ctzll(m)) {
    const unsigned long long key = ((unsigned long long)(i >> 2) << 8) | (unsigned long long)FQG_E_LINE_TOO_LONG;
    atomicMin(&cs->first_key, key);
  }
}

// record descriptors (FASTQ_ENTRY geometry) for a range of records
__global__ __launch_bounds__(kBlock) void k_records(FrameView f, uint64_t first, uint64_t count,
                                                    fqg_record* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const uint64_t r = first + i;
  uint64_t e[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const uint64_t idx = 4 * r + k;
    e[k] = idx == 0 ? ~0ull : f.line_end[idx - 1];
  }
  fqg_record d;
  d.offset = e[0] + 1;
  uint32_t len[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint64_t s = e[k] + 1;
    len[k] = (uint32_t)(e[k + 1] - s) + (e[k + 1] < f.nbytes ? 1u : 0u);
  }
  d.hdr1_len = len[0];
  d.seq_len = len[1];
  d.hdr2_len = len[2];
  d.qual_len = len[3];
  // strlen(seq): stop at the first NUL inside the line
  uint32_t rl = len[1];
  const uint8_t* p = f.img + e[1] + 1;
  for (uint32_t k = 0; k < len[1]; ++k)
    if (p[k] == 0) {
      rl = k;
      break;
    }
  d.read_len = rl;
  d.reserved = 0;
  out[i] = d;
}

// ------------------------------------------------------------------------------------------
// synthetic FASTQ (bench / tests): fixed-geometry records, every byte a pure function of
// (seed, record index, offset in record)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

// header: "@SYN001:1:FC1:<lane>:<tile 4d>:<x 1d>:<index 10d> <mate>:N:0:ACGT\n" = 45 bytes
constexpr int kSynthHdr = 45;
__device__ __forceinline__ uint8_t synth_header_byte(uint64_t idx, uint32_t o, int mate) {
  const char* a = "@SYN001:1:FC1:";  // 14
  if (o < 14) return (uint8_t)a[o];
  if (o == 14) return (uint8_t)('1' + (idx / 40000000ull) % 8);
  if (o == 15 || o == 20 || o == 22) return ':';
  if (o >= 16 && o < 20) {
    uint32_t t = (uint32_t)((idx / 10000ull) % 10000ull);
    const uint32_t d[4] = {t / 1000, (t / 100) % 10, (t / 10) % 10, t % 10};
    return (uint8_t)('0' + d[o - 16]);
  }
  if (o == 21) return (uint8_t)('0' + idx % 10);
  if (o >= 23 && o < 33) {
    uint64_t v = idx;
    for (uint32_t k = 32; k > o; --k) v /= 10;
    return (uint8_t)('0' + v % 10);
  }
  const char* z = " 1:N:0:ACGT\n";  // 12: offsets 33..44
  if (o == 34) return (uint8_t)('0' + mate);
  return (uint8_t)z[o - 33];
}

__global__ __launch_bounds__(kBlock) void k_synth(uint8_t* __restrict__ out, uint64_t n_records,
                                                  uint32_t read_len, uint64_t first_index,
                                                  uint64_t seed, int mate) {
  const uint64_t R = (uint64_t)kSynthHdr + 2ull * (read_len + 1) + 2;
  const uint64_t total = n_records * R;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock * 16;
  for (uint64_t p0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 16; p0 < total; p0 += stride) {
    uint8_t b[16];
    uint64_t rec = p0 / R;
    uint32_t o = (uint32_t)(p0 - rec * R);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t idx = first_index + rec;
      uint8_t c;
      if (o < (uint32_t)kSynthHdr) c = synth_header_byte(idx, o, mate);
      else if (o < kSynthHdr + read_len) {
        const uint64_t h = mix64(seed ^ (idx * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(o - kSynthHdr) << 1));
        c = ((h >> 8) & 1023u) == 0 ? 'N' : (uint8_t)"ACGT"[h & 3u];
      } else if (o == kSynthHdr + read_len) c = '\n';
      else if (o == kSynthHdr + read_len + 1) c = '+';
      else if (o == kSynthHdr + read_len + 2) c = '\n';
      else if (o < kSynthHdr + 2 * read_len + 3) {
        const uint64_t h = mix64(seed ^ (idx * 0xD1B54A32D192ED03ull) ^ (((uint64_t)(o - kSynthHdr - read_len - 3) << 1) | 1ull));
        c = (uint8_t)(33 + 2 + (uint32_t)((h >> 11) % 39u));
      } else c = '\n';
      b[j] = c;
      if (++o == R) {
        o = 0;
        ++rec;
      }
    }
    if (p0 + 16 <= total) {
      uint4 v;
      v.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
      v.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
      v.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
      v.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((uint32_t)b[15] << 24);
      *reinterpret_cast<uint4*>(out + p0) = v;
    } else {
      for (uint64_t i = p0; i < total; ++i) out[i] = b[i - p0];
    }
  }
}

}  // namespace fqg
