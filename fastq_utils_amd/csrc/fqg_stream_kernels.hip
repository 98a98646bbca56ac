// fqg_stream_kernels.hip - single-pass ("streaming") framing + byte-class checks for gfx950.
//
// The two-pass path of fqg_kernels.hip reads the image twice: a newline census gives every 4 KiB
// chunk its global line rank, then k_frame_fast_t uses the rank for the line index and for the
// line type (index mod 4) that the byte-class checks depend on.  Here the image is read ONCE:
//
//   k_stream_boot   quality range of the first records (line types are known at the image start)
//   k_stream_pass1  one wavefront per 4 KiB chunk, no dependence on any other chunk:
//                     - newline mask, NUL / CR / high-byte detection (SWAR on bytes < 0x80, byte
//                       marks compacted to bit masks with v_dot4_u32_u8)
//                     - the line type of the chunk's first byte is SPECULATED from the first
//                       one-character line of the chunk (in FASTQ that is the "+" line, type 2)
//                     - with that type: bases outside ACGTN on sequence lines -> suspect byte
//                       positions are queued; quality bytes are only tested against the boot
//                       range, and a chunk with bytes outside it computes its exact min / max
//                     - every '\n' is staged as a 16-bit entry (offset in chunk, class of the byte
//                       after it, "the byte after that is '\n'") through a per-wave LDS table, so
//                       that the global stores are contiguous
//   k_scan_a/b      prefix over the per-chunk newline counts (fqg_kernels.hip)
//   k_stream_pass2  one wavefront per chunk, now with the true rank: staged entries -> line_end[],
//                   header-start checks ('@', not empty; "+\n") by line type, verification of the
//                   speculation (a wrong or missing one sends the chunk to the redo list, where
//                   k_frame_fast_t repeats the checks with the true type), quality range merge
//   k_stream_queue  queued suspect positions -> record bits (binary search in line_end[])
//
// Nothing here decides an error code: as in the two-pass path, records the checks cannot vouch for
// are marked in the suspect bitmap and k_validate_exact alone decides.  Over-marking is harmless;
// what must be exact are the line index, the quality range (hull of the boot range, of verified
// per-chunk ranges and of redone chunks) and the image flags that force the two-pass / exact path.
#include "fqg_device.h"

namespace fqg {

constexpr uint32_t kH = 0x80808080u;

// four words of 0x80-per-byte marks -> 16-bit mask (bit i = byte i of the 16)
__device__ __forceinline__ uint32_t pack_marks16(uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3) {
  uint32_t lo = __builtin_amdgcn_udot4(m0, 0x08040201u, 0u, false);
  lo = __builtin_amdgcn_udot4(m1, 0x80402010u, lo, false);
  uint32_t hi = __builtin_amdgcn_udot4(m2, 0x08040201u, 0u, false);
  hi = __builtin_amdgcn_udot4(m3, 0x80402010u, hi, false);
  return (lo >> 7) | (hi << 1);
}

// Newline marks of a word whose bytes are all < 0x80, with the control-character test as ONE three-input boolean
// instruction per word: `bad` collects bit 7 of every byte that is < 0x20 and no '\n'.  (x = byte ^ 0x0A keeps a byte on its side of 0x20; t = 0x80 - x has bit 7
// where x == 0; t + 0x1F = 0x9F - x has bit 7 where x < 0x20.  No byte carries into its neighbour: t is 0x01..0x80.)
// What an instruction costs on gfx950 decides the form (tools/kbench/valubench.hip, profiles/r04*_valubench.txt): a
// two-operand VALU instruction on registers or a literal - and v_bitop3_b32 - issues in 2 cycles per wavefront, a
// three-operand one (v_or3, v_and_or, v_perm, v_dot4, v_xad, v_lshl_or ..) or one that reads an SGPR in 4.
__device__ __forceinline__ uint32_t nl_marks7(uint32_t w, uint32_t& bad) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  const uint32_t t = kH - x;
  const uint32_t c = t + 0x1F1F1F1Fu;
  bad = __builtin_amdgcn_bitop3_b32(bad, c, t, 0xF4);  // bad | (c & ~t): truth table of A | (B & ~C) with A = 0xF0, B = 0xCC, C = 0xAA
  return t & kH;
}
__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE); }

// a wave-uniform value in a VECTOR register: v_sub / v_add on two vector registers issue in 2 cycles, with an SGPR
// operand in 4 (the compiler would keep such a value in an SGPR)
__device__ __forceinline__ uint32_t in_vgpr(uint32_t x) {
  uint32_t v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
  return v;
}

// 0x80 in every byte (< 0x80) that is not one of A C G T N
__device__ __forceinline__ uint32_t not_acgtn7(uint32_t w) {
  // byte table indexed by (c & 7): 7F 'A' 7F 'C' 'T' 7F 'N' 'G'; 0x7F never matches (its low bits are 7)
  const uint32_t want = __builtin_amdgcn_perm(0x474E7F54u, 0x437F417Fu, w & 0x07070707u);
  return ((w ^ want) + 0x7F7F7F7Fu) & kH;
}

// 0x80 in every byte (< 0x80) inside [lo, hi]; lob = lo * 0x01010101, hihb = (hi | 0x80) * 0x01010101
__device__ __forceinline__ uint32_t in_range7(uint32_t w, uint32_t lob, uint32_t hihb, uint32_t kh = kH) {
  return ((w | kH) - lob) & (hihb - w) & kh;
}
// The same with the lower bound as an ADDEND: lo_add = (0x80 - lo) * 0x01010101 (lo <= 127).  A byte b < 0x80 plus
// 0x80 - lo stays below 0x100 - no byte carries into its neighbour - and has bit 7 exactly when b >= lo: one
// instruction where `(w | 0x80..) - lo..` takes two.
__device__ __forceinline__ uint32_t in_range7a(uint32_t w, uint32_t lo_add, uint32_t hihb, uint32_t kh) {
  return (w + lo_add) & (hihb - w) & kh;
}

// ------------------------------------------------------------------------------------------
// quality range of the complete records inside the first `span` bytes (span: multiple of 256,
// span <= image size).  One workgroup.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kBootMax = 64u << 10;  // span <= kBootMax (host: kStreamBootBytes)
constexpr int kBootBlock = 1024;          // 16 wavefronts, 4 KiB of the window each: the walks below are serial per
                                          // wavefront, and the whole call waits for this one workgroup (0.040 -> 0.02 ms)
__global__ __launch_bounds__(kBootBlock) void k_stream_boot(const uint8_t* __restrict__ img, uint32_t span,
                                                            CallState* __restrict__ cs) {
  constexpr int kBootWaves = kBootBlock / kWave;
  __shared__ uint32_t s_cnt[kBootWaves];
  __shared__ __attribute__((aligned(16))) uint8_t s_img[kBootMax];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const uint32_t part = span / kBootWaves;  // (span is a multiple of 256: part is a multiple of 16)
  // the prefix goes to LDS first (all loads in flight at once): the two byte-wise walks below would
  // otherwise pay a memory round trip per 64 bytes each
  for (uint32_t o = threadIdx.x * 16u; o < span; o += kBootBlock * 16u)
    *reinterpret_cast<uint4*>(s_img + o) = *reinterpret_cast<const uint4*>(img + o);
  __syncthreads();
  const uint8_t* p = s_img + (uint64_t)wv * part;
  uint32_t cnt = 0;
  for (uint32_t o = lane; o < part; o += kWave) cnt += (p[o] == '\n');
  cnt = wave_sum(cnt);
  if (lane == 0) s_cnt[wv] = cnt;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int w = 0; w < kBootWaves; ++w) {
    if (w < wv) before += s_cnt[w];
    total += s_cnt[w];
  }
  const uint32_t limit = total & ~3u;
  if (threadIdx.x == 0) cs->boot_lines = total;
  uint32_t line = before, qmin = 255, qmax = 0;
  for (uint32_t o = 0; o < part; o += kWave) {
    const uint32_t c = o + lane < part ? (uint32_t)p[o + lane] : 0u;
    const uint64_t bal = __ballot(o + lane < part && c == '\n');
    const uint32_t mine = line + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
    if ((mine & 3u) == 3u && mine < limit && c != '\n' && o + lane < part) {
      qmin = c < qmin ? c : qmin;
      qmax = c > qmax ? c : qmax;
    }
    line += (uint32_t)__builtin_popcountll(bal);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
    qmin = a < qmin ? a : qmin;
    qmax = b > qmax ? b : qmax;
  }
  if (lane == 0 && qmin <= qmax) {
    atomicMin(&cs->boot_qmin, qmin);
    atomicMax(&cs->boot_qmax, qmax);
    atomicMin(&cs->qmin_byte, qmin);
    atomicMax(&cs->qmax_byte, qmax);
  }
}

// ------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------
struct StreamOut {
  uint32_t* counts;            // newlines per chunk
  uint32_t* cinfo;             // chunk info word
  uint16_t* stage;             // kStageCap entries per chunk
  unsigned long long* queue;   // suspect byte positions
  unsigned long long queue_cap;
};

__device__ __forceinline__ void queue_suspect(const StreamOut& o, CallState* cs, uint64_t pos) {
  if (__hip_atomic_load(&cs->queue_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > o.queue_cap) return;
  const unsigned long long at = atomicAdd(&cs->queue_count, 1ull);
  if (at < o.queue_cap) o.queue[at] = pos;
}

// Work decomposition of pass 1: ONE WAVEFRONT per 4 KiB chunk, walked as kHalves = 2 slices of
// 2 KiB; inside a slice every lane owns 32 CONTIGUOUS bytes (two 16-byte loads at a 32-byte lane
// stride - measured as fast as the fully interleaved pattern), so that all per-lane masks are
// 32 bits wide and the cross-lane work (scan, look-ahead) is paid once per 32 bytes.
constexpr int kHalves = 2;
constexpr int kLaneBytes = 32;
constexpr int kHalfBytes = kWave * kLaneBytes;  // 2 KiB

__device__ __forceinline__ uint32_t prefix_xor32(uint32_t x) {
  x ^= x << 1;
  x ^= x << 2;
  x ^= x << 4;
  x ^= x << 8;
  x ^= x << 16;
  return x;
}

// masks of the bytes on sequence (M1) and quality (M3) lines; t0 = type of the first byte
__device__ __forceinline__ void type_masks32(uint32_t nl, uint32_t t0, uint32_t& M1, uint32_t& M3) {
  const uint32_t e = nl << 1;
  const uint32_t P = prefix_xor32(e);        // bit 0 of the running newline count
  const uint32_t Q = prefix_xor32(e & ~P);   // bit 1 (carry when bit 0 wraps)
  const uint32_t a0 = 0u - (t0 & 1u), a1 = 0u - ((t0 >> 1) & 1u);
  const uint32_t L = P ^ a0;
  const uint32_t Hh = Q ^ a1 ^ (P & a0);
  M1 = ~Hh & L;
  M3 = Hh & L;
}

struct Piece32 {
  uint4 a, b;  // bytes 0..15, 16..31
};

// Staging of the newlines of one chunk.  Every lane writes (offset | "second next byte is a newline")
// of its newlines to the wave's LDS table at their rank; lane i then takes entry i, adds the class
// of the byte after that newline and stores the 16-bit entry (contiguous 2-byte stores).  The byte
// after a newline comes from an LDS copy of the chunk, in image order (a gather from global memory
// would miss the L2 often enough to cost 10 % more HBM reads; the copy's 16-byte stores at a 32-byte
// lane stride meet two-way bank conflicts, which the store's own issue time covers - a conflict-free
// layout of round 3 paid for itself with six address instructions per look-up).
// The name-capturing instantiation reads header lines from the copy 8 bytes at a time at any alignment: kCopyFront
// bytes of slack in front of it and kCopyBack behind - a header's 64-byte window starts 3 bytes before its '@' and may
// reach past the chunk.
constexpr uint32_t kCopyFront = 16, kCopyBack = 64;
constexpr uint32_t kCopyRow = kCopyFront + kChunkBytes + kCopyBack;

__device__ __forceinline__ uint32_t stage_entry(uint32_t ent, uint32_t c1) {
  return ent | ((c1 == '@' ? kClsAt : (c1 == '+' ? kClsPlus : 0u)) << 12);
}

template <uint32_t ABL>
__device__ __forceinline__ void stage_chunk(const uint8_t* __restrict__ img, uint64_t n, uint64_t cb, uint32_t chunk,
                                            const uint32_t (&nl)[kHalves], const uint32_t (&nl2)[kHalves],
                                            const uint32_t (&ex)[kHalves], uint32_t tot,
                                            uint16_t* __restrict__ slots, const uint8_t* __restrict__ copy,
                                            uint32_t tail, uint16_t* __restrict__ stage, bool interior) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < kHalves; ++k) {
    uint32_t m = nl[k], r = ex[k];
    const uint32_t at = (uint32_t)k * kHalfBytes + (uint32_t)lane * kLaneBytes;
    while (m) {
      const uint32_t j = (uint32_t)__builtin_ctz(m);
      m &= m - 1;
      if (r < (uint32_t)kStageCap) slots[r] = (uint16_t)((at + j) | (((nl2[k] >> j) & 1u) << 14));
      ++r;
    }
  }
  __builtin_amdgcn_wave_barrier();
  uint16_t* dst = stage + (uint64_t)chunk * kStageCap;
  for (uint32_t i = lane; i < tot; i += kWave) {
    const uint32_t ent = slots[i];
    const uint32_t o = (ent & 0xFFFu) + 1;  // the byte after the '\n'
    uint32_t c1 = 0;
    if (!(ABL & 4u)) {
      if (interior) c1 = o < (uint32_t)kChunkBytes ? (uint32_t)copy[o] : (tail & 0xFFu);
      else c1 = cb + o < n ? (uint32_t)img[cb + o] : 0u;
    }
    dst[i] = (uint16_t)stage_entry(ent, c1);
  }
}

// ABL: ablation mask for tools/kbench (product code instantiates 0): 1 = no byte-class checks,
// 2 = no staging, 4 = no fetch of the byte after a newline, 8 = no base check, 16 = no quality test
// NAMES: also capture the header lines that begin in the chunk (NameCapture, fqg_device.h)
// NAMES: 0 = no capture, 1 = 64-byte records, 2 = 16-byte digests (fqg_device.h)
// LDS of a pass-1 workgroup: the staging slots and the copies of its four chunks (static in k_stream_pass1, a piece of the
// dynamic LDS in the kernel that also runs the line workers, k_stream_pass1_lines)
template <int NAMES>
struct Pass1Lds {
  uint16_t slots[kBlock / kWave][kStageCap];
  __attribute__((aligned(16))) uint8_t copy[kBlock / kWave][NAMES ? kCopyRow : (uint32_t)kChunkBytes];
};
// block: which workgroup of the pass this is (chunks 4 block .. 4 block + 3), n_chunks: the pass ends in front of it
template <uint32_t ABL, int NAMES>
__device__ __forceinline__ void stream_pass1_body(const uint8_t* __restrict__ img, uint64_t n, uint32_t n_chunks, const StreamOut& o,
                                                  CallState* __restrict__ cs, const NameCapture& nc, uint32_t block,
                                                  Pass1Lds<NAMES>& lds) {
  static_assert(kHalves == 2 && kHalves * kHalfBytes == kChunkBytes, "one packed scan covers the two slices");
  static_assert(NAMES == 0 || !(ABL & 7u), "the name capture needs the speculation, the staged entries and the copy");
  auto& s_slots = lds.slots;
  auto& s_copy = lds.copy;
  // (the wave index is uniform: telling the compiler keeps chunk-level values in scalar registers)
  const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* const copy = s_copy[wv] + (NAMES ? kCopyFront : 0u);
  const uint32_t chunk = block * (kBlock / kWave) + wv;
  if (chunk >= n_chunks) return;
  const uint64_t cb = (uint64_t)chunk * kChunkBytes;
  const uint64_t wb = cb + (uint64_t)lane * kLaneBytes;
  const bool interior = cb + kChunkBytes + 4 <= n;
  uint32_t nl[kHalves], ex[kHalves];
  Piece32 v[kHalves];
  uint32_t flags = 0;

  uint32_t tail = 0;  // the 4 bytes after the chunk (look-ahead of the last lane)
  uint32_t prev = '\n';  // NAMES: the byte in front of the chunk (the image starts behind a virtual newline)
  const uint32_t boot_lo = cs->boot_qmin, boot_hi = cs->boot_qmax;
  if (interior) {
    tail = *reinterpret_cast<const uint32_t*>(img + cb + kChunkBytes);
    if (NAMES && chunk) prev = *reinterpret_cast<const uint32_t*>(img + cb - 4) >> 24;
#pragma unroll
    for (int k = 0; k < kHalves; ++k) {
      v[k].a = *reinterpret_cast<const uint4*>(img + wb + (uint64_t)k * kHalfBytes);
      v[k].b = *reinterpret_cast<const uint4*>(img + wb + (uint64_t)k * kHalfBytes + 16);
    }
    if (!(ABL & 6u)) {
#pragma unroll
      for (int k = 0; k < kHalves; ++k) {
        *reinterpret_cast<uint4*>(&copy[k * kHalfBytes + lane * kLaneBytes]) = v[k].a;
        *reinterpret_cast<uint4*>(&copy[k * kHalfBytes + lane * kLaneBytes + 16]) = v[k].b;
      }
    }
    uint32_t hiacc = 0, badacc = 0;
#pragma unroll
    for (int k = 0; k < kHalves; ++k) {
      hiacc = or3(or3(or3(or3(hiacc, v[k].a.x, v[k].a.y), v[k].a.z, v[k].a.w), v[k].b.x, v[k].b.y), v[k].b.z, v[k].b.w);
      const uint32_t lo = pack_marks16(nl_marks7(v[k].a.x, badacc), nl_marks7(v[k].a.y, badacc),
                                       nl_marks7(v[k].a.z, badacc), nl_marks7(v[k].a.w, badacc));
      const uint32_t hi = pack_marks16(nl_marks7(v[k].b.x, badacc), nl_marks7(v[k].b.y, badacc),
                                       nl_marks7(v[k].b.z, badacc), nl_marks7(v[k].b.w, badacc));
      nl[k] = lo | (hi << 16);
    }
    const bool high = __ballot((hiacc & kH) != 0) != 0;
    const bool ctrl = __ballot((badacc & kH) != 0) != 0;
    if (high) flags |= kFlagHigh;  // (the masks above are meaningless then; the host drops this pass)
    if (ctrl && !high) {
      // rare: a control byte other than '\n'.  Only NUL and CR change what a line is (C strings,
      // src/fastq.c:250,317); anything else (a tab in a header ...) is just a byte.
      uint32_t nul = 0, cr = 0;
#pragma unroll
      for (int k = 0; k < kHalves; ++k) {
        const uint32_t w[8] = {v[k].a.x, v[k].a.y, v[k].a.z, v[k].a.w, v[k].b.x, v[k].b.y, v[k].b.z, v[k].b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          nul |= eq_bytes(w[j], 0u);
          cr |= eq_bytes(w[j], 0x0D0D0D0Du);
        }
      }
      if (__ballot(nul != 0)) flags |= kFlagNul;
      if (__ballot(cr != 0)) flags |= kFlagCr;
    }
  } else {
    // the last one or two chunks: bytes may be missing, look-ahead may cross the end of the image
    uint32_t bad = 0;
#pragma unroll
    for (int k = 0; k < kHalves; ++k) {
      const uint64_t off = wb + (uint64_t)k * kHalfBytes;
      nl[k] = 0;
      v[k].a = v[k].b = make_uint4(0, 0, 0, 0);
      for (uint64_t i = off; i < n && i < off + kLaneBytes; ++i) {
        const uint32_t c = img[i];
        if (c == '\n') nl[k] |= 1u << (i - off);
        else if (c == 0u) bad |= kFlagNul;
        else if (c == '\r') bad |= kFlagCr;
        else if (c >= 0x80u) bad |= kFlagHigh;
      }
    }
    if (__ballot((bad & kFlagNul) != 0)) flags |= kFlagNul;
    if (__ballot((bad & kFlagCr) != 0)) flags |= kFlagCr;
    if (__ballot((bad & kFlagHigh) != 0)) flags |= kFlagHigh;
  }

  // ranks of the newlines inside the chunk: one packed scan for both slices
  const uint32_t c01 = __popc(nl[0]) | (__popc(nl[1]) << 16);
  const uint32_t s01 = wave_scan_incl_dpp(c01);
  const uint32_t t01 = __builtin_amdgcn_readlane(s01, 63);
  const uint32_t total = (t01 & 0xFFFFu) + (t01 >> 16);
  ex[0] = (s01 & 0xFFFFu) - (c01 & 0xFFFFu);
  ex[1] = (t01 & 0xFFFFu) + (s01 >> 16) - (c01 >> 16);
  if (total > (uint32_t)kStageCap) flags |= kFlagStageOverflow;

  // newline bits of the bytes 1 and 2 further on (funnel shifts over {next lane, this lane})
  uint32_t tail_nl = 0;
  if (interior) {
    tail_nl = ((tail & 0xFFu) == '\n' ? 1u : 0u) | (((tail >> 8) & 0xFFu) == '\n' ? 2u : 0u);
  } else {
    for (uint64_t i = 0; i < 2; ++i)
      if (cb + kChunkBytes + i < n && img[cb + kChunkBytes + i] == '\n') tail_nl |= 1u << i;
  }
  uint32_t nl2[kHalves], cand[kHalves];
#pragma unroll
  for (int k = 0; k < kHalves; ++k) {
    uint32_t nxt = dpp0<0x130, 0xf, 0xf>(nl[k]);  // lane i <- lane i+1
    const uint32_t first_next = (uint32_t)__builtin_amdgcn_readlane(nl[k + 1 < kHalves ? k + 1 : k], 0);
    if (lane == 63) nxt = k + 1 < kHalves ? first_next : tail_nl;
    const uint32_t n1 = __builtin_amdgcn_alignbit(nxt, nl[k], 1);
    nl2[k] = __builtin_amdgcn_alignbit(nxt, nl[k], 2);
    cand[k] = nl[k] & nl2[k] & ~n1;  // a '\n' followed by exactly one byte and another '\n'
  }

  const uint32_t tot = total < (uint32_t)kStageCap ? total : (uint32_t)kStageCap;

  uint32_t info = kInfoUnknown;
  bool found = false;  // a line type was speculated
  uint32_t t0 = 0;     // ... for the chunk's first byte
  if (!(ABL & 1u) && interior && !(flags & kFlagHigh)) {
    // ---- speculation: the first line of exactly one byte is taken for a "+" line (type 2) ----
#pragma unroll
    for (int k = 0; k < kHalves; ++k) {
      const uint64_t bal = __ballot(cand[k] != 0);
      if (!found && bal) {
        const int l = __builtin_ctzll(bal);
        const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane(cand[k], l);
        const uint32_t nll = (uint32_t)__builtin_amdgcn_readlane(nl[k], l);
        const uint32_t exl = (uint32_t)__builtin_amdgcn_readlane(ex[k], l);
        const uint32_t j = (uint32_t)__builtin_ctz(cl);
        const uint32_t a = exl + __popc(nll & ((1u << j) - 1u));  // rank in the chunk of the '\n' in front of the "+"
        t0 = (1u - a) & 3u;  // the next '\n' (rank a + 1) ends a type-2 line
        found = true;
      }
    }
    if (found) {
      info = t0;
      uint32_t lo = boot_lo, hi = boot_hi;
      if (lo > hi || lo > 127u) {
        lo = 127u;
        hi = 0u;
      }
      const uint32_t hihb_s = ((hi & 0x7Fu) | 0x80u) * 0x01010101u;
      // (in vector registers: a subtraction that reads an SGPR issues in 4 cycles instead of 2, and a three-operand
      // instruction cannot hold a literal - the compiler would keep all three in SGPRs)
      const uint32_t lo_add = in_vgpr((0x80u - lo) * 0x01010101u), hihb = in_vgpr(hihb_s), khv = in_vgpr(kH);
      QRange q{0x00FF00FFu, 0x00FF00FFu, 0u, 0u};
      bool any_viol = false;
#pragma unroll
      for (int k = 0; k < kHalves; ++k) {
        uint32_t M1, M3;
        type_masks32(nl[k], t0 + ex[k], M1, M3);
        const uint4 &a = v[k].a, &b = v[k].b;
        const uint32_t inv = pack_marks16(not_acgtn7(a.x), not_acgtn7(a.y), not_acgtn7(a.z), not_acgtn7(a.w)) |
                             (pack_marks16(not_acgtn7(b.x), not_acgtn7(b.y), not_acgtn7(b.z), not_acgtn7(b.w)) << 16);
        const uint32_t bad = (ABL & 8u) ? 0u : inv & M1 & ~nl[k];
        if (bad) queue_suspect(o, cs, wb + (uint64_t)k * kHalfBytes + (uint32_t)__builtin_ctz(bad));
        const uint32_t okq =
            pack_marks16(in_range7a(a.x, lo_add, hihb, khv), in_range7a(a.y, lo_add, hihb, khv), in_range7a(a.z, lo_add, hihb, khv),
                         in_range7a(a.w, lo_add, hihb, khv)) |
            (pack_marks16(in_range7a(b.x, lo_add, hihb, khv), in_range7a(b.y, lo_add, hihb, khv), in_range7a(b.z, lo_add, hihb, khv),
                          in_range7a(b.w, lo_add, hihb, khv)) << 16);
        const uint32_t qm = (ABL & 16u) ? 0u : M3 & ~nl[k];
        if (__ballot((qm & ~okq) != 0)) {  // rare: exact range of this slice's quality bytes
          any_viol = true;
          qrange_accum(a, qm & 0xFFFFu, q);
          qrange_accum(b, qm >> 16, q);
        }
      }
      if (any_viol) {
        uint32_t qmin = pk_min_u16(q.mn_e, q.mn_o), qmax = pk_max_u16(q.mx_e, q.mx_o);
        qmin = (qmin & 0xFFFFu) < (qmin >> 16) ? (qmin & 0xFFFFu) : (qmin >> 16);
        qmax = (qmax & 0xFFFFu) > (qmax >> 16) ? (qmax & 0xFFFFu) : (qmax >> 16);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const uint32_t x = __shfl_xor(qmin, d, 64), y = __shfl_xor(qmax, d, 64);
          qmin = x < qmin ? x : qmin;
          qmax = y > qmax ? y : qmax;
        }
        if (qmin <= qmax) info |= kInfoRange | ((qmin & 0xFFu) << 8) | ((qmax & 0xFFu) << 16);
      }
    } else if (total == 0) {
      // A chunk without any newline lies inside ONE line (reads of kilobases: most chunks).  There is no type to
      // speculate on, but both kinds of check can be made on all of its bytes, and the rank says later which one
      // counts (chunk_info_redo / chunk_info_range).  A branch of its own: the path above is the 150 bp path and
      // pays for every instruction.
      info = kInfoOneLine;
      uint32_t lo = boot_lo, hi = boot_hi;
      if (lo > hi || lo > 127u) {
        lo = 127u;
        hi = 0u;
      }
      const uint32_t lob = lo * 0x01010101u, hihb = ((hi & 0x7Fu) | 0x80u) * 0x01010101u;
      uint32_t not_base = 0, outside = 0;
#pragma unroll
      for (int k = 0; k < kHalves; ++k) {
        const uint32_t w[8] = {v[k].a.x, v[k].a.y, v[k].a.z, v[k].a.w, v[k].b.x, v[k].b.y, v[k].b.z, v[k].b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          not_base |= not_acgtn7(w[j]);
          outside |= ~in_range7(w[j], lob, hihb) & kH;
        }
      }
      if (__ballot(not_base != 0)) info |= kInfoNotBases;  // (which byte: the repeated check finds it, if the line is a sequence)
      if (__ballot(outside != 0)) {  // rare: the exact range of the chunk's bytes
        QRange q{0x00FF00FFu, 0x00FF00FFu, 0u, 0u};
#pragma unroll
        for (int k = 0; k < kHalves; ++k) {
          qrange_accum(v[k].a, 0xFFFFu, q);
          qrange_accum(v[k].b, 0xFFFFu, q);
        }
        uint32_t qmin = pk_min_u16(q.mn_e, q.mn_o), qmax = pk_max_u16(q.mx_e, q.mx_o);
        qmin = (qmin & 0xFFFFu) < (qmin >> 16) ? (qmin & 0xFFFFu) : (qmin >> 16);
        qmax = (qmax & 0xFFFFu) > (qmax >> 16) ? (qmax & 0xFFFFu) : (qmax >> 16);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const uint32_t x = __shfl_xor(qmin, d, 64), y = __shfl_xor(qmax, d, 64);
          qmin = x < qmin ? x : qmin;
          qmax = y > qmax ? y : qmax;
        }
        if (qmin <= qmax) info |= kInfoRange | ((qmin & 0xFFu) << 8) | ((qmax & 0xFFu) << 16);
      }
    }
  }

  if (!(ABL & 2u)) {
    stage_chunk<ABL>(img, n, cb, chunk, nl, nl2, ex, tot, s_slots[wv], copy, tail, o.stage, interior);
  }
  if constexpr (NAMES == 2) {
    // ---- header lines that begin in this chunk -> 16-byte digests: canonical name + hash, four lanes per header ----
    // The line that starts at v (v = 0: the chunk's first byte, else behind the chunk's v-th newline) has type t0 + v:
    // headers are every fourth line start from v0 on, so quad q of the wavefront takes the header at v0 + 4 q without a
    // compaction.  Lane `sub` of a quad looks at name bytes [16 sub, 16 sub + 16) - the bytes behind the '@'.
    uint32_t hc = kNoCapture;
    if (found && total <= (uint32_t)kStageCap) {
      hc = 0;
      const uint16_t* slots = s_slots[wv];
      const bool first_is_start = prev == '\n';
      uint32_t v0 = (0u - t0) & 3u;
      if (v0 == 0u && !first_is_start) v0 = 4u;
      const uint32_t sub = (uint32_t)lane & 3u, q = (uint32_t)lane >> 2;
      const bool casava = nc.fmt == FQG_NAME_CASAVA18;
      const uint32_t pe_cut = (nc.fmt == FQG_NAME_DEFAULT && nc.is_pe) ? 1u : 0u;
      for (uint32_t base = 0; v0 + 4u * base <= tot; base += kWave / 4) {
        const uint32_t j = base + q, vv = v0 + 4u * j;
        bool is_hdr = vv <= tot;
        uint32_t at = 0;
        if (is_hdr && vv > 0) {
          at = (slots[vv - 1] & 0xFFFu) + 1u;
          is_hdr = at < (uint32_t)kChunkBytes;
        }
        hc += (uint32_t)__builtin_popcountll(__ballot(is_hdr && sub == 0u));
        if (is_hdr && j < nc.K) {
          const bool end_known = vv < tot;
          uint32_t lnm = end_known ? (uint32_t)(slots[vv] & 0xFFFu) - at - 1u : ~0u;  // name bytes in front of the '\n'
          const bool at_sign = copy[at] == '@';
          bool ok = true;
          uint4 x;
          if (end_known) {
            __builtin_memcpy(&x, copy + at + 1u + 16u * sub, 16);
          } else if (cb + at + 1u + kDigestText <= n) {
            // the line runs into the next chunk (one header in eight at 150 bp): its bytes from the image - the
            // wavefront next door is loading them anyway - and its length from the '\n' among them, if there is one
            __builtin_memcpy(&x, img + cb + at + 1u + 16u * sub, 16);
            const uint32_t m = pack_marks16(eq_bytes(x.x, 0x0A0A0A0Au), eq_bytes(x.y, 0x0A0A0A0Au), eq_bytes(x.z, 0x0A0A0A0Au),
                                            eq_bytes(x.w, 0x0A0A0A0Au));
            uint32_t mine = m ? 16u * sub + (uint32_t)__builtin_ctz(m) : ~0u;
            uint32_t o = dpp0<0xB1, 0xf, 0xf>(mine);  // quad_perm [1,0,3,2]
            mine = o < mine ? o : mine;
            o = dpp0<0x4E, 0xf, 0xf>(mine);           // quad_perm [2,3,0,1]
            lnm = o < mine ? o : mine;
          } else {
            x = make_uint4(0, 0, 0, 0);
            ok = false;
          }
          uint32_t nlen = 0, acct = 0;
          if (casava) {
            // the name ends at the first blank of the line (src/fastq.c:496-503); "/1" in front of it is dropped
            const uint32_t m = pack_marks16(eq_bytes(x.x, 0x20202020u), eq_bytes(x.y, 0x20202020u), eq_bytes(x.z, 0x20202020u),
                                            eq_bytes(x.w, 0x20202020u));
            uint32_t mine = m ? 16u * sub + (uint32_t)__builtin_ctz(m) : ~0u;
            uint32_t o = dpp0<0xB1, 0xf, 0xf>(mine);
            mine = o < mine ? o : mine;
            o = dpp0<0x4E, 0xf, 0xf>(mine);
            const uint32_t sp = o < mine ? o : mine;
            if (sp >= lnm || sp >= kDigestText) ok = false;  // no blank in the line (or none in what the quad sees of it)
            // the byte two places in front of the blank: with the lane that holds it
            const uint32_t two = sp - 2u;
            uint32_t slash = 0;
            if (ok && sp >= 2u && (two >> 4) == sub) {
              const uint32_t wsel = (two >> 2) & 3u;
              const uint32_t wd = wsel == 0u ? x.x : wsel == 1u ? x.y : wsel == 2u ? x.z : x.w;
              slash = ((wd >> (8u * (two & 3u))) & 0xFFu) == '/' ? 1u : 0u;
            }
            slash |= dpp0<0xB1, 0xf, 0xf>(slash);
            slash |= dpp0<0x4E, 0xf, 0xf>(slash);
            nlen = acct = ok ? sp - 2u * slash : 0u;
          } else {
            // strlen(&hdr[1]) counts the '\n' (src/fastq.c:505-511); a line whose end nobody saw, or one so short that
            // the name would hold the '\n', is left to the byte-wise path
            const uint32_t L = lnm + 1u;
            if (lnm == ~0u || L < 1u + pe_cut) ok = false;
            const uint32_t l = L - pe_cut;
            acct = l;
            nlen = l - 1u;
            if (nlen > kDigestText) ok = false;
          }
          if (nlen > 1023u || acct > 1023u) ok = false;
          if (!ok) nlen = acct = 0;
          // the hash: this lane's two words, bytes behind the name zeroed, summed over the quad
          const uint32_t lo = 16u * sub;
          const uint32_t k0 = nlen > lo ? nlen - lo : 0u;             // name bytes in this lane's first word and beyond
          const uint32_t k1 = nlen > lo + 8u ? nlen - lo - 8u : 0u;   // ... in its second word and beyond
          uint64_t w0 = ((uint64_t)x.y << 32) | x.x, w1 = ((uint64_t)x.w << 32) | x.z;
          w0 = k0 >= 8u ? w0 : (w0 & ((1ull << (8u * k0)) - 1ull));
          w1 = k1 >= 8u ? w1 : (w1 & ((1ull << (8u * k1)) - 1ull));
          // (the keys by selection: sub is one of four; a word counts when it holds a name byte - name_word)
          const uint32_t a0 = sub == 0u ? name_key_a(0) : sub == 1u ? name_key_a(2) : sub == 2u ? name_key_a(4) : name_key_a(6);
          const uint32_t b0 = sub == 0u ? name_key_b(0) : sub == 1u ? name_key_b(2) : sub == 2u ? name_key_b(4) : name_key_b(6);
          const uint32_t a1 = sub == 0u ? name_key_a(1) : sub == 1u ? name_key_a(3) : sub == 2u ? name_key_a(5) : name_key_a(7);
          const uint32_t b1 = sub == 0u ? name_key_b(1) : sub == 1u ? name_key_b(3) : sub == 2u ? name_key_b(5) : name_key_b(7);
          uint64_t part = 0;
          if (k0) part = (uint64_t)((uint32_t)w0 + a0) * (uint64_t)((uint32_t)(w0 >> 32) + b0);
          if (k1) part += (uint64_t)((uint32_t)w1 + a1) * (uint64_t)((uint32_t)(w1 >> 32) + b1);
          uint32_t plo = (uint32_t)part, phi = (uint32_t)(part >> 32);
          uint64_t oth = ((uint64_t)dpp0<0xB1, 0xf, 0xf>(phi) << 32) | dpp0<0xB1, 0xf, 0xf>(plo);
          part += oth;
          plo = (uint32_t)part;
          phi = (uint32_t)(part >> 32);
          oth = ((uint64_t)dpp0<0x4E, 0xf, 0xf>(phi) << 32) | dpp0<0x4E, 0xf, 0xf>(plo);
          part += oth;
          if (sub == 0u) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2 dg;
            dg.x = name_fin(name_seed(nlen) + part);
            dg.y = (unsigned long long)(nlen | (acct << 10) | (vv << 20) | (at_sign ? kDigestAt : 0u) | (ok ? kDigestOk : 0u));
            *reinterpret_cast<u64x2*>(nc.recs + ((uint64_t)chunk * nc.K + j) * kDigestWords) = dg;
          }
        }
      }
      if (hc >= kNoCapture) hc = kNoCapture - 1u;
    }
    if (lane == 0) nc.hcount[chunk] = (uint16_t)hc;
  }
  if constexpr (NAMES == 1) {
    // ---- header lines that begin in this chunk -> 64-byte records (NameCapture) ----
    // Line starts: the chunk's first byte when the byte in front of it is a '\n' (v = 0), and the byte behind the
    // chunk's v-th newline unless that newline is the chunk's last byte; the line that starts at v has type t0 + v.
    uint32_t hc = kNoCapture;
    if (found && total <= (uint32_t)kStageCap) {
      hc = 0;
      const uint16_t* slots = s_slots[wv];
      const bool first_is_start = prev == '\n';
      for (uint32_t base = 0; base <= tot; base += kWave) {
        const uint32_t vv = base + (uint32_t)lane;
        bool is_hdr = vv <= tot && ((t0 + vv) & 3u) == 0u && (vv > 0 || first_is_start);
        uint32_t at = 0;
        if (is_hdr && vv > 0) {
          at = (slots[vv - 1] & 0xFFFu) + 1u;
          is_hdr = at < (uint32_t)kChunkBytes;
        }
        const unsigned long long hm = __ballot(is_hdr);
        const uint32_t j = hc + (uint32_t)__builtin_popcountll(hm & ((1ull << lane) - 1ull));
        hc += (uint32_t)__builtin_popcountll(hm);
        if (is_hdr && j < nc.K) {
          bool end_known = vv < tot, real = false;
          uint32_t len = (end_known ? (uint32_t)(slots[vv] & 0xFFFu) : (uint32_t)kChunkBytes) - at;
          unsigned long long w[kNameRecWords];
          if (end_known || cb + at + kNameRecText + 1 > n) {
#pragma unroll
            for (uint32_t k = 0; k < kNameRecWords; ++k) __builtin_memcpy(&w[k], copy + at + 8u * k - 3u, 8);  // byte 3 = the line's first byte
          } else {
            // the line runs into the next chunk (one header in eight does at 150 bp): its bytes from the image - the
            // wavefront next door is loading them anyway - and its length from the '\n' among them, if there is one
            const uint8_t* g = img + cb + at - 3u;
            unsigned long long nlm[kNameRecWords];
#pragma unroll
            for (uint32_t k = 0; k < kNameRecWords; ++k) {
              __builtin_memcpy(&w[k], g + 8u * k, 8);
              const unsigned long long y = w[k] ^ 0x0A0A0A0A0A0A0A0Aull;
              nlm[k] = ~(((y & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | y | 0x7F7F7F7F7F7F7F7Full);
            }
            nlm[0] &= ~0xFFFFFFull;  // (the three bytes in front of the line)
            real = true;
            len = kNameRecText + 1;  // at least: no '\n' among the bytes of the record
#pragma unroll
            for (int k = (int)kNameRecWords - 1; k >= 0; --k)
              if (nlm[k]) {
                len = 8u * (uint32_t)k + ((uint32_t)__builtin_ctzll(nlm[k]) >> 3) - 3u;
                end_known = true;
              }
          }
          len = len > 1023u ? 1023u : len;
          const uint32_t meta = len | (vv << 10) | (end_known ? 1u << 19 : 0u) | (((w[0] >> 24) & 0xFFu) == '@' ? 1u << 20 : 0u) |
                                (real ? 1u << 21 : 0u);
          w[0] = (w[0] & 0xFFFFFFFF00000000ull) | meta;
          typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
          u64x2* dst = reinterpret_cast<u64x2*>(nc.recs + ((uint64_t)chunk * nc.K + j) * kNameRecWords);
#pragma unroll
          for (uint32_t k = 0; k < kNameRecWords / 2; ++k) {
            u64x2 x;
            x.x = w[2 * k];
            x.y = w[2 * k + 1];
            dst[k] = x;
          }
        }
      }
      if (hc >= kNoCapture) hc = kNoCapture - 1u;
    }
    if (lane == 0) nc.hcount[chunk] = (uint16_t)hc;
  }
  if (lane == 0) {
    o.counts[chunk] = total;
    o.cinfo[chunk] = info;
    if (flags) atomicOr(&cs->flags, flags);
  }
}

template <uint32_t ABL, int NAMES = 0>
__global__ __launch_bounds__(kBlock) void k_stream_pass1(const uint8_t* __restrict__ img, uint64_t n,
                                                         uint32_t n_chunks, StreamOut o,
                                                         CallState* __restrict__ cs, NameCapture nc = NameCapture{}) {
  __shared__ Pass1Lds<NAMES> lds;
  stream_pass1_body<ABL, NAMES>(img, n, n_chunks, o, cs, nc, blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// pass 2: staged entries -> line index, header-start checks, verification of the speculation
// ------------------------------------------------------------------------------------------
constexpr int kP2Batch = 8;  // chunks per wavefront (their loads are issued together)

__global__ __launch_bounds__(kBlock) void k_stream_pass2(const uint8_t* __restrict__ img, uint64_t n,
                                                         uint32_t n_chunks, const uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ cinfo,
                                                         const uint16_t* __restrict__ stage,
                                                         const uint32_t* __restrict__ chunk_local,
                                                         const unsigned long long* __restrict__ span_excl,
                                                         uint64_t* __restrict__ line_end, uint64_t line_cap,
                                                         uint64_t limit, SuspectMap suspect,
                                                         uint32_t* __restrict__ redo, CallState* __restrict__ cs) {
  const int lane = lane_id();
  const uint32_t c0 = (blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kP2Batch;
  if (c0 >= n_chunks) return;
  // lane b < kP2Batch holds the scalars of chunk c0 + b
  uint32_t cnt = 0, info = 0;
  uint64_t rank0 = 0;
  const uint32_t mine = c0 + (uint32_t)lane;
  if (lane < kP2Batch && mine < n_chunks) {
    cnt = counts[mine];
    info = cinfo[mine];
    rank0 = span_excl[mine / kScanSpan] + chunk_local[mine];
  }
  uint32_t e[kP2Batch], tot[kP2Batch];
  uint64_t r0[kP2Batch];
#pragma unroll
  for (int b = 0; b < kP2Batch; ++b) {
    const uint32_t cb_cnt = (uint32_t)__builtin_amdgcn_readlane(cnt, b);
    tot[b] = cb_cnt < (uint32_t)kStageCap ? cb_cnt : (uint32_t)kStageCap;
    r0[b] = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)rank0, b) |
            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(rank0 >> 32), b) << 32);
    e[b] = (uint32_t)lane < tot[b] ? stage[(uint64_t)(c0 + b) * kStageCap + lane] : 0u;
  }
  auto entry = [&](uint32_t chunk, uint64_t rank, uint32_t i, uint32_t ent) {
    const uint64_t L = rank + i;
    if (L < line_cap) line_end[L] = (uint64_t)chunk * kChunkBytes + (ent & 0xFFFu);
    const uint64_t nx = L + 1;  // the line that starts after this '\n'
    if (nx < limit) {
      const uint32_t cls = (ent >> 12) & 3u, nl2 = (ent >> 14) & 1u;
      const uint32_t t = (uint32_t)nx & 3u;
      const bool bad = t == 0 ? !(cls == kClsAt && !nl2) : (t == 2 ? !(cls == kClsPlus && nl2) : false);
      if (bad) mark_suspect(suspect, nx >> 2);
    }
  };
#pragma unroll
  for (int b = 0; b < kP2Batch; ++b) {
    if ((uint32_t)lane < tot[b]) entry(c0 + b, r0[b], (uint32_t)lane, e[b]);
    for (uint32_t i = kWave + lane; i < tot[b]; i += kWave)  // more than 64 newlines in 4 KiB: rare
      entry(c0 + b, r0[b], i, stage[(uint64_t)(c0 + b) * kStageCap + i]);
  }
  const bool own = lane < kP2Batch && mine < n_chunks;
  bool again = false;
  if (own) {
    if (mine == 0 && limit > 0 && (img[0] != '@' || (n > 1 && img[1] == '\n'))) mark_suspect(suspect, 0);
    // only chunks that hold bytes of complete records need their byte checks to stand
    if (rank0 < limit) {
      // (a chunk that reaches beyond the last complete record is repeated too: its quality range may
      // include bytes of an incomplete record)
      if (chunk_info_redo(info, (uint32_t)rank0) || rank0 + cnt >= limit) again = true;
      else if (chunk_info_range(info, (uint32_t)rank0)) {
        // (look first: the hull settles after a few chunks, and every later one would still be an atomic on one address)
        const uint32_t qlo = (info >> 8) & 0xFFu, qhi = (info >> 16) & 0xFFu;
        if (qlo < __atomic_load_n(&cs->qmin_byte, __ATOMIC_RELAXED)) atomicMin(&cs->qmin_byte, qlo);
        if (qhi > __atomic_load_n(&cs->qmax_byte, __ATOMIC_RELAXED)) atomicMax(&cs->qmax_byte, qhi);
      }
    }
    if (mine == n_chunks - 1 && n > 0 && !cs->last_byte_is_nl && cs->n_newlines < line_cap)
      line_end[cs->n_newlines] = n;
  }
  // the redo list: one reservation per wavefront (long reads send EVERY chunk here - no "+" line in 4 KiB to
  // speculate from - and two million single adds to one address took longer than the rest of the kernel)
  const unsigned long long am = __ballot(again);
  if (am) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&cs->redo_count, (uint32_t)__builtin_popcountll(am));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (again) {
      const uint32_t at = base + (uint32_t)__builtin_popcountll(am & ((1ull << lane) - 1ull));
      if (at < n_chunks) redo[at] = mine;
    }
  }
}

// ------------------------------------------------------------------------------------------
// pass 2, second form: the per-chunk duties (k_stream_chunks) and the per-line / per-record duties
// (k_stream_lines) separately.  k_stream_pass2 above walks the image chunk by chunk, so a wavefront's
// line_end[] stores start wherever its first chunk's rank happens to fall, and the per-record statistics
// needed a further pass over the 32 B / record it had just written (k_records_fast).  Here a LANE OWNS A
// RECORD: it fetches the five staged entries that bound its four lines (2-byte loads, neighbouring lanes
// neighbouring addresses), writes the record's four line ends as one aligned 32-byte piece (a wavefront writes
// 2 KiB, aligned), checks the header starts, and derives the lengths, the statistics and the suspect bit from
// registers - the line index is written once and not read again.
// ------------------------------------------------------------------------------------------
struct ChunkRanks {
  const uint32_t* counts;              // newlines per chunk
  const uint32_t* local;               // exclusive prefix inside a span of kScanSpan chunks
  const unsigned long long* span_excl; // exclusive prefix over the spans
  uint32_t n_chunks;
  __device__ __forceinline__ uint64_t rank0(uint32_t c) const { return span_excl[c / kScanSpan] + local[c]; }
};

__global__ __launch_bounds__(kBlock) void k_stream_chunks(ChunkRanks cr, const uint32_t* __restrict__ cinfo, uint64_t limit,
                                                          uint32_t* __restrict__ redo, CallState* __restrict__ cs) {
  const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
  const int lane = lane_id();
  bool again = false;
  if (c < cr.n_chunks) {
    const uint64_t rank0 = cr.rank0(c);
    const uint32_t cnt = cr.counts[c], info = cinfo[c];
    // only chunks that hold bytes of complete records need their byte checks to stand (a chunk that reaches
    // beyond the last complete record is repeated too: its quality range may include bytes of an incomplete one)
    if (rank0 < limit) {
      if (chunk_info_redo(info, (uint32_t)rank0) || rank0 + cnt >= limit) again = true;
      else if (chunk_info_range(info, (uint32_t)rank0)) {
        // (look first: the hull settles after a few chunks, and every later one would still be an atomic on one address)
        const uint32_t qlo = (info >> 8) & 0xFFu, qhi = (info >> 16) & 0xFFu;
        if (qlo < __atomic_load_n(&cs->qmin_byte, __ATOMIC_RELAXED)) atomicMin(&cs->qmin_byte, qlo);
        if (qhi > __atomic_load_n(&cs->qmax_byte, __ATOMIC_RELAXED)) atomicMax(&cs->qmax_byte, qhi);
      }
    }
  }
  // the redo list: one reservation per WORKGROUP (reads of kilobases send a quarter of their chunks here - a
  // reservation per wavefront was 33 000 atomics with a return value on one address, most of this kernel)
  __shared__ uint32_t s_cnt[kBlock / kWave], s_base;
  const unsigned long long am = __ballot(again);
  const int wv = (int)(threadIdx.x >> 6);
  if (lane == 0) s_cnt[wv] = (uint32_t)__builtin_popcountll(am);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) tot += s_cnt[w];
    s_base = tot ? atomicAdd(&cs->redo_count, tot) : 0u;
  }
  __syncthreads();
  if (again) {
    uint32_t at = s_base + (uint32_t)__builtin_popcountll(am & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; ++w) at += s_cnt[w];
    if (at < cr.n_chunks) redo[at] = c;
  }
}

// the chunk that holds the newline of rank R (R < total newlines): the LAST chunk whose first rank is <= R (chunks
// without newlines share their first rank with the chunk behind them).  `g` = where to start looking.
__device__ __forceinline__ uint32_t chunk_of_rank(const ChunkRanks& cr, uint64_t R, uint32_t g) {
  const uint32_t n = cr.n_chunks;
  if (g >= n) g = n - 1;
  uint32_t lo, hi, step = 1;  // invariant: rank0(lo) <= R and (hi == n or rank0(hi) > R)
  if (cr.rank0(g) <= R) {
    lo = g;
    for (;;) {
      const uint32_t t = lo + step;
      if (t >= n) {
        hi = n;
        break;
      }
      if (cr.rank0(t) <= R) {
        lo = t;
        step <<= 1;
      } else {
        hi = t;
        break;
      }
    }
  } else {
    hi = g;
    for (;;) {
      const uint32_t t = hi > step ? hi - step : 0;  // (rank0(0) = 0 <= R)
      if (cr.rank0(t) <= R) {
        lo = t;
        break;
      }
      hi = t;
      step <<= 1;
    }
  }
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (cr.rank0(mid) <= R) lo = mid;
    else hi = mid;
  }
  return lo;
}

struct LinesArgs {
  const uint8_t* img;
  uint64_t n;                 // image bytes
  ChunkRanks cr;
  const uint16_t* stage;
  uint64_t* line_end;
  uint64_t line_cap;
  uint64_t n_newlines;        // ranks 0 .. n_newlines - 1 have a staged entry
  uint64_t n_lines;           // + 1 when the image ends in an unterminated line
  uint64_t limit;             // 4 * complete records
  uint32_t* suspect_bits;     // one bit per record; this kernel owns whole words
  uint64_t suspect_cap;
  unsigned int* flags;
  int space;
  uint32_t weight;
  AccState* acc;              // null: no statistics
  unsigned long long* hist;
  int ablate;                 // measurement only (FQGPU_LINES_ABL): 1 = no line-index stores
  // The line index on demand: a call that only validates never reads most of it back (32 B per record written for
  // nothing: 3.6 GB per 100 M records).  no_index: only the records that hold a line >= keep_from are stored (the last
  // complete record - its end is what the call consumed - and the lines of an incomplete one); index_only: the stores
  // and nothing else - the second run, when something does ask for the index (index_now in fqg_abi.hip).
  uint32_t no_index, index_only;
  uint64_t keep_from;
  uint64_t step_lo = 0, step_hi = 0;  // k_stream_lines_fast: the steps of this launch ([0, every step) by default)
};

constexpr int kLinesHist = 3072;
constexpr int kLinesPer = 2;  // records per lane and step: their staged-entry loads are in flight together (4: 1.29 -> 1.41 ms; 6 wavefronts per SIMD instead of 5: no change)
template <bool TODO>
__global__ __launch_bounds__(kBlock) void k_stream_lines(LinesArgs A, const uint8_t* __restrict__ todo) {
  __shared__ uint32_t s_hist[kLinesHist];
  __shared__ unsigned long long s_red[3][kBlock / kWave];
  __shared__ uint64_t s_wr0[kBlock / kWave][kWave];   // per wavefront: first rank of the window's chunks
  __shared__ uint32_t s_wcnt[kBlock / kWave][kWave];  // ... and their newline counts
  for (int i = threadIdx.x; i < kLinesHist; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
  uint32_t win0 = 0;
  // a step = kLinesPer * 64 consecutive records (kLinesPer * 256 lines) per wavefront
  const uint64_t n_groups = (A.n_lines + 4 * kWave - 1) / (4 * kWave);
  const uint64_t n_steps = (n_groups + kLinesPer - 1) / kLinesPer;
  // `todo` (what k_stream_lines_fast left over: one byte per step): only the steps it marks, each with a window
  // request of its own - they are few, or this kernel would have been launched alone
  const uint64_t wave0 = (uint64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const double chunks_per_line = A.n_newlines ? (double)A.cr.n_chunks / (double)A.n_newlines : 0.0;
  unsigned long long n_ok = 0, min_rl = ~0ull, max_rl = 0;
  // The window of the next step is REQUESTED here and looked at when that step begins: nothing is computed from the
  // loaded values before (no sum, no select - an instruction that reads a loaded value is where the wavefront waits for
  // it), and the request has no branch (a lane outside the chunk table asks for the last chunk; `nx_in` says so).
  unsigned long long nx_se = 0;
  uint32_t nx_loc = 0, nx_cnt = 0, nx_win = 0;
  bool nx_in = false;
  auto window_request = [&](uint64_t step) {
    const uint64_t g = (step < n_steps ? step : n_steps - 1) * kLinesPer;
    const uint64_t Rw = g ? 4 * g * kWave - 1 : 0;
    const double est = (double)Rw * chunks_per_line;
    uint32_t cw = est > 2.0 ? (uint32_t)(est - 2.0) : 0u;
    if (cw + kWave > A.cr.n_chunks) cw = A.cr.n_chunks > (uint32_t)kWave ? A.cr.n_chunks - kWave : 0u;
    const uint32_t cm = cw + (uint32_t)lane;
    nx_in = cm < A.cr.n_chunks;
    const uint32_t cl = nx_in ? cm : A.cr.n_chunks - 1;
    nx_se = A.cr.span_excl[cl / kScanSpan];
    nx_loc = A.cr.local[cl];
    nx_cnt = A.cr.counts[cl];
    nx_win = cw;
  };
  if (!TODO && wave0 < n_steps) window_request(wave0);
  // TODO: the marks of 64 consecutive steps at a time, a lane each (a wavefront that asked for them one by one waited
  // for 150 loads in a row to find its two or three steps: 66 of the 886 us of the line kernels)
  uint64_t blk = wave0 * (uint64_t)kWave, blk_cur = 0;
  unsigned long long marks = 0;
  for (uint64_t step = wave0;; step += n_waves) {
    if constexpr (TODO) {
      while (!marks && blk < n_steps) {
        const uint64_t s = blk + (uint64_t)lane;
        marks = __ballot(s < n_steps && todo[s] != 0);
        blk_cur = blk;
        blk += n_waves * (uint64_t)kWave;
      }
      if (!marks) break;
      step = blk_cur + (uint64_t)__builtin_ctzll(marks);
      marks &= marks - 1;
      window_request(step);
    } else if (step >= n_steps) break;
    // e[0] = end of the line before mine, e[1..4] = ends of my four lines; ent[] = their staged entries
    uint64_t e[kLinesPer][5];
    uint32_t ent[kLinesPer][5];
    bool have[kLinesPer][5];
    // The chunks this wavefront's ranks live in: a window of 64 chunks starting a little before the estimated
    // chunk of its first rank, loaded by all lanes at once into LDS.  A lane whose rank falls outside it (reads
    // of kilobases: few newlines per chunk) searches the chunk prefix on its own.
    s_wr0[wv][lane] = nx_in ? nx_se + nx_loc : ~0ull;
    s_wcnt[wv][lane] = nx_in ? nx_cnt : 0u;
    win0 = nx_win;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- phase 1a: where the entries are (LDS only, but for lanes outside the window) ----
    uint64_t at[kLinesPer][5];   // index into stage[] of the entry to fetch
    bool ld[kLinesPer][5];       // ... when there is one
#pragma unroll
    for (int q = 0; q < kLinesPer; ++q) {
      const uint64_t g = step * kLinesPer + q;
      const uint64_t r = g * kWave + (uint64_t)lane;   // my record
      const uint64_t L0 = 4 * r;                       // its first line
      uint64_t R = r ? L0 - 1 : 0;  // first rank to fetch
      const int k0 = r ? 0 : 1;
      e[q][0] = ~0ull;                 // (record 0: the line before starts at -1)
      ent[q][0] = (kClsAt << 12);      // ... and the image's first byte is checked directly below
      have[q][0] = r == 0;
      at[q][0] = 0;
      ld[q][0] = false;
      uint32_t c = 0, i = 0, cnt = 0;
      bool windowed = false;  // c is an index INTO the window (then counts come from LDS)
      if (R < A.n_newlines) {
        if (R >= s_wr0[wv][0] && (R < s_wr0[wv][kWave - 1] + s_wcnt[wv][kWave - 1])) {
          uint32_t lo = 0, hi = kWave;  // last window entry whose first rank is <= R
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_wr0[wv][mid] <= R) lo = mid;
            else hi = mid;
          }
          c = lo;
          i = (uint32_t)(R - s_wr0[wv][lo]);
          cnt = s_wcnt[wv][lo];
          windowed = true;
        } else {
          c = chunk_of_rank(A.cr, R, (uint32_t)((double)R * chunks_per_line));
          i = (uint32_t)(R - A.cr.rank0(c));
          cnt = A.cr.counts[c];
        }
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if (k < k0) continue;
        at[q][k] = 0;
        ld[q][k] = false;
        if (R < A.n_newlines) {
          while (i >= cnt) {  // next chunk that holds a newline
            ++c;
            i = 0;
            if (windowed && c >= (uint32_t)kWave) {  // ran out of the window
              windowed = false;
              c += win0;
            }
            if (windowed) cnt = s_wcnt[wv][c];
            else cnt = __builtin_nontemporal_load(&A.cr.counts[c]);  // (a plain load here: the compiler merges the two into a load through a generic pointer and fails on it)
          }
          const uint32_t cg = windowed ? win0 + c : c;
          at[q][k] = (uint64_t)cg * kStageCap + i;
          ld[q][k] = true;
          ent[q][k] = 0;
          e[q][k] = (uint64_t)cg * kChunkBytes;  // (+ the entry's offset, below)
          have[q][k] = true;
          ++i;
        } else if (R == A.n_newlines && A.n_lines > A.n_newlines) {  // the unterminated last line
          ent[q][k] = 0;
          e[q][k] = A.n;
          have[q][k] = true;
        } else {
          ent[q][k] = 0;
          e[q][k] = 0;
          have[q][k] = false;
        }
        ++R;
      }
    }
    // ---- phase 1b: every request of the step, nothing else: the next step's window, then the kLinesPer * 5 staged
    // entries of the lane (a lane without an entry asks for entry 0) ----
    if constexpr (!TODO) window_request(step + n_waves);
    uint16_t raw[kLinesPer][5];
#pragma unroll
    for (int q = 0; q < kLinesPer; ++q)
#pragma unroll
      for (int k = 0; k < 5; ++k) raw[q][k] = A.stage[at[q][k]];
#pragma unroll
      for (int q = 0; q < kLinesPer; ++q)
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (ld[q][k]) ent[q][k] = raw[q][k];

    __builtin_amdgcn_wave_barrier();  // (the window is rewritten by the next step)
    // ---- phase 2: the line index, the checks, the statistics ----
#pragma unroll
    for (int q = 0; q < kLinesPer; ++q) {
      const uint64_t g = step * kLinesPer + q;
      const uint64_t r = g * kWave + (uint64_t)lane;
      const uint64_t L0 = 4 * r;
      {
        uint64_t R = r ? L0 - 1 : 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          if (k < (r ? 0 : 1)) continue;
          if (R < A.n_newlines) e[q][k] += (ent[q][k] & 0xFFFu);
          ++R;
        }
      }
      // four ends per lane, 32 contiguous bytes
      if ((A.ablate & 1) || (A.no_index && L0 + 4 <= A.keep_from)) {
      } else if (have[q][4] && L0 + 3 < A.line_cap) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 lo2, hi2;
        lo2.x = e[q][1]; lo2.y = e[q][2]; hi2.x = e[q][3]; hi2.y = e[q][4];
        __builtin_nontemporal_store(lo2, reinterpret_cast<u64x2*>(A.line_end + L0));
        __builtin_nontemporal_store(hi2, reinterpret_cast<u64x2*>(A.line_end + L0 + 2));
      } else {
#pragma unroll
        for (int k = 1; k < 5; ++k)
          if (have[q][k] && L0 + k - 1 < A.line_cap) A.line_end[L0 + k - 1] = e[q][k];
      }
      // checks (complete records only)
      bool sus = false, complete = have[q][4] && L0 + 3 < A.limit && !A.index_only, counted = false;
      uint64_t rl = 0;
      if (complete) {
        // header line: '@' and not empty; third line: exactly "+\n"  (what the entry BEFORE a line says about it)
        if (r == 0) sus |= A.img[0] != '@' || (A.n > 1 && A.img[1] == '\n');
        else sus |= !(((ent[q][0] >> 12) & 3u) == kClsAt && !((ent[q][0] >> 14) & 1u));
        sus |= !(((ent[q][2] >> 12) & 3u) == kClsPlus && ((ent[q][2] >> 14) & 1u));
        const uint64_t l0 = e[q][1] - e[q][0] - 1, l1 = e[q][2] - e[q][1] - 1, l2 = e[q][3] - e[q][2] - 1, l3 = e[q][4] - e[q][3] - 1;
        const uint32_t has_nl = e[q][4] < A.n ? 1u : 0u;  // only the very last line can lack it
        sus |= l1 < 1 || l1 != l3 || A.space != FQG_SPACE_SEQ;
        sus |= l0 + 1 > FQG_MAX_LABEL_LENGTH - 1 || l2 + 1 > FQG_MAX_LABEL_LENGTH - 1 ||
               l1 + 1 > FQG_MAX_READ_LENGTH - 1 || l3 + has_nl > FQG_MAX_READ_LENGTH - 1;
        counted = A.acc && l1 + 1 <= FQG_MAX_READ_LENGTH - 1;
        if (counted) {
          rl = l1 + 1;  // strlen(seq): the sequence line always ends in '\n' here
          ++n_ok;
          min_rl = rl < min_rl ? rl : min_rl;
          max_rl = rl > max_rl ? rl : max_rl;
        }
      }
      // the length histogram: reads of ONE length are the usual file, and 64 lanes adding 1 to one LDS word are 64
      // serialised atomics (they were most of this kernel's LDS cycles: SQ_LDS_BANK_CONFLICT 3.5x SQ_ACTIVE_INST_LDS in
      // profiles/r03b_pass1_sq_counters.json) - the lanes that share the first counted lane's length add once, together
      {
        const unsigned long long cm = __ballot(counted);
        if (cm) {
          const int first = __builtin_ctzll(cm);
          const uint32_t rl0 = (uint32_t)__builtin_amdgcn_readlane((uint32_t)rl, first);
          const bool with_first = counted && rl == (uint64_t)rl0;
          const unsigned long long same = __ballot(with_first);
          if (lane == first) {
            if (rl < (uint64_t)kLinesHist) atomicAdd(&s_hist[rl], (uint32_t)__builtin_popcountll(same));
            else atomicAdd(&A.hist[rl], (unsigned long long)A.weight * (unsigned long long)__builtin_popcountll(same));
          }
          if (counted && !with_first) {
            if (rl < (uint64_t)kLinesHist) atomicAdd(&s_hist[rl], 1u);
            else atomicAdd(&A.hist[rl], (unsigned long long)A.weight);
          }
        }
      }
      // the wavefront owns records g*64 .. g*64+63 = two whole words of the bitmap
      const unsigned long long sm = __ballot(sus);
      if (lane == 0 && sm) {
        const uint64_t w = g * 2;
        if ((g + 1) * kWave <= A.suspect_cap) {
          if ((uint32_t)sm) A.suspect_bits[w] = (uint32_t)sm;
          if ((uint32_t)(sm >> 32)) A.suspect_bits[w + 1] = (uint32_t)(sm >> 32);
        } else atomicOr(A.flags, kFlagSuspectOverflow);
      }
    }
  }
  if (!A.acc) return;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    n_ok += __shfl_down(n_ok, d, 64);
    const unsigned long long a = __shfl_down(min_rl, d, 64), b = __shfl_down(max_rl, d, 64);
    min_rl = a < min_rl ? a : min_rl;
    max_rl = b > max_rl ? b : max_rl;
  }
  if (lane == 0) {
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
      atomicAdd(&A.acc->num_rds, n_ok * A.weight);
      if (min_rl < A.acc->min_rl) atomicMin(&A.acc->min_rl, min_rl);
      if (max_rl > A.acc->max_rl) atomicMax(&A.acc->max_rl, max_rl);
    }
  }
  for (int i = threadIdx.x; i < kLinesHist; i += kBlock) {
    const uint32_t cc = s_hist[i];
    if (cc) atomicAdd(&A.hist[i], (unsigned long long)cc * A.weight);
  }
}

// ------------------------------------------------------------------------------------------
// The same pass for the steps of an ordinary file - every record of the step complete, the newlines of its 513 ranks in
// at most kFastSlots pieces of 64 staged entries - without a search per lane: the WAVEFRONT fetches the staged entries
// of the chunks its ranks live in, piece by piece with all lanes (coalesced 2-byte loads, all pieces in flight
// together), and drops them into an LDS array BY RANK; a lane then finds the five entries of a record as five
// neighbouring words.  k_stream_lines above works out, per lane and record, which chunk a rank lives in (a binary
// search of the window, then a walk from chunk to chunk: 585 vector and 350 scalar instructions per 128 records,
// most of them 64-bit index arithmetic); here that is one comparison per window chunk.  Steps that do not qualify (the
// first and the last records of an image, reads of kilobases whose ranks spread over more chunks than the window
// holds, chunks with more than 64 newlines) are marked for the general kernel, which runs behind this one on the marks;
// an image whose chunks hold more than 64 newlines on average (reads below 100 bases) goes to the general kernel alone.
// ------------------------------------------------------------------------------------------
constexpr int kFastSlots = 16;
constexpr int kFastRanks = 4 * kWave * kLinesPer + 1;  // 513: the ranks of 128 records and the one in front of them

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) |
         ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
}

struct LinesFastLds {
  uint32_t hist[kLinesHist];
  unsigned long long red[3][kBlock / kWave];
  __attribute__((aligned(16))) uint32_t ent[kBlock / kWave][(kFastRanks + 7) & ~3];
};
// worker / n_workers: which workgroup of the persistent grid this is; the steps it shares are [A.step_lo, A.step_hi)
// (step_hi = 0: up to the last step of the image)
__device__ __forceinline__ void stream_lines_fast_body(const LinesArgs& A, uint8_t* __restrict__ todo, uint32_t worker,
                                                       uint32_t n_workers, LinesFastLds& lds) {
  auto& s_hist = lds.hist;
  auto& s_red = lds.red;
  auto& s_ent = lds.ent;
  for (int i = threadIdx.x; i < kLinesHist; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
  const uint64_t n_groups = (A.n_lines + 4 * kWave - 1) / (4 * kWave);
  const uint64_t n_steps_all = (n_groups + kLinesPer - 1) / kLinesPer;
  const uint64_t n_steps = A.step_hi && A.step_hi < n_steps_all ? A.step_hi : n_steps_all;
  const uint64_t wave0 = A.step_lo + (uint64_t)worker * (kBlock / kWave) + (threadIdx.x >> 6);
  const uint64_t n_waves = (uint64_t)n_workers * (kBlock / kWave);
  const double chunks_per_line = A.n_newlines ? (double)A.cr.n_chunks / (double)A.n_newlines : 0.0;
  uint32_t n_ok32 = 0, min_rl32 = ~0u, max_rl32 = 0;  // (a wavefront's share of the records and a step's lengths fit)
  // the window of the NEXT step is requested a step ahead and looked at when that step begins (see k_stream_lines)
  unsigned long long nx_se = 0;
  uint32_t nx_loc = 0, nx_cnt = 0, nx_win = 0;
  bool nx_in = false;
  auto window_request = [&](uint64_t step) {
    const uint64_t g = (step < n_steps ? step : n_steps - 1) * kLinesPer;
    const uint64_t Rw = g ? 4 * g * kWave - 1 : 0;
    const double est = (double)Rw * chunks_per_line;
    uint32_t cw = est > 2.0 ? (uint32_t)(est - 2.0) : 0u;
    if (cw + kWave > A.cr.n_chunks) cw = A.cr.n_chunks > (uint32_t)kWave ? A.cr.n_chunks - kWave : 0u;
    const uint32_t cm = cw + (uint32_t)lane;
    nx_in = cm < A.cr.n_chunks;
    const uint32_t cl = nx_in ? cm : A.cr.n_chunks - 1;
    nx_se = A.cr.span_excl[cl / kScanSpan];
    nx_loc = A.cr.local[cl];
    nx_cnt = A.cr.counts[cl];
    nx_win = cw;
  };
  if (wave0 < n_steps) window_request(wave0);
  for (uint64_t step = wave0; step < n_steps; step += n_waves) {
    // lane j holds window chunk j: its first rank and its number of newlines
    const uint64_t w0 = nx_in ? nx_se + nx_loc : ~0ull;
    const uint32_t wc = nx_in ? nx_cnt : 0u;
    const uint32_t win0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx_win);  // (the same in every lane: scalar arithmetic below)
    const uint64_t rb = step * (uint64_t)(kWave * kLinesPer);   // the step's first record
    const uint64_t Rlo = 4 * rb - 1, Rhi = 4 * rb + 4 * kWave * kLinesPer;  // its ranks: [Rlo, Rhi)
    bool fast = rb > 0 && Rhi <= A.n_newlines && Rhi <= A.limit && Rhi <= A.line_cap &&
                (rb / kWave + kLinesPer) * kWave <= A.suspect_cap;
    // the chunks the ranks live in: from the last chunk whose first rank is <= Rlo to the last whose first rank is < Rhi
    // (first ranks do not decrease along the window; lanes beyond the image hold ~0)
    const unsigned long long b_lo = __ballot(w0 <= Rlo), b_hi = __ballot(w0 < Rhi);
    // (every value below is the same in all lanes; saying so - readfirstlane - keeps it in scalar registers and the
    // branches scalar: the compiler cannot see it through the comparisons that made it)
    const int jfirst = __builtin_amdgcn_readfirstlane(__builtin_popcountll(b_lo) - 1);
    const int jlast = __builtin_amdgcn_readfirstlane(__builtin_popcountll(b_hi) - 1);
    const uint32_t n_cov = (uint32_t)(jlast - jfirst + 1);
    if (fast && b_lo != 0) {
      // every covering chunk in ONE piece of 64 entries (reads of 100 bases and more), and the window holds the step's
      // last rank too
      const unsigned long long many = __ballot(lane >= jfirst && lane <= jlast && wc > (uint32_t)kWave);
      const uint64_t last_r0 = readlane64(w0, jlast);
      const uint32_t last_cnt = (uint32_t)__builtin_amdgcn_readlane((int)wc, jlast);
      fast = many == 0 && n_cov <= (uint32_t)kFastSlots && last_r0 + last_cnt >= Rhi;
    } else fast = false;
    if (!__builtin_amdgcn_readfirstlane((int)fast)) {
      if (lane == 0) todo[step] = 1;  // (its one owner writes it: no atomic - a counter all steps append to was 3.5 ms
                                      // of a 4.2 ms pass on reads of 30 - 150 bases, where no step qualifies)
      window_request(step + n_waves);
      continue;
    }
    // ---- every request of the step: the next step's window, then the staged entries chunk by chunk ----
    window_request(step + n_waves);
    // where chunk j's first entry goes in s_ent (32-bit, may be "negative": the chunk begins in front of the step's ranks)
    const uint32_t rel0 = (uint32_t)(w0 - Rlo);
    constexpr uint32_t kDump = (uint32_t)kFastRanks;  // a word of s_ent nobody reads: where entries outside the step go
    const uint16_t* const base0 = A.stage + (uint64_t)(win0 + (uint32_t)jfirst) * kStageCap;  // (uniform: scalar arithmetic)
    uint16_t raw[kFastSlots];
#pragma unroll
    for (int sl = 0; sl < kFastSlots; ++sl) {
      // (a chunk that is not needed asks for the first one again: no branch around a request)
      const uint32_t cj = (uint32_t)sl < n_cov ? (uint32_t)sl : 0u;
      const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)wc, jfirst + (int)cj);
      raw[sl] = base0[cj * (uint32_t)kStageCap + ((uint32_t)lane < cnt ? (uint32_t)lane : 0u)];
    }
#pragma unroll
    for (int sl = 0; sl < kFastSlots; ++sl) {
      const uint32_t cj = (uint32_t)sl < n_cov ? (uint32_t)sl : 0u;
      const uint32_t cnt = (uint32_t)sl < n_cov ? (uint32_t)__builtin_amdgcn_readlane((int)wc, jfirst + (int)cj) : 0u;
      const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)rel0, jfirst + (int)cj) + (uint32_t)lane;  // >= 2^31 in front of the step
      // no branch: an entry that is not wanted lands in the dump word
      s_ent[wv][((uint32_t)lane < cnt && idx < (uint32_t)kFastRanks) ? idx : kDump] = ((uint32_t)(jfirst + (int)cj) << 16) | (uint32_t)raw[sl];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- the line index, the checks, the statistics: every record of the step is complete ----
#pragma unroll
    for (int q = 0; q < kLinesPer; ++q) {
      const uint64_t g = step * kLinesPer + q;
      const uint64_t r = g * kWave + (uint64_t)lane;
      const uint64_t L0 = 4 * r;
      const uint32_t at = 4u * ((uint32_t)q * kWave + (uint32_t)lane);   // entry of rank 4r - 1
      const uint4 e03 = *reinterpret_cast<const uint4*>(&s_ent[wv][at]);  // (16-byte aligned: at is a multiple of 4)
      const uint32_t en[5] = {e03.x, e03.y, e03.z, e03.w, s_ent[wv][at + 4]};
      // positions relative to the window's first chunk, 32 bits (a step's ranks live in at most 16 chunks): the lengths
      // and the checks need no more; the 64-bit ends only where the index is stored
      uint32_t rel[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) rel[k] = ((en[k] >> 16) << 12) | (en[k] & 0xFFFu);
      if (!(A.ablate & 1) && !(A.no_index && L0 + 4 <= A.keep_from)) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const uint64_t wbase = (uint64_t)win0 * kChunkBytes;
        u64x2 lo2, hi2;
        lo2.x = wbase + rel[1]; lo2.y = wbase + rel[2]; hi2.x = wbase + rel[3]; hi2.y = wbase + rel[4];
        __builtin_nontemporal_store(lo2, reinterpret_cast<u64x2*>(A.line_end + L0));
        __builtin_nontemporal_store(hi2, reinterpret_cast<u64x2*>(A.line_end + L0 + 2));
      }
      if (A.index_only) continue;
      // header line: '@' and not empty; third line: exactly "+\n"  (what the entry BEFORE a line says about it)
      bool sus = !(((en[0] >> 12) & 3u) == kClsAt && !((en[0] >> 14) & 1u));
      sus |= !(((en[2] >> 12) & 3u) == kClsPlus && ((en[2] >> 14) & 1u));
      const uint32_t l0 = rel[1] - rel[0] - 1, l1 = rel[2] - rel[1] - 1, l2 = rel[3] - rel[2] - 1, l3 = rel[4] - rel[3] - 1;
      sus |= l1 < 1 || l1 != l3 || A.space != FQG_SPACE_SEQ;
      sus |= l0 + 1 > FQG_MAX_LABEL_LENGTH - 1 || l2 + 1 > FQG_MAX_LABEL_LENGTH - 1 ||
             l1 + 1 > FQG_MAX_READ_LENGTH - 1 || l3 + 1 > FQG_MAX_READ_LENGTH - 1;
      const bool counted = A.acc && l1 + 1 <= FQG_MAX_READ_LENGTH - 1;
      const uint32_t rl = l1 + 1;  // strlen(seq): the sequence line ends in '\n'
      if (counted) {
        ++n_ok32;
        min_rl32 = rl < min_rl32 ? rl : min_rl32;
        max_rl32 = rl > max_rl32 ? rl : max_rl32;
      }
      {  // the length histogram: the lanes that share the first counted lane's length add once, together (see k_stream_lines)
        const unsigned long long cm = __ballot(counted);
        if (cm) {
          const int first = __builtin_ctzll(cm);
          const uint32_t rl0 = (uint32_t)__builtin_amdgcn_readlane((int)rl, first);
          const bool with_first = counted && rl == rl0;
          const unsigned long long same = __ballot(with_first);
          if (lane == first) {
            if (rl < (uint32_t)kLinesHist) atomicAdd(&s_hist[rl], (uint32_t)__builtin_popcountll(same));
            else atomicAdd(&A.hist[rl], (unsigned long long)A.weight * (unsigned long long)__builtin_popcountll(same));
          }
          if (counted && !with_first) {
            if (rl < (uint32_t)kLinesHist) atomicAdd(&s_hist[rl], 1u);
            else atomicAdd(&A.hist[rl], (unsigned long long)A.weight);
          }
        }
      }
      const unsigned long long sm = __ballot(sus);
      if (lane == 0 && sm) {
        const uint64_t w = g * 2;
        if ((uint32_t)sm) A.suspect_bits[w] = (uint32_t)sm;
        if ((uint32_t)(sm >> 32)) A.suspect_bits[w + 1] = (uint32_t)(sm >> 32);
      }
    }
    __builtin_amdgcn_wave_barrier();  // (s_ent is rewritten by the next step)
  }
  if (!A.acc) return;
  unsigned long long n_ok = n_ok32, min_rl = min_rl32 == ~0u ? ~0ull : (unsigned long long)min_rl32, max_rl = max_rl32;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    n_ok += __shfl_down(n_ok, d, 64);
    const unsigned long long a = __shfl_down(min_rl, d, 64), b = __shfl_down(max_rl, d, 64);
    min_rl = a < min_rl ? a : min_rl;
    max_rl = b > max_rl ? b : max_rl;
  }
  if (lane == 0) {
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
      atomicAdd(&A.acc->num_rds, n_ok * A.weight);
      if (min_rl < A.acc->min_rl) atomicMin(&A.acc->min_rl, min_rl);
      if (max_rl > A.acc->max_rl) atomicMax(&A.acc->max_rl, max_rl);
    }
  }
  for (int i = threadIdx.x; i < kLinesHist; i += kBlock) {
    const uint32_t cc = s_hist[i];
    if (cc) atomicAdd(&A.hist[i], (unsigned long long)cc * A.weight);
  }
}

__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_stream_lines_fast(LinesArgs A, uint8_t* __restrict__ todo) {
  __shared__ LinesFastLds lds;
  stream_lines_fast_body(A, todo, blockIdx.x, gridDim.x, lds);
}

// ------------------------------------------------------------------------------------------
// Pass 1 of one part of the image and the line workers of the part BEFORE it in ONE kernel (round 6).  Pass 1 is bound by
// vector-ALU cycles, the line kernel by the latency of its dependent loads: side by side they hide each other (two
// streams with priorities: 7.97 -> 7.58 ms, profiles/r04c_kbench_overlap.txt).  As one launch the overlap needs no second
// stream, no host in between, no priorities: the first `line_workers` workgroups are the persistent line workers - they
// start first and keep one slot per CU -, the others are pass-1 workgroups of chunks [chunk_first, n_chunks).  The line
// workers take what they need from DEVICE memory: the newline count behind the parts scanned so far (CallState::
// part_newlines, k_scan_b) gives their steps - every step whose ranks are all staged - and the flags that send an image
// to the two-pass path (NUL, CR, high bytes, a chunk with too many newlines) make them do nothing.  Their statistics go
// to accumulators the caller merges once the whole image has passed (k_acc_merge): a later part can still raise a flag.
// ------------------------------------------------------------------------------------------
template <int NAMES>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_stream_pass1_lines(
    const uint8_t* __restrict__ img, uint64_t n, uint32_t chunk_first, uint32_t n_chunks, StreamOut o, CallState* __restrict__ cs,
    LinesArgs A, uint8_t* __restrict__ todo, uint32_t line_workers, uint32_t part /* whose lines: >= 0 */, NameCapture nc) {
  union Lds {
    Pass1Lds<NAMES> p1;
    LinesFastLds lf;
  };
  __shared__ Lds lds;
  if (blockIdx.x >= line_workers) {
    stream_pass1_body<0u, NAMES>(img, n, n_chunks, o, cs, nc, chunk_first / (kBlock / kWave) + (blockIdx.x - line_workers), lds.p1);
    return;
  }
  // ---- a line worker: the steps of part `part`, from what the scans have left in the call state ----
  if (__hip_atomic_load(&cs->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (kFlagNul | kFlagCr | kFlagHigh | kFlagStageOverflow)) return;
  const unsigned long long R = cs->part_newlines[part], R0 = part ? cs->part_newlines[part - 1] : 0ull;
  const double per_chunk = (double)R / (double)(chunk_first ? chunk_first : 1u);
  if (per_chunk > 60.0 || per_chunk < 32.0) return;  // (the host decides the same way for the whole image: frame_stream)
  A.n_newlines = R;
  A.n_lines = R;
  A.limit = R & ~3ull;
  A.cr.n_chunks = chunk_first;
  A.step_lo = R0 / (uint64_t)(4 * kWave * kLinesPer);
  A.step_hi = R / (uint64_t)(4 * kWave * kLinesPer);  // full steps only: every rank of theirs has a staged entry
  if (blockIdx.x == 0 && threadIdx.x == 0) cs->lines_done_steps = A.step_hi;
  if (A.step_hi <= A.step_lo) return;
  stream_lines_fast_body(A, todo, blockIdx.x, line_workers, lds.lf);
}

// the statistics of the line workers -> the caller's accumulator (discard: only the clearing), and the scratch cleared
__global__ __launch_bounds__(kBlock) void k_acc_merge(AccState* __restrict__ src, unsigned long long* __restrict__ src_hist,
                                                      AccState* __restrict__ dst, unsigned long long* __restrict__ dst_hist,
                                                      int discard) {
  __shared__ unsigned long long s_lo, s_hi;
  if (threadIdx.x == 0) {
    s_lo = src->min_rl;
    s_hi = src->max_rl;
  }
  __syncthreads();
  const unsigned long long lo = s_lo, hi = s_hi;
  if (lo <= hi)
    for (unsigned long long i = lo + (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i <= hi; i += (unsigned long long)gridDim.x * kBlock) {
      const unsigned long long v = src_hist[i];
      if (v) {
        if (!discard && dst_hist) atomicAdd(&dst_hist[i], v);
        src_hist[i] = 0;
      }
    }
  if (!discard && dst && blockIdx.x == 0 && threadIdx.x == 0 && src->num_rds) {
    atomicAdd(&dst->num_rds, src->num_rds);
    atomicMin(&dst->min_rl, src->min_rl);
    atomicMax(&dst->max_rl, src->max_rl);
  }
}
__global__ void k_acc_scratch_reset(AccState* __restrict__ a) {
  a->num_rds = 0;
  a->min_rl = FQG_MAX_READ_LENGTH;
  a->max_rl = 0;
  a->min_qbyte = 255;
  a->max_qbyte = 0;
}

// After every marking kernel: the suspect bitmap -> the list the exact validator walks; the image's quality range
// -> the statistics (what k_records_fast does at its end on the other paths)
__global__ __launch_bounds__(kBlock) void k_suspect_list(const uint32_t* __restrict__ bits, uint64_t n_records,
                                                         unsigned long long* __restrict__ list,
                                                         unsigned long long list_cap,
                                                         unsigned long long* __restrict__ list_count,
                                                         AccState* __restrict__ acc, const CallState* __restrict__ cs) {
  const uint64_t n_words = (n_records + 31) / 32;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  // one reservation per wavefront and step (a file whose every read is a suspect - lower-case bases, say - would
  // otherwise add to one address a hundred million times)
  const int lane = lane_id();
  for (uint64_t w0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); w0 < n_words; w0 += stride) {
    const uint64_t w = w0 + (uint64_t)lane;
    uint32_t m = w < n_words ? bits[w] : 0u;
    if (w * 32 + 32 > n_records) m &= w * 32 < n_records ? (uint32_t)((1ull << (n_records - w * 32)) - 1ull) : 0u;
    const uint32_t cnt = (uint32_t)__popc(m);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (!total) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(list_count, (unsigned long long)total);
    base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    unsigned long long at = base + (incl - cnt);
    while (m) {
      const uint32_t j = (uint32_t)__builtin_ctz(m);
      m &= m - 1;
      if (at < list_cap) list[at] = w * 32 + j;
      ++at;
    }
  }
  if (acc && blockIdx.x == 0 && threadIdx.x == 0 && cs->qmin_byte <= cs->qmax_byte) {
    atomicMin(&acc->min_qbyte, cs->qmin_byte);
    atomicMax(&acc->max_qbyte, cs->qmax_byte);
  }
}

// queued suspect byte positions -> records
__global__ __launch_bounds__(kBlock) void k_stream_queue(const unsigned long long* __restrict__ queue,
                                                         unsigned long long queue_cap,
                                                         const uint64_t* __restrict__ line_end, uint64_t n_lines,
                                                         uint64_t n_records, SuspectMap suspect,
                                                         CallState* __restrict__ cs) {
  unsigned long long cnt = cs->queue_count;
  if (cnt > queue_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&cs->flags, kFlagQueueOverflow);
    cnt = queue_cap;
  }
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < cnt; i += stride) {
    const uint64_t pos = queue[i];
    // first line whose end is >= pos
    uint64_t lo = 0, hi = n_lines;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (line_end[mid] < pos) lo = mid + 1;
      else hi = mid;
    }
    const uint64_t rec = lo >> 2;
    if (rec < n_records) mark_suspect(suspect, rec);
  }
}

}  // namespace fqg
