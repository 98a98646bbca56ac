// fqg_bamtags_kernels.hip - the alignment loop of bam_add_tags (reference src/bam_add_tags.c:250-294) on gfx950:
// the barcodes that fastq_pre_barcodes wrote into the read names (STAGS_CELL=.._UMI=.._SAMPLE=.._ETAGS_, parsed by
// get_barcodes :43-99) become aux tags at the end of every alignment record (bam_aux_append: tag, 'Z', value, NUL):
// RX or UB (--10x), CR, BC, and with --tx the reference's name (tx) and its gene (GX, --tx_2_gx).
//
// Input: the inflated BAM stream and the offset of every alignment (fqg_bam_index_records), as for bam_umi_count.
//   k_bt_tile<false>  one wavefront per tile of T consecutive alignments, whose bytes are ONE span of the stream: the
//               span is copied to LDS with 16-byte loads, every lane runs get_barcodes on the name of ITS alignment
//               there and writes the size the record grows to
//   scan        64-bit exclusive prefix of the new record sizes (k_scan64_a / _b)
//   k_bt_tile<true>   the same tiles again: the lane parses the name once more (cheaper than keeping what pass 1
//               found), rebuilds its record in an LDS image of the tile's OUTPUT span (block_size patched, tags
//               appended), and the image goes out with 16-byte stores.
//   Tiles that do not fit LDS (long reads) read the stream itself and are copied record by record.
// Everything else of the program is host work: BGZF, the header (copied verbatim), the transcript -> gene map.
#include "fqg_device.h"

namespace fqg {

constexpr int kBtMaxBarcode = 50;  // MAX_BARCODE_LENGTH (src/fastq.h): the arrays get_barcodes fills
constexpr uint32_t kBtNone = 0xFFFFFFFFu;
constexpr uint32_t kBtInCap = 20 * 1024, kBtOutCap = 28 * 1024;  // LDS bytes per wavefront

struct BtParams {
  int32_t tenx, tx_tag;
  uint32_t n_targets;
  const uint32_t* tx_off;  // per reference: its name in `names` ...
  const uint32_t* tx_len;
  const uint32_t* gx_off;  // ... and its gene (tx_2_gx), length kBtNone when there is none
  const uint32_t* gx_len;
  const uint8_t* names;
};
struct BtRec {
  uint32_t off[3];  // umi, cell, sample: offset of the value from the start of the record (its block_size field)
  uint8_t len[3];
  uint8_t ok;       // get_barcodes returned 1
};
struct BtCall {
  unsigned long long first_finding;  // min (record << 8 | code)
  unsigned long long n_tagged[64];   // 64 copies picked by tile (every wavefront adds: one address would serialise them)
};

// get_barcodes (src/bam_add_tags.c:43-99) on the C string at buf[s..]; the scans for '_' are scans of memory (they do
// not stop at the NUL), `end` = end of the record.  1 tags found, 0 not a tagged name, -1 the reference would read
// behind the record or write behind its 50-byte arrays.  The name is read 8 bytes at a time: word(i) = bytes i .. i+7
// of the stream (what lies behind `end` is read, never looked at).
template <class Word>
__device__ __forceinline__ int bt_get_barcodes(Word word, uint64_t s, uint64_t end, uint64_t rec0, BtRec& r) {
  // the || chains of the reference stop at the first difference: a difference inside the record is "no" even when the
  // literal would run past the record's end
  auto expect = [&](uint64_t i, uint64_t lit, int n) {
    const uint64_t room = end > i ? end - i : 0;
    const int cmp = room < (uint64_t)n ? (int)room : n;
    const uint64_t mask = cmp >= 8 ? ~0ull : (1ull << (8 * cmp)) - 1ull;
    if ((word(i) ^ lit) & mask) return 0;
    return cmp < n ? -1 : 1;
  };
  auto value = [&](uint64_t i, int slot, uint64_t* next) {
    uint64_t z = i;
    for (;;) {
      if (z >= end) return -1;
      uint64_t m = bytes_eq(word(z), (uint8_t)'_');
      if (end - z < 8) m &= (1ull << (8 * (end - z))) - 1ull;
      if (m) {
        z += (uint64_t)(__builtin_ctzll(m) >> 3);
        break;
      }
      z += 8;
    }
    if (z - i >= (uint64_t)kBtMaxBarcode) return -1;
    r.off[slot] = (uint32_t)(i - rec0);
    r.len[slot] = (uint8_t)(z - i);
    *next = z + 1;
    return 1;
  };
  constexpr uint64_t kStags = 0x5F5347415453ull;        // "STAGS_"
  constexpr uint64_t kCell = 0x3D4C4C4543ull;           // "CELL="
  constexpr uint64_t kUmi = 0x3D494D55ull;              // "UMI="
  constexpr uint64_t kSample = 0x3D454C504D4153ull;     // "SAMPLE="
  int e;
  uint64_t i = s;
  if ((e = expect(i, kStags, 6)) != 1) return e;
  i += 6;
  if ((e = expect(i, kCell, 5)) != 1) return e;
  if ((e = value(i + 5, 1, &i)) != 1) return e;
  if ((e = expect(i, kUmi, 4)) != 1) return e;
  if ((e = value(i + 4, 0, &i)) != 1) return e;
  if ((e = expect(i, kSample, 7)) != 1) return e;
  if ((e = value(i + 7, 2, &i)) != 1) return e;
  return 1;
}

typedef __attribute__((address_space(3))) uint8_t* BtLds;
typedef uint64_t __attribute__((aligned(1), may_alias)) bt_u64;
typedef uint32_t __attribute__((aligned(1), may_alias)) bt_u32;
__device__ __forceinline__ uint64_t bt_ld8(BtLds p) { return *(__attribute__((address_space(3))) bt_u64*)p; }
__device__ __forceinline__ void bt_st8(BtLds p, uint64_t v) { *(__attribute__((address_space(3))) bt_u64*)p = v; }
__device__ __forceinline__ void bt_st4(BtLds p, uint32_t v) { *(__attribute__((address_space(3))) bt_u32*)p = v; }
// n bytes inside LDS, exactly n (what lies behind dst belongs to another lane)
__device__ __forceinline__ void bt_copy(BtLds dst, BtLds src, uint32_t n) {
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8) bt_st8(dst + i, bt_ld8(src + i));
  for (; i < n; ++i) dst[i] = src[i];
}

struct BtTiles {
  const uint8_t* buf;
  uint64_t nbytes;
  const unsigned long long* offs;
  uint32_t n, T;                  // alignments; per tile
  uint32_t in_cap, out_cap;       // LDS bytes of the two areas (dynamic shared memory: in_cap + out_cap + 64)
  uint32_t* new_size;             // pass 1 writes, pass 2 reads
  const unsigned long long* out_local;  // exclusive prefix of new_size: local part + span sums
  const unsigned long long* out_sums;
  uint8_t* out;
  BtParams P;
  BtCall* call;
};

// what the record at stream offset o grows by (and its parsed name): shared by both passes
template <class Word>
__device__ __forceinline__ uint32_t bt_growth(const BtParams& P, Word word, uint64_t o, uint32_t block, int32_t tid, BtRec& r,
                                              uint32_t* finding) {
  r.off[0] = r.off[1] = r.off[2] = 0;
  r.len[0] = r.len[1] = r.len[2] = 0;
  r.ok = 0;
  *finding = 0;
  uint32_t add = 0;
  const int g = bt_get_barcodes(word, o + 36, o + 4 + block, o, r);
  if (g < 0) *finding = (uint32_t)FQG_E_TAGS_NAME;
  if (g == 1) {
    r.ok = 1;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (r.len[k]) add += 4u + r.len[k];
    if (P.tx_tag && tid >= 0) {
      if ((uint32_t)tid >= P.n_targets) *finding = (uint32_t)FQG_E_TAGS_TID;
      else {
        add += 4u + P.tx_len[tid];
        if (P.gx_len[tid] != kBtNone) add += 4u + P.gx_len[tid];
      }
    }
  } else {
    r.len[0] = r.len[1] = r.len[2] = 0;  // (a name that fails half-way gets no tag at all)
  }
  return add;
}

// the tags of one record, written byte by byte through put(byte)
template <class Src, class Put>
__device__ __forceinline__ void bt_tags(const BtParams& P, const BtRec& r, int32_t tid, Src src, Put put) {
  auto z = [&](char a, char b, uint32_t off, uint32_t len) {
    put((uint8_t)a);
    put((uint8_t)b);
    put((uint8_t)'Z');
    for (uint32_t k = 0; k < len; ++k) put(src(off + k));
    put((uint8_t)0);
  };
  if (!r.ok) return;
  if (r.len[0]) z(P.tenx ? 'U' : 'R', P.tenx ? 'B' : 'X', r.off[0], r.len[0]);  // GET_UMI_TAG, src/sam_tags.h:40-47
  if (r.len[1]) z('C', 'R', r.off[1], r.len[1]);
  if (r.len[2]) z('B', 'C', r.off[2], r.len[2]);
  if (P.tx_tag && tid >= 0 && (uint32_t)tid < P.n_targets) {
    auto name = [&](char a, char b, uint32_t off, uint32_t len) {
      put((uint8_t)a);
      put((uint8_t)b);
      put((uint8_t)'Z');
      for (uint32_t k = 0; k < len; ++k) put(P.names[off + k]);
      put((uint8_t)0);
    };
    name('t', 'x', P.tx_off[tid], P.tx_len[tid]);
    if (P.gx_len[tid] != kBtNone) name('G', 'X', P.gx_off[tid], P.gx_len[tid]);
  }
}

// One wavefront per tile of A.T consecutive alignments, one lane per alignment.  EMIT = false: the new record sizes
// (pass 1).  EMIT = true: the records with their tags (pass 2, after the scan of the sizes).  Both read the tile's
// bytes once, as one span, through LDS; a tile that does not fit there (long reads) works on the stream itself.
template <bool EMIT>
__global__ __launch_bounds__(kWave) void k_bt_tile(BtTiles A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_bt[];
  uint8_t* s_in = s_bt;
  uint8_t* s_out = s_bt + A.in_cap + 32;
  const int lane = (int)threadIdx.x;
  const uint32_t i0 = blockIdx.x * A.T;
  if (i0 >= A.n) return;
  const uint32_t Tn = A.n - i0 < A.T ? A.n - i0 : A.T;
  const bool valid = (uint32_t)lane < Tn;
  const uint32_t i = i0 + (valid ? (uint32_t)lane : Tn - 1);
  const uint64_t in_off = A.offs[i];
  const uint64_t in0 = rfl64(in_off);
  // the tile's span: up to the end of its last record, which only the record itself tells
  uint64_t out_off = 0, out0 = 0, out_end = 0;
  uint32_t out_len = 0;
  if (EMIT) {
    out_len = A.new_size[i];
    out_off = A.out_local[i] + A.out_sums[i / kScan64Span];
    out0 = rfl64(out_off);
    out_end = rl64(out_off + out_len, (int)Tn - 1);
  }
  const uint32_t in_skew = (uint32_t)(in0 & 15u);  // (the stream starts at a 16-byte boundary)
  const uint32_t out_skew = EMIT ? (uint32_t)((uintptr_t)(A.out + out0) & 15u) : 0u;
  // bound of the span before the last record's length is known: its start + 4 (the length field) is inside for sure;
  // the true end follows from the staged length field
  const uint64_t last_off = rl64(in_off, (int)Tn - 1);
  uint32_t last_block = 0;
  if (last_off + 4 <= A.nbytes) __builtin_memcpy(&last_block, A.buf + last_off, 4);
  const uint64_t in_end = last_off + 4ull + last_block;
  const bool fits = in_skew + (in_end - in0) + 16 <= (uint64_t)A.in_cap &&
                    (!EMIT || out_skew + (out_end - out0) + 16 <= (uint64_t)A.out_cap);
  uint32_t block = 0;
  int32_t tid = -1;
  BtRec r;
  uint32_t finding = 0, add = 0;
  if (fits) {
    const uint32_t span = in_skew + (uint32_t)(in_end - in0);
    const uint64_t base = in0 - in_skew;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t units = (span + 15u) >> 4;
    // whole 16-byte units that lie inside the stream: 8 loads in flight per lane; the last unit(s) of the stream byte by byte
    const uint64_t safe_units = A.nbytes > base ? (A.nbytes - base) >> 4 : 0;
    for (uint32_t u0 = 0; u0 < units; u0 += 8 * kWave) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        uint32_t u = u0 + j * kWave + (uint32_t)lane;
        u = u < units ? u : units - 1;
        if ((uint64_t)u < safe_units) v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A.buf + base + 16ull * u));
        else {
          u32x4 t = {0, 0, 0, 0};
          for (uint64_t b = 0; base + 16ull * u + b < A.nbytes && b < 16; ++b) reinterpret_cast<uint8_t*>(&t)[b] = A.buf[base + 16ull * u + b];
          v[j] = t;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t u = u0 + j * kWave + (uint32_t)lane;
        if (u < units) *reinterpret_cast<u32x4*>(s_in + 16u * u) = v[j];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const BtLds img = (BtLds)s_in;  // img[k] = stream byte base + k
    block = *(__attribute__((address_space(3))) bt_u32*)(img + (uint32_t)(in_off - base));
    tid = (int32_t) * (__attribute__((address_space(3))) bt_u32*)(img + (uint32_t)(in_off - base) + 4);
    add = bt_growth(A.P, [&](uint64_t p) { return bt_ld8(img + (uint32_t)(p - base)); }, in_off, block, tid, r, &finding);
    if (EMIT) {
      if (valid) {
        const BtLds src = img + (uint32_t)(in_off - base);
        BtLds dst = (BtLds)s_out + out_skew + (uint32_t)(out_off - out0);
        const uint32_t in_len = 4u + block;
        bt_copy(dst, src, in_len);
        bt_st4(dst, out_len - 4u);  // block_size
        BtLds w = dst + in_len;
        // the tags, values 8 bytes per LDS access (bt_tags, byte by byte, is the form for the slow path)
        if (r.ok) {
          auto head = [&](char a, char b) {
            w[0] = (uint8_t)a;
            w[1] = (uint8_t)b;
            w[2] = (uint8_t)'Z';
            w += 3;
          };
          auto from_record = [&](char a, char b, uint32_t off, uint32_t len) {
            head(a, b);
            bt_copy(w, src + off, len);
            w[len] = 0;
            w += len + 1u;
          };
          auto from_names = [&](char a, char b, uint32_t off, uint32_t len) {  // (the table has 8 readable bytes behind it)
            head(a, b);
            const uint8_t* g = A.P.names + off;
            uint32_t k = 0;
            for (; k + 8 <= len; k += 8) {
              uint64_t v;
              __builtin_memcpy(&v, g + k, 8);
              bt_st8(w + k, v);
            }
            if (k < len) {
              uint64_t v;
              __builtin_memcpy(&v, g + k, 8);
              for (; k < len; ++k, v >>= 8) w[k] = (uint8_t)v;
            }
            w[len] = 0;
            w += len + 1u;
          };
          if (r.len[0]) from_record(A.P.tenx ? 'U' : 'R', A.P.tenx ? 'B' : 'X', r.off[0], r.len[0]);  // GET_UMI_TAG
          if (r.len[1]) from_record('C', 'R', r.off[1], r.len[1]);
          if (r.len[2]) from_record('B', 'C', r.off[2], r.len[2]);
          if (A.P.tx_tag && tid >= 0 && (uint32_t)tid < A.P.n_targets) {
            from_names('t', 'x', A.P.tx_off[tid], A.P.tx_len[tid]);
            if (A.P.gx_len[tid] != kBtNone) from_names('G', 'X', A.P.gx_off[tid], A.P.gx_len[tid]);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      emit_flush(s_out, out_skew, (uint32_t)(out_end - out0), A.out + out0, lane);
    }
  } else {
    __builtin_memcpy(&block, A.buf + in_off, 4);
    __builtin_memcpy(&tid, A.buf + in_off + 4, 4);
    const uint8_t* gb = A.buf;
    const uint64_t nbytes = A.nbytes;
    add = bt_growth(A.P,
                    [&](uint64_t p) {
                      uint64_t v = 0;
                      if (p + 8 <= nbytes) __builtin_memcpy(&v, gb + p, 8);
                      else
                        for (uint64_t k = 0; p + k < nbytes && k < 8; ++k) v |= (uint64_t)gb[p + k] << (8 * k);
                      return v;
                    },
                    in_off, block, tid, r, &finding);
    if (EMIT) {
      // long records: straight from the stream to the output, one record after the other
      const uint32_t in_len = 4u + block;
      for (uint32_t k = 0; k < Tn; ++k) {
        const uint64_t so = rl64(in_off, (int)k), dofs = rl64(out_off, (int)k);
        const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)in_len, (int)k);
        const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)out_len, (int)k);
        const uint8_t* src = A.buf + so;
        uint8_t* dst = A.out + dofs;
        const uint32_t nb = dl - 4u;
        for (uint32_t j = (uint32_t)lane; j < sl; j += kWave) dst[j] = j < 4 ? (uint8_t)(nb >> (8 * j)) : src[j];
        if ((uint32_t)lane == k) {
          uint8_t* w = dst + sl;
          bt_tags(A.P, r, tid, [&](uint32_t o) { return src[o]; }, [&](uint8_t b) { *w++ = b; });
        }
      }
    }
  }
  if (!EMIT) {
    if (in_off + 4ull + block > A.nbytes) finding = (uint32_t)FQG_E_TAGS_NAME;  // (a block_size that leaves the stream: nothing is emitted)
    if (valid) {
      A.new_size[i] = 4u + block + add;
      if (finding) atomicMin(&A.call->first_finding, ((unsigned long long)i << 8) | finding);
    }
    const unsigned long long m = __ballot(valid && r.ok);
    if (lane == 0 && m) atomicAdd(&A.call->n_tagged[blockIdx.x & 63], (unsigned long long)__popcll(m));
  }
}

}  // namespace fqg
