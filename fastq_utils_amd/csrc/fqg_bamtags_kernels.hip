// fqg_bamtags_kernels.hip - the alignment loop of bam_add_tags (reference src/bam_add_tags.c:250-294) on gfx950:
// the barcodes that fastq_pre_barcodes wrote into the read names (STAGS_CELL=.._UMI=.._SAMPLE=.._ETAGS_, parsed by
// get_barcodes :43-99) become aux tags at the end of every alignment record (bam_aux_append: tag, 'Z', value, NUL):
// RX or UB (--10x), CR, BC, and with --tx the reference's name (tx) and its gene (GX, --tx_2_gx).
//
// Input: the inflated BAM stream and the offset of every alignment (fqg_bam_index_records), as for bam_umi_count.
//   k_bt_plan   one thread per alignment: get_barcodes on the name, the bytes the record grows by
//   scan        64-bit exclusive prefix of the new record sizes (k_scan64_a / _b)
//   k_bt_emit   one wavefront per tile of T consecutive alignments: their bytes are one span of the input and one
//               span of the output.  The input span is copied to LDS with 16-byte loads, every lane rebuilds ITS
//               record in an LDS image of the output span (block_size patched, tags appended), and the image goes
//               out with 16-byte stores.  Tiles that do not fit (long reads) are copied record by record.
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
  unsigned long long n_tagged;
};

// get_barcodes (src/bam_add_tags.c:43-99) on the C string at buf[s..]; the scans for '_' are scans of memory (they do
// not stop at the NUL), `end` = end of the record.  1 tags found, 0 not a tagged name, -1 the reference would read
// behind the record or write behind its 50-byte arrays.
__device__ __forceinline__ int bt_get_barcodes(const uint8_t* __restrict__ buf, uint64_t s, uint64_t end, uint64_t rec0,
                                               BtRec& r) {
  auto expect = [&](uint64_t i, const char* lit, int n) {  // 1 all equal, 0 a difference, -1 out of the record first
    for (int k = 0; k < n; ++k) {
      if (i + k >= end) return -1;
      if (buf[i + k] != (uint8_t)lit[k]) return 0;
    }
    return 1;
  };
  auto value = [&](uint64_t i, int slot, uint64_t* next) {
    uint64_t z = i;
    for (;;) {
      if (z >= end) return -1;
      if (buf[z] == '_') break;
      ++z;
    }
    if (z - i >= (uint64_t)kBtMaxBarcode) return -1;
    r.off[slot] = (uint32_t)(i - rec0);
    r.len[slot] = (uint8_t)(z - i);
    *next = z + 1;
    return 1;
  };
  int e;
  uint64_t i = s;
  if ((e = expect(i, "STAGS_", 6)) != 1) return e;
  i += 6;
  if ((e = expect(i, "CELL=", 5)) != 1) return e;
  if ((e = value(i + 5, 1, &i)) != 1) return e;
  if ((e = expect(i, "UMI=", 4)) != 1) return e;
  if ((e = value(i + 4, 0, &i)) != 1) return e;
  if ((e = expect(i, "SAMPLE=", 7)) != 1) return e;
  if ((e = value(i + 7, 2, &i)) != 1) return e;
  return 1;
}

__global__ __launch_bounds__(kBlock) void k_bt_plan(const uint8_t* __restrict__ buf, const unsigned long long* __restrict__ offs,
                                                    uint32_t n, BtParams P, BtRec* __restrict__ rec,
                                                    uint32_t* __restrict__ new_size, BtCall* __restrict__ call) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  bool tagged = false;
  if (i < n) {
    const uint64_t o = offs[i];
    uint32_t block;
    int32_t tid;
    __builtin_memcpy(&block, buf + o, 4);
    __builtin_memcpy(&tid, buf + o + 4, 4);
    BtRec r;
    r.off[0] = r.off[1] = r.off[2] = 0;
    r.len[0] = r.len[1] = r.len[2] = 0;
    r.ok = 0;
    uint32_t add = 0;
    const int g = bt_get_barcodes(buf, o + 36, o + 4 + block, o, r);
    if (g < 0) atomicMin(&call->first_finding, ((unsigned long long)i << 8) | (unsigned)FQG_E_TAGS_NAME);
    if (g == 1) {
      r.ok = 1;
      tagged = true;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (r.len[k]) add += 4u + r.len[k];
      if (P.tx_tag && tid >= 0) {
        if ((uint32_t)tid >= P.n_targets) atomicMin(&call->first_finding, ((unsigned long long)i << 8) | (unsigned)FQG_E_TAGS_TID);
        else {
          add += 4u + P.tx_len[tid];
          if (P.gx_len[tid] != kBtNone) add += 4u + P.gx_len[tid];
        }
      }
    } else {
      r.len[0] = r.len[1] = r.len[2] = 0;  // (a name that fails half-way gets no tag at all)
    }
    rec[i] = r;
    new_size[i] = 4u + block + add;
  }
  const unsigned long long m = __ballot(tagged);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&call->n_tagged, (unsigned long long)__popcll(m));
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

struct BtEmit {
  const uint8_t* buf;
  uint64_t nbytes;
  const unsigned long long* offs;
  uint32_t n, T;
  const BtRec* rec;
  const uint32_t* new_size;
  const unsigned long long* out_local;  // exclusive prefix of new_size: local part + span sums
  const unsigned long long* out_sums;
  uint8_t* out;
  BtParams P;
};

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

__global__ __launch_bounds__(kWave) void k_bt_emit(BtEmit A) {
  __shared__ __attribute__((aligned(16))) uint8_t s_in[kBtInCap + 32];
  __shared__ __attribute__((aligned(16))) uint8_t s_out[kBtOutCap + 32];
  const int lane = (int)threadIdx.x;
  const uint32_t i0 = blockIdx.x * A.T;
  if (i0 >= A.n) return;
  const uint32_t Tn = A.n - i0 < A.T ? A.n - i0 : A.T;
  const bool valid = (uint32_t)lane < Tn;
  const uint32_t i = i0 + (valid ? (uint32_t)lane : Tn - 1);
  const uint64_t in_off = A.offs[i];
  uint32_t block;
  int32_t tid;
  __builtin_memcpy(&block, A.buf + in_off, 4);
  __builtin_memcpy(&tid, A.buf + in_off + 4, 4);
  const uint32_t in_len = 4u + block, out_len = A.new_size[i];
  const uint64_t out_off = A.out_local[i] + A.out_sums[i / kScan64Span];
  const BtRec r = A.rec[i];
  const uint64_t in0 = rfl64(in_off), out0 = rfl64(out_off);
  const uint64_t in_end = rl64(in_off + in_len, (int)Tn - 1), out_end = rl64(out_off + out_len, (int)Tn - 1);
  const uint32_t in_skew = (uint32_t)(in0 & 15u);  // (the stream starts at a 16-byte boundary)
  const uint32_t out_skew = (uint32_t)((uintptr_t)(A.out + out0) & 15u);
  const bool fits = in_skew + (in_end - in0) + 16 <= (uint64_t)kBtInCap && out_skew + (out_end - out0) + 16 <= (uint64_t)kBtOutCap;
  if (fits) {
    const uint32_t span = in_skew + (uint32_t)(in_end - in0);
    const uint64_t base = in0 - in_skew;
    for (uint32_t u = (uint32_t)lane * 16u; u < span; u += 16u * kWave) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 v = {0, 0, 0, 0};
      if (base + u + 16 <= A.nbytes) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A.buf + base + u));
      else
        for (uint64_t b = 0; base + u + b < A.nbytes && b < 16; ++b) reinterpret_cast<uint8_t*>(&v)[b] = A.buf[base + u + b];
      *reinterpret_cast<u32x4*>(s_in + u) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const BtLds src = (BtLds)s_in + in_skew + (uint32_t)(in_off - in0);
      BtLds dst = (BtLds)s_out + out_skew + (uint32_t)(out_off - out0);
      bt_copy(dst, src, in_len);
      bt_st4(dst, out_len - 4u);  // block_size
      BtLds w = dst + in_len;
      bt_tags(A.P, r, tid, [&](uint32_t o) { return (uint8_t)src[o]; }, [&](uint8_t b) { *w++ = b; });
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    emit_flush(s_out, out_skew, (uint32_t)(out_end - out0), A.out + out0, lane);
  } else {
    // long records: straight from the stream to the output, one record after the other
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

}  // namespace fqg
