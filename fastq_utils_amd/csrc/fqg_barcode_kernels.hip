// fqg_barcode_kernels.hip - the per-read transform of fastq_pre_barcodes on the GPU
// (reference src/fastq_pre_barcodes.c:594-727): lock-step records of up to five files, read
// names must agree, UMI / cell / sample substrings are cut out (with an optional minimum base
// quality), the header gets the STAGS_..._ETAGS_ prefix, reads are optionally sliced, and the
// result is emitted as FASTQ text (one or two outputs) or as SAM lines.
//
// Three launches per batch of iterations:
//   k_bc_plan   one thread per iteration: status (keep / discarded / finding) and the exact
//               number of output bytes of every output
//   scan        64-bit exclusive prefix of the byte counts -> where each iteration writes
//   k_bc_emit   one wavefront per kept iteration: cooperative byte copies into the output image
// Inputs are framed images (line index from fqg_validate); nothing is re-parsed on the host.
#include "fqg_device.h"

namespace fqg {

constexpr int kBcFiles = 6;  // index 1..5 as in the reference (READ1, READ2, INDEX1..3)
enum : uint8_t { kBcKeep = 0, kBcDiscardShort = 1, kBcDiscardQual = 2, kBcFinding = 3,
                 kBcKeepBig = 4 };  // kept, but records / text exceed the LDS buffers of k_bc_emit
constexpr int kEmitBuf = 2048;   // output text per wavefront (k_bc_emit)
constexpr int kEmitSrc = 2560;   // staged input records per wavefront

struct BcFile {
  FrameView fv;
  uint64_t first;  // record of iteration 0
  uint32_t step;   // records per iteration (2 for the two interleaved references)
  uint32_t add;    // 1 for the second interleaved reference
  int32_t present;
  int32_t fmt;     // read-name format of this file
};

struct BcParams {
  BcFile f[kBcFiles];
  int32_t n_inputs;
  int32_t umi_read, cell_read, sample_read;
  int32_t phred, min_qual;
  int32_t out_sam, tenx;
  int32_t emit[3];
  int64_t umi_off, umi_size, cell_off, cell_size, sample_off, sample_size;
  int64_t read_off[3], read_size[3];
  uint64_t first_read_number;  // processed_reads before this batch
};

struct BcCall {
  unsigned long long first_finding;  // min over iterations of (iteration << 8 | code << 3 | file)
  unsigned long long first_discard;  // first discarded iteration (interleaved input re-syncs there)
  unsigned long long discarded, short_warnings;
  unsigned long long big;  // kBcKeepBig iterations
};

struct BcLine {
  const uint8_t* p;
  uint32_t len;  // bytes before '\n'
  uint32_t nl;
};

__device__ __forceinline__ void bc_lines(const BcFile& f, uint64_t k, BcLine ln[4]) {
  const uint64_t r = f.first + k * f.step + f.add;
  uint64_t prev = r == 0 ? ~0ull : f.fv.line_end[4 * r - 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t e = f.fv.line_end[4 * r + i];
    ln[i].p = f.fv.img + prev + 1;
    ln[i].len = (uint32_t)(e - prev - 1);
    ln[i].nl = e < f.fv.nbytes ? 1u : 0u;
    prev = e;
  }
}

// canonical read name with is_pe set (every input of fastq_pre_barcodes has it, :570)
__device__ inline uint32_t bc_name_len(const BcLine& h, int fmt) {
  const uint32_t L = h.len + h.nl - 1;  // strlen(&hdr[1]); images with NUL bytes are refused earlier
  if (fmt == FQG_NAME_CASAVA18) {
    uint32_t sp = L;
    for (uint32_t i = 0; i < L; ++i)
      if (h.p[1 + i] == ' ') {
        sp = i;
        break;
      }
    if (sp >= 2 && h.p[1 + sp - 2] == '/') sp -= 2;
    return sp;
  }
  long l = (long)L;
  if (fmt == FQG_NAME_DEFAULT) l--;
  return l >= 1 ? (uint32_t)(l - 1) : L;
}

// slice_read's effect on one line (src/fastq_pre_barcodes.c:168-189): the result is
// s[from .. from+n) followed by '\n' when add_nl.  s = the line as a C string of length L.
struct Cut {
  uint32_t from, n, add_nl;
};
__device__ __forceinline__ Cut bc_cut(uint32_t L, long off, long size) {
  Cut c{0, L, 0};
  if (size == 0) return Cut{0, 0, 1};
  if (off > 0 && size == -1) return Cut{0, 0, 0};  // seq[-1]='\n'; seq[0]='\0'
  uint32_t from = 0, Lt = L;
  if (off > 0) {
    from = (uint32_t)((unsigned long)off < L ? off : L);
    Lt = L - from;
  }
  if (size < 0) return Cut{from, Lt, 0};
  if ((unsigned long)size < Lt) return Cut{from, (uint32_t)size, 1};
  if ((unsigned long)size == Lt) return Cut{from, Lt, 1};
  return Cut{from, Lt, 0};
}

__device__ __forceinline__ bool bc_slices(const BcParams& P, int x) {
  if (P.read_off[x] == -1) return false;
  if (P.read_off[x] == 0 && P.read_size[x] == -1) return false;
  return true;
}

__device__ __forceinline__ uint32_t dec_digits(unsigned long v) {
  uint32_t d = 1;
  while (v >= 10) {
    v /= 10;
    ++d;
  }
  return d;
}

struct BcTags {
  uint32_t n[3];  // umi, cell, sample lengths (0: absent)
  const uint8_t* s[3];
  const uint8_t* q[3];
};

// get_barcode for one tag (src/fastq_pre_barcodes.c:218-259).  0 ok, 1 short, 2 low quality
__device__ inline int bc_get(const BcLine ln[4], long off, long size, int phred, int min_qual, uint32_t* n,
                             const uint8_t** s, const uint8_t** q) {
  *n = 0;
  if (off == -1 || size == 0) return 0;
  const unsigned long rl1 = (unsigned long)(ln[1].len + ln[1].nl) - 1ul;
  if ((unsigned long)off > rl1 || (unsigned long)(off + size) > rl1) return 1;
  if (min_qual > 0)
    for (long x = off; x < off + size; ++x) {
      const int c = (int)(signed char)(x < (long)(ln[3].len + ln[3].nl) ? ln[3].p[x] : 0);
      if (c - phred < min_qual) return 2;
    }
  *n = (uint32_t)size;
  *s = ln[1].p + off;
  *q = ln[3].p + off;
  return 0;
}

// status + tags of one iteration; shared by plan and emit so that both see the same thing
__device__ inline uint8_t bc_decide(const BcParams& P, uint64_t k, BcTags* tags, uint32_t* finding) {
  *finding = 0;
  tags->n[0] = tags->n[1] = tags->n[2] = 0;
  BcLine ln[4], l1[4];
  if (P.n_inputs > 1) {
    // names: every file against READ1 (src/fastq_pre_barcodes.c:606-635); '@' first (src/fastq.c:448)
    for (int x = 1; x < kBcFiles; ++x)
      if (P.f[x].present) {
        bc_lines(P.f[x], k, ln);
        if (ln[0].p[0] != '@') {
          *finding = (FQG_E_WRONG_HEADER << 3) | x;
          return kBcFinding;
        }
      }
    bc_lines(P.f[1], k, l1);
    const uint32_t n1 = bc_name_len(l1[0], P.f[1].fmt);
    for (int x = 2; x < kBcFiles; ++x)
      if (P.f[x].present) {
        bc_lines(P.f[x], k, ln);
        const uint32_t nx = bc_name_len(ln[0], P.f[x].fmt);
        bool same = nx == n1;
        for (uint32_t i = 0; same && i < n1; ++i) same = l1[0].p[1 + i] == ln[0].p[1 + i];
        if (!same) {
          *finding = (FQG_E_NAME_MISMATCH << 3) | x;
          return kBcFinding;
        }
      }
  }
  // extract_info for each file in order: umi, sample, cell (src/fastq_pre_barcodes.c:262-285)
  for (int x = 1; x < kBcFiles; ++x)
    if (P.f[x].present) {
      bc_lines(P.f[x], k, ln);
      int rc = 0;
      if (P.umi_read == x) rc = bc_get(ln, P.umi_off, P.umi_size, P.phred, P.min_qual, &tags->n[0], &tags->s[0], &tags->q[0]);
      if (!rc && P.sample_read == x)
        rc = bc_get(ln, P.sample_off, P.sample_size, P.phred, P.min_qual, &tags->n[2], &tags->s[2], &tags->q[2]);
      if (!rc && P.cell_read == x)
        rc = bc_get(ln, P.cell_off, P.cell_size, P.phred, P.min_qual, &tags->n[1], &tags->s[1], &tags->q[1]);
      if (rc) return rc == 1 ? kBcDiscardShort : kBcDiscardQual;
    }
  return kBcKeep;
}

// tags of an iteration that bc_decide has already found to be kept: geometry only, none of the
// byte-by-byte checks (the emit kernel runs one wavefront per iteration: a serial scan there costs
// a memory round trip per byte)
__device__ __forceinline__ void bc_tags_of_kept(const BcParams& P, const BcLine (&lines)[kBcFiles][4], BcTags* tags) {
  tags->n[0] = tags->n[1] = tags->n[2] = 0;
#pragma unroll
  for (int x = 1; x < kBcFiles; ++x)
    if (P.f[x].present && (P.umi_read == x || P.sample_read == x || P.cell_read == x)) {
      const BcLine(&ln)[4] = lines[x];
      auto take = [&](int slot, long off, long size) {
        if (off == -1 || size == 0) return;
        tags->n[slot] = (uint32_t)size;
        tags->s[slot] = ln[1].p + off;
        tags->q[slot] = ln[3].p + off;
      };
      if (P.umi_read == x) take(0, P.umi_off, P.umi_size);
      if (P.sample_read == x) take(2, P.sample_off, P.sample_size);
      if (P.cell_read == x) take(1, P.cell_off, P.cell_size);
    }
}

// byte counts of the FASTQ record written for file x (src/fastq_pre_barcodes.c:713-718)
__device__ inline uint32_t bc_fastq_len(const BcParams& P, int x, uint64_t k, const BcTags& t) {
  BcLine ln[4];
  bc_lines(P.f[x], k, ln);
  const bool tagged = (t.n[0] | t.n[1] | t.n[2]) != 0;
  const bool sliced = bc_slices(P, x);
  uint32_t n = ln[0].len + ln[0].nl + (tagged ? 31u + t.n[0] + t.n[1] + t.n[2] : 0u);
  n += (tagged || sliced) ? 2u : ln[2].len + ln[2].nl;
  if (sliced) {
    const Cut cs = bc_cut(ln[1].len + ln[1].nl, P.read_off[x], P.read_size[x]);
    const Cut cq = bc_cut(ln[3].len + ln[3].nl, P.read_off[x], P.read_size[x]);
    n += cs.n + cs.add_nl + cq.n + cq.add_nl;
  } else {
    n += ln[1].len + ln[1].nl + ln[3].len + ln[3].nl;
  }
  return n;
}

// what the SAM line prints for mate x: sequence / quality without their last character
// (src/fastq_pre_barcodes.c:666-700)
struct SamGeom {
  Cut cs, cq;
  uint32_t seq_n, qual_n;  // characters printed
  uint32_t shown;          // the number in column 9
  uint32_t name_n;         // characters of the on:Z: value
};
__device__ inline SamGeom bc_sam_geom(const BcParams& P, int x, const BcLine ln[4]) {
  SamGeom g;
  const uint32_t Ls = ln[1].len + ln[1].nl, Lq = ln[3].len + ln[3].nl;
  if (bc_slices(P, x)) {
    g.cs = bc_cut(Ls, P.read_off[x], P.read_size[x]);
    g.cq = bc_cut(Lq, P.read_off[x], P.read_size[x]);
  } else {
    g.cs = Cut{0, Ls, 0};
    g.cq = Cut{0, Lq, 0};
  }
  const uint32_t ls = g.cs.n + g.cs.add_nl, lq = g.cq.n + g.cq.add_nl;  // strlen after slicing
  g.seq_n = ls ? ls - 1 : 0;
  g.qual_n = lq ? lq - 1 : 0;
  g.shown = x == 1 ? ls - 1u : ls;  // unsigned arithmetic as in the reference (len-1 for mate 1, len for mate 2)
  // format_read_name: up to the first '\n' of the header, without the '@'
  g.name_n = ln[0].len ? ln[0].len - 1 : 0;
  return g;
}

__device__ inline uint32_t bc_sam_len(const BcParams& P, uint64_t k, const BcTags& t) {
  uint32_t total = 0;
  const bool se = !P.f[2].present;
  for (int x = 1; x <= (se ? 1 : 2); ++x) {
    BcLine ln[4];
    bc_lines(P.f[x], k, ln);
    const SamGeom g = bc_sam_geom(P, x, ln);
    const unsigned flag = se ? 4u : (x == 1 ? 77u : 141u);
    uint32_t n = dec_digits(P.first_read_number + k + 1) + 1 + dec_digits(flag);
    n += 15;  // "\t*\t0\t255\t*\t*\t0\t"
    n += dec_digits(g.shown) + 1 + g.seq_n + 1 + g.qual_n + 6 + g.name_n + 6 + g.qual_n;
    if (t.n[0]) n += 12 + 2 * t.n[0];
    if (t.n[1]) n += 12 + 2 * t.n[1];
    if (t.n[2]) n += 12 + 2 * t.n[2];
    total += n + 1;
  }
  return total;
}

__global__ __launch_bounds__(kBlock) void k_bc_plan(BcParams P, uint64_t n_iter, uint8_t* __restrict__ status,
                                                    uint32_t* __restrict__ len0, uint32_t* __restrict__ len1,
                                                    uint32_t* __restrict__ len2, BcCall* __restrict__ call) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n_iter) return;
  BcTags t;
  uint32_t finding;
  uint8_t st = bc_decide(P, k, &t, &finding);
  uint32_t a = 0, b = 0, c = 0;
  if (st == kBcKeep) {
    if (P.out_sam) a = bc_sam_len(P, k, t);
    else {
      if (P.emit[1]) b = bc_fastq_len(P, 1, k, t);
      if (P.emit[2]) c = bc_fastq_len(P, 2, k, t);
    }
    // do the records and the text fit the LDS buffers of the emit kernel?
    uint64_t in_bytes = 0;
    for (int x = 1; x < kBcFiles; ++x)
      if (P.f[x].present) {
        BcLine ln[4];
        bc_lines(P.f[x], k, ln);
        in_bytes += (uint64_t)((ln[3].p + ln[3].len + ln[3].nl) - ln[0].p);
      }
    if (in_bytes > (uint64_t)kEmitSrc || a > (uint32_t)kEmitBuf || b > (uint32_t)kEmitBuf || c > (uint32_t)kEmitBuf) {
      st = kBcKeepBig;
      atomicAdd(&call->big, 1ull);
    }
  } else if (st == kBcFinding) {
    atomicMin(&call->first_finding, (unsigned long long)((k << 8) | finding));
  } else {
    atomicMin(&call->first_discard, (unsigned long long)k);
  }
  status[k] = st;
  len0[k] = a;
  len1[k] = b;
  len2[k] = c;
}

// discards / warnings among the first n_done iterations (counted after the host has decided
// where the batch ends)
__global__ __launch_bounds__(kBlock) void k_bc_count(const uint8_t* __restrict__ status, uint64_t n_done,
                                                     BcCall* __restrict__ call) {
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint8_t st = k < n_done ? status[k] : kBcKeep;
  const uint64_t d = __ballot(st == kBcDiscardShort || st == kBcDiscardQual), s = __ballot(st == kBcDiscardShort);
  if ((threadIdx.x & 63) == 0 && d) {
    atomicAdd(&call->discarded, (unsigned long long)__popcll(d));
    if (s) atomicAdd(&call->short_warnings, (unsigned long long)__popcll(s));
  }
}

// ---- 64-bit exclusive scan of u32 lengths: 2048 per workgroup ---------------------------------
constexpr int kScan64Span = kBlock * 8;
__global__ __launch_bounds__(kBlock) void k_scan64_a(const uint32_t* __restrict__ in, uint64_t n,
                                                     unsigned long long* __restrict__ local,
                                                     unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long s_w[kBlock / kWave];
  const uint64_t first = (uint64_t)blockIdx.x * kScan64Span + threadIdx.x * 8;
  unsigned long long v[8], sum = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = first + i < n ? in[first + i] : 0u;
    sum += v[i];
  }
  unsigned long long incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(incl, d, 64);
    if ((int)(threadIdx.x & 63) >= d) incl += o;
  }
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned long long before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) {
    if (w < (int)(threadIdx.x >> 6)) before += s_w[w];
    all += s_w[w];
  }
  unsigned long long run = before + incl - sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (first + i < n) local[first + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = all;
}
// one workgroup: exclusive prefix over the span sums, total into *total
__global__ __launch_bounds__(kBlock) void k_scan64_b(unsigned long long* __restrict__ sums, uint64_t nb,
                                                     unsigned long long* __restrict__ total) {
  __shared__ unsigned long long s_part[kBlock];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint64_t base = 0; base < nb; base += kBlock) {
    const uint64_t i = base + threadIdx.x;
    const unsigned long long v = i < nb ? sums[i] : 0ull;
    s_part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < kBlock; d <<= 1) {
      const unsigned long long o = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0ull;
      __syncthreads();
      s_part[threadIdx.x] += o;
      __syncthreads();
    }
    const unsigned long long carry = s_carry;
    if (i < nb) sums[i] = carry + s_part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) s_carry = carry + s_part[kBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}

// ---- emit -----------------------------------------------------------------------------------
// up to 16 characters of a literal as two 64-bit immediates (no memory access when written)
constexpr uint64_t lit_pack(const char* s, int n, int from) {
  uint64_t v = 0;
  for (int i = 0; i < 8 && from + i < n; ++i) v |= (uint64_t)(uint8_t)s[from + i] << (8 * i);
  return v;
}
#define BC_LIT(w, str) (w).lit_imm(lit_pack(str, (int)sizeof(str) - 1, 0), lit_pack(str, (int)sizeof(str) - 1, 8), (uint32_t)sizeof(str) - 1)

struct Writer {
  uint8_t* p;
  int lane;
  __device__ __forceinline__ void lit_imm(uint64_t lo, uint64_t hi, uint32_t n) {
    if ((uint32_t)lane < n) p[lane] = (uint8_t)((lane < 8 ? lo >> (8 * lane) : hi >> (8 * (lane - 8))) & 0xFFu);
    p += n;
  }
  __device__ __forceinline__ void bytes(const uint8_t* s, uint32_t n) {
    for (uint32_t i = lane; i < n; i += kWave) p[i] = s[i];
    p += n;
  }
  __device__ __forceinline__ void lit(const char* s, uint32_t n) { bytes(reinterpret_cast<const uint8_t*>(s), n); }
  __device__ __forceinline__ void ch(char c) {
    if (lane == 0) p[0] = (uint8_t)c;
    p += 1;
  }
  __device__ __forceinline__ void dec(unsigned long v) {
    const uint32_t d = dec_digits(v);
    if (lane == 0) {
      for (uint32_t i = 0; i < d; ++i) {
        p[d - 1 - i] = (uint8_t)('0' + v % 10);
        v /= 10;
      }
    }
    p += d;
  }
  // header text for on:Z: - blanks become '@' (format_read_name, src/fastq_pre_barcodes.c:300-308)
  __device__ __forceinline__ void name(const uint8_t* s, uint32_t n) {
    for (uint32_t i = lane; i < n; i += kWave) p[i] = s[i] == ' ' ? (uint8_t)'@' : s[i];
    p += n;
  }
  __device__ __forceinline__ void cut(const uint8_t* s, const Cut& c) {
    bytes(s + c.from, c.n);
    if (c.add_nl) ch('\n');
  }
};

__device__ __forceinline__ void bc_emit_fastq(const BcParams& P, int x, const BcLine (&ln)[4], const BcTags& t, Writer& w) {
  const bool tagged = (t.n[0] | t.n[1] | t.n[2]) != 0;
  const bool sliced = bc_slices(P, x);
  if (tagged) {  // add_tags2readname, src/fastq_pre_barcodes.c:192-216
    w.ch((char)ln[0].p[0]);
    BC_LIT(w, "STAGS_CELL=");
    w.bytes(t.s[1], t.n[1]);
    BC_LIT(w, "_UMI=");
    w.bytes(t.s[0], t.n[0]);
    BC_LIT(w, "_SAMPLE=");
    w.bytes(t.s[2], t.n[2]);
    BC_LIT(w, "_ETAGS_");
    w.bytes(ln[0].p + 1, ln[0].len + ln[0].nl - 1);
  } else {
    w.bytes(ln[0].p, ln[0].len + ln[0].nl);
  }
  if (sliced) w.cut(ln[1].p, bc_cut(ln[1].len + ln[1].nl, P.read_off[x], P.read_size[x]));
  else w.bytes(ln[1].p, ln[1].len + ln[1].nl);
  if (tagged || sliced) {
    w.ch((char)ln[2].p[0]);
    w.ch('\n');
  } else {
    w.bytes(ln[2].p, ln[2].len + ln[2].nl);
  }
  if (sliced) w.cut(ln[3].p, bc_cut(ln[3].len + ln[3].nl, P.read_off[x], P.read_size[x]));
  else w.bytes(ln[3].p, ln[3].len + ln[3].nl);
}

__device__ __forceinline__ void bc_emit_sam(const BcParams& P, uint64_t k, const BcLine (&lines)[kBcFiles][4], const BcTags& t,
                                   Writer& w) {
  const bool se = !P.f[2].present;
#pragma unroll
  for (int x = 1; x <= 2; ++x) {
    if (x == 2 && se) break;
    const BcLine(&ln)[4] = lines[x];
    const SamGeom g = bc_sam_geom(P, x, ln);
    const unsigned flag = se ? 4u : (x == 1 ? 77u : 141u);  // BAM_FUNMAP | FMUNMAP | FPAIRED | FREAD1/2
    w.dec(P.first_read_number + k + 1);
    w.ch('\t');
    w.dec(flag);
    BC_LIT(w, "\t*\t0\t255\t*\t*\t0\t");
    w.dec(g.shown);
    w.ch('\t');
    w.bytes(ln[1].p + g.cs.from, g.seq_n);
    w.ch('\t');
    w.bytes(ln[3].p + g.cq.from, g.qual_n);
    BC_LIT(w, "\ton:Z:");
    w.name(ln[0].p + 1, g.name_n);
    BC_LIT(w, "\top:Z:");
    w.bytes(ln[3].p + g.cq.from, g.qual_n);
    if (t.n[0]) {
      if (P.tenx) BC_LIT(w, "\tUB:Z:"); else BC_LIT(w, "\tRX:Z:");
      w.bytes(t.s[0], t.n[0]);
      if (P.tenx) BC_LIT(w, "\tUY:Z:"); else BC_LIT(w, "\tQX:Z:");
      w.bytes(t.q[0], t.n[0]);
    }
    if (t.n[1]) {
      if (x == 1) BC_LIT(w, "\tCR:Z:"); else BC_LIT(w, " CR:Z:");  // the second mate gets a blank (src/fastq_pre_barcodes.c:705)
      w.bytes(t.s[1], t.n[1]);
      BC_LIT(w, "\tCY:Z:");
      w.bytes(t.q[1], t.n[1]);
    }
    if (t.n[2]) {
      BC_LIT(w, "\tBC:Z:");
      w.bytes(t.s[2], t.n[2]);
      BC_LIT(w, "\tQT:Z:");
      w.bytes(t.q[2], t.n[2]);
    }
    w.ch('\n');
  }
}

// k_bc_emit<SAM>: one wavefront per kept iteration whose records and text fit the LDS buffers (all
// but long reads; the others are kBcKeepBig and go through k_bc_emit_direct).
//   1. The records of the iteration (the four lines are contiguous in the image) are copied into an
//      LDS staging area with all loads in flight at once - the ~25 pieces of a SAM line would otherwise
//      cost one memory round trip each.
//   2. The output text is assembled piece by piece in a second LDS buffer (LDS -> LDS byte copies,
//      literals as immediates).
//   3. The text goes to its place in the output image with 16-byte stores: the buffer starts at the
//      same residue mod 16 as the global address, so aligned 16-byte units of LDS and of the output
//      coincide.
constexpr int kEmitChunks = 10;  // kEmitSrc / 64 / 4: byte loads in flight per lane and round

__device__ __forceinline__ void emit_flush(const uint8_t* __restrict__ buf, uint32_t skew, uint32_t len,
                                           uint8_t* __restrict__ dst, int lane) {
  // buf[skew .. skew+len) -> dst[0 .. len), with (dst - skew) 16-byte aligned
  uint8_t* g0 = dst - skew;
  const uint32_t end = skew + len;
  const uint32_t first_full = (skew + 15u) & ~15u, last_full = end & ~15u;
  for (uint32_t i = skew + lane; i < (first_full < end ? first_full : end); i += kWave) g0[i] = buf[i];
  for (uint32_t u = first_full + 16u * lane; u + 16u <= last_full; u += 16u * kWave)
    *reinterpret_cast<uint4*>(g0 + u) = *reinterpret_cast<const uint4*>(buf + u);
  if (last_full >= first_full)
    for (uint32_t i = last_full + lane; i < end; i += kWave) g0[i] = buf[i];
}

struct EmitOut {
  const uint32_t* len;
  const unsigned long long* off;
  const unsigned long long* sum;
  uint8_t* out;
};

template <bool SAM>
__global__ __launch_bounds__(kBlock) void k_bc_emit(BcParams P, uint64_t n_done, const uint8_t* __restrict__ status,
                                                    EmitOut o0, EmitOut o1, EmitOut o2) {
  __shared__ __attribute__((aligned(16))) uint8_t s_buf[kBlock / kWave][kEmitBuf + 16];
  __shared__ uint8_t s_src[kBlock / kWave][kEmitSrc];
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  // the wave index is uniform: say so, and everything derived from k (line index loads, pointers,
  // lengths) lives in scalar registers instead of 64 copies
  const int lane = (int)(threadIdx.x & 63), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* buf = s_buf[wv];
  uint8_t* src = s_src[wv];
  for (uint64_t k = (uint64_t)blockIdx.x * (kBlock / kWave) + wv; k < n_done; k += n_waves) {
    if (status[k] != kBcKeep) continue;
    BcLine lines[kBcFiles][4];
    uint32_t at = 0;
#pragma unroll
    for (int x = 1; x < kBcFiles; ++x) {
      if (!P.f[x].present) continue;
      if (SAM && x > 2 && P.umi_read != x && P.cell_read != x && P.sample_read != x) continue;  // not printed, no tag
      bc_lines(P.f[x], k, lines[x]);
      const uint8_t* g = lines[x][0].p;
      const uint32_t n = (uint32_t)((lines[x][3].p + lines[x][3].len + lines[x][3].nl) - g);
      // stage: every lane issues its byte loads before the first one is consumed
      for (uint32_t base = 0; base < n; base += kEmitChunks * kWave) {
        uint8_t v[kEmitChunks];
#pragma unroll
        for (int c = 0; c < kEmitChunks; ++c) {
          const uint32_t i = base + c * kWave + lane;
          v[c] = i < n ? g[i] : (uint8_t)0;
        }
#pragma unroll
        for (int c = 0; c < kEmitChunks; ++c) {
          const uint32_t i = base + c * kWave + lane;
          if (i < n) src[at + i] = v[c];
        }
      }
#pragma unroll
      for (int l = 0; l < 4; ++l) lines[x][l].p = src + at + (uint32_t)(lines[x][l].p - g);  // now in LDS
      at += n;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    BcTags t;
    bc_tags_of_kept(P, lines, &t);
#pragma unroll
    for (int which = SAM ? 0 : 1; which < (SAM ? 1 : 3); ++which) {
      if (!SAM && !P.emit[which]) continue;
      const EmitOut& o = which == 0 ? o0 : (which == 1 ? o1 : o2);
      uint8_t* dst = o.out + o.off[k] + o.sum[k / kScan64Span];
      const uint32_t len = o.len[k];
      const uint32_t skew = (uint32_t)((uintptr_t)dst & 15u);
      Writer w{buf + skew, lane};
      if (SAM) bc_emit_sam(P, k, lines, t, w);
      else bc_emit_fastq(P, which, lines[which], t, w);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      emit_flush(buf, skew, len, dst, lane);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// kBcKeepBig iterations (long reads): the text is written straight to the output image
__global__ __launch_bounds__(kBlock) void k_bc_emit_direct(BcParams P, uint64_t n_done,
                                                           const uint8_t* __restrict__ status, EmitOut o0, EmitOut o1,
                                                           EmitOut o2) {
  const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const int lane = (int)(threadIdx.x & 63), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (uint64_t k = (uint64_t)blockIdx.x * (kBlock / kWave) + wv; k < n_done; k += n_waves) {
    if (status[k] != kBcKeepBig) continue;
    BcLine lines[kBcFiles][4];
#pragma unroll
    for (int x = 1; x < kBcFiles; ++x)
      if (P.f[x].present) bc_lines(P.f[x], k, lines[x]);
    BcTags t;
    bc_tags_of_kept(P, lines, &t);
    for (int which = 0; which < 3; ++which) {
      if (which == 0 ? !P.out_sam : (P.out_sam || !P.emit[which])) continue;
      const EmitOut& o = which == 0 ? o0 : (which == 1 ? o1 : o2);
      Writer w{o.out + o.off[k] + o.sum[k / kScan64Span], lane};
      if (which == 0) bc_emit_sam(P, k, lines, t, w);
      else if (which == 1) bc_emit_fastq(P, 1, lines[1], t, w);
      else bc_emit_fastq(P, 2, lines[2], t, w);
    }
  }
}

}  // namespace fqg
