// fqg_device.h - device-side data structures shared by the kernels and the C-ABI host code.
// gfx950 only (64-lane wavefronts are assumed throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fqg.h"

namespace fqg {

constexpr int kWave = 64;
constexpr int kBlock = 256;                 // 4 waves
constexpr int kChunkBytes = 4096;           // bytes framed by one wavefront per step
constexpr int kSlices = kChunkBytes / (kWave * 16);  // 1 KiB slices (64 lanes x 16 B) per chunk
constexpr int kScanSpan = 4096;             // chunk counts scanned by one workgroup
constexpr uint64_t kNoRecord = ~0ull;

// image-level flags raised by the framing pass
constexpr uint32_t kFlagNul = 1u;       // a NUL byte somewhere in the image
constexpr uint32_t kFlagCr = 2u;        // a '\r' somewhere in the image
constexpr uint32_t kFlagHigh = 8u;      // a byte >= 0x80 (streaming pass: its SWAR tests need 7-bit bytes)
constexpr uint32_t kFlagSuspectOverflow = 16u;  // a suspect record lies beyond the bitmap
constexpr uint32_t kFlagStageOverflow = 32u;    // a chunk with more newlines than the staging area holds
constexpr uint32_t kFlagQueueOverflow = 64u;    // more suspect positions than the queue holds

// streaming (single-pass) framing: per 4 KiB chunk, newline offsets are staged as 16-bit entries
constexpr int kStageCap = 256;  // entries per chunk; denser images take the two-pass path
// staged entry: bits 0..11 offset of the '\n' inside the chunk; 12..13 class of the byte after it
// (1 = '@', 2 = '+', 0 = other); bit 14: the byte after that one is a '\n'
constexpr uint32_t kClsAt = 1u, kClsPlus = 2u;
// chunk info word: bits 0..1 speculated line type of the chunk's first byte; bit 2: no speculation
// (the chunk must be re-checked once its true rank is known); bit 3: qmin/qmax valid (bits 8..15, 16..23);
// bit 4: the chunk holds no newline - it lies inside ONE line, whose type the rank will tell - and was checked for
// both kinds of line: bit 5 = it holds a byte that is no base (matters if the line is a sequence), qmin/qmax = the
// range of ALL its bytes when one is outside the boot range (matters if the line is a quality line)
constexpr uint32_t kInfoUnknown = 4u, kInfoRange = 8u, kInfoOneLine = 16u, kInfoNotBases = 32u;
// does the chunk with this info word and this true rank of its first byte need its checks repeated?  (the range of a
// quality line's bytes is merged by the caller when this says no)
__host__ __device__ inline bool chunk_info_redo(uint32_t info, uint32_t rank0) {
  if (info & kInfoOneLine) return (rank0 & 3u) == 1u && (info & kInfoNotBases);
  return (info & kInfoUnknown) || (info & 3u) != (rank0 & 3u);
}
__host__ __device__ inline bool chunk_info_range(uint32_t info, uint32_t rank0) {
  if (!(info & kInfoRange)) return false;
  return (info & kInfoOneLine) ? (rank0 & 3u) == 3u : true;
}

// ---- header lines captured by the streaming pass (FQG_VALIDATE_NAMES) -----------------------------
// While a chunk's bytes are in LDS, pass 1 copies the start of every header line that BEGINS in the chunk
// (under its speculated line type) into a 64-byte record, so that the name kernels never go back to the image:
//     bytes 0..3   meta: bits 0..9  bytes of the line in front of its '\n', the '@' included (1023 = that or more)
//                        bits 10..18 v: the line starts behind the chunk's v-th newline (v = 0: at the chunk's
//                                    first byte) - with the chunk's first rank this gives the record index
//                        bit 19      the length is known (the line's '\n' was seen)
//                        bit 20      the line starts with '@'
//                        bit 21      the length is not known, but every text byte of the record is the line's (a
//                                    header that runs into the next chunk and has no '\n' in its first 61 bytes)
//     bytes 4..63  the 60 bytes behind the '@' (whatever follows the line when it is shorter)
// Records sit at [chunk * K + ordinal of the header in the chunk]; hcount[chunk] = headers the chunk saw
// (kNoCapture: none captured - no speculation in this chunk).  A consumer trusts a chunk's records only when the
// speculated type was the true one and hcount <= K; every other header is found through the line index.
constexpr uint32_t kNameRecWords = 8;       // 64-bit words per record
constexpr uint32_t kNameRecText = 60;       // header bytes behind the '@' in a record
constexpr uint32_t kNameInline = 56;        // name bytes the index keeps per record (NameRec)
constexpr uint32_t kNoCapture = 0xFFFFu;
struct NameCapture {
  unsigned long long* recs;  // n_chunks * K records (or digests)
  uint16_t* hcount;          // per chunk
  uint32_t K;                // record slots per chunk
  int fmt, is_pe;            // digests: the read-name format / is_pe the names are canonicalised under (fqg_file_state)
};
// ---- name DIGESTS instead of records (FQG_VALIDATE_NAME_DIGESTS) ------------------------------------
// For an index that only tests names for uniqueness (fastq_info on ONE file: nobody will look a name up, so nobody needs
// its bytes unless two tags agree) the streaming pass canonicalises and hashes every header line itself while the chunk
// is in LDS - four lanes per header, 16 name bytes each - and stores 16 bytes per header instead of 64:
//     word 0   the hash of the canonical name (name_fin below: the value every other path computes for it)
//     word 1   bits 0..9 n, the canonical name's length; 10..19 the `len` the reference accounts for it
//              (src/fastq.c:609); 20..28 v (as in a record); bit 29 the line starts with '@'; bit 30 the digest is
//              good - clear: the line's end was not seen, the name is longer than kDigestText bytes, a Casava header has
//              no blank ... the name kernels then read that header through the line index, as they do for records
// Same places as records: [chunk * K + ordinal], hcount[chunk], trusted under the same conditions.
constexpr uint32_t kDigestWords = 2;
constexpr uint32_t kDigestText = 64;        // name bytes the four lanes of a header look at
constexpr uint32_t kDigestAt = 1u << 29, kDigestOk = 1u << 30;

// The hash of a name (round 6): H = fin(seed(n) + sum over the name's 8-byte words (lo_k, hi_k) of
// (lo_k + A_k) * (hi_k + B_k)), zero bytes behind the name, the words that hold at least one name byte (8 k < n) - the NH
// construction: ONE 32 x 32 -> 64 multiply-add per eight bytes.  A SUM, so that the words can be taken in any order and
// by different lanes: the streaming pass hashes a header with four lanes of 16 bytes each while the chunk is in LDS
// (name digests, fqg_stream_kernels.hip).  What a multiply costs decides the form: v_mul_lo_u32 / v_mad_u64_u32 issue
// at a quarter of the rate and a 64 x 64 product is three of them (rounds 1 - 5 chained one per word, eight dependent
// on one lane; the first form of this round - w_k * M_k with 64-bit M_k - was nine quarter-rate operations per lane of
// pass 1, 0.6 ms of its 10).  Equality is decided on the name BYTES wherever this value is used: it only chooses a
// slot and a tag.
__host__ __device__ constexpr uint64_t name_mul(uint32_t k) {
  return (0x9E3779B97F4A7C15ull + (uint64_t)k * 0xD1B54A32D192ED03ull) | 1ull;
}
__host__ __device__ constexpr uint32_t name_key_a(uint32_t k) { return (uint32_t)name_mul(k); }
__host__ __device__ constexpr uint32_t name_key_b(uint32_t k) { return (uint32_t)(name_mul(k) >> 32); }
// word k of a name (its bytes 8 k .. 8 k + 7, zero behind the name) into the sum
__device__ __forceinline__ uint64_t name_word(uint64_t sum, uint64_t w, uint32_t k) {
  return sum + (uint64_t)((uint32_t)w + name_key_a(k)) * (uint64_t)((uint32_t)(w >> 32) + name_key_b(k));
}
__device__ __forceinline__ uint64_t name_seed(uint32_t n) { return 0x2545F4914F6CDD1Dull + (uint64_t)n; }
__device__ __forceinline__ uint64_t name_fin(uint64_t h) {
  const uint32_t hi = (uint32_t)(h >> 32), t = (uint32_t)h ^ hi;  // (the low bits of a sum of products are its weak ones)
  const uint64_t p = (uint64_t)t * 0xD6E8FEB9u, q = (uint64_t)hi * 0x85EBCA6Bu;
  return p ^ ((q << 32) | (q >> 32));
}

// Scalars of one fqg_validate() call.  Lives in device memory; the host copies it back once.
struct CallState {
  unsigned long long first_key;   // min over failing records of (record << 8 | code)
  unsigned long long stop_record; // min record whose first line starts with NUL
  unsigned long long n_newlines;  // total '\n' in the image
  unsigned long long aux0, aux1;  // filled by the explain launch for first_key's record
  unsigned long long list_count;  // records queued by the fast path for the exact validator
  unsigned int flags;
  unsigned int last_byte_is_nl;
  unsigned int qmin_byte, qmax_byte;  // quality range seen by the tiled pass (255 / 0 when none)
  unsigned long long queue_count;     // streaming pass: suspect byte positions queued
  unsigned int redo_count;            // streaming pass: chunks whose checks must be repeated
  unsigned int boot_qmin, boot_qmax;  // streaming pass: quality range of the image's first records
  unsigned int boot_lines;            // streaming pass: newlines in the boot window (mean record size -> NameCapture::K)
  unsigned int pad_;
  unsigned long long trunc_record;    // frame-only images: min record whose 2nd, 3rd or 4th line starts with NUL
  // the streaming pass in parts (k_stream_pass1_lines): newlines up to and including part p; the steps the line workers
  // inside the pass-1 launches have done
  unsigned long long part_newlines[4];
  unsigned long long lines_done_steps;
};

// Device counterpart of FASTQ_FILE's counters (src/fastq.h:116-122).
struct AccState {
  unsigned long long num_rds;
  unsigned long long min_rl;   // init FQG_MAX_READ_LENGTH
  unsigned long long max_rl;   // init 0
  unsigned int min_qbyte;      // init 255 (unsigned-byte domain; mapped at read-out)
  unsigned int max_qbyte;      // init 0
};

// 1 bit per record: the tiled pass could not vouch for it.
struct SuspectMap {
  uint32_t* bits;
  unsigned long long cap;  // records covered
  unsigned int* flags;     // &CallState::flags
};

struct FrameView {
  const uint8_t* img;
  uint64_t nbytes;
  const uint64_t* line_end;  // per line: offset of its '\n', or nbytes for an unterminated last line
  uint64_t n_lines;
  uint64_t n_records;
  uint32_t reframed = 0;  // FQG_VALIDATE_REFRAMED: lines are within the gzgets limits by construction
};

}  // namespace fqg
