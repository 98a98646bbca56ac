/*
 * fqg.h - bulk C-ABI of libfqgpu.so: the MI355X (gfx950) implementation of fastq_utils'
 * per-read hot path.
 *
 * The reference (nunofonseca/fastq_utils 0.25.3) has no FFI layer: its programs call a
 * per-record C API (src/fastq.h:133-158, src/hash.h:64-78) in a serial loop.  A per-record
 * call cannot feed a GPU, so the boundary is one level up: each entry point below replaces one
 * whole *loop* of the reference and works on a block of records ("image": the decompressed
 * bytes of a FASTQ file or of a record-aligned piece of one).  The reference interface every
 * entry point stands in for is cited next to it.  INTEGRATION.md shows how the reference's own
 * programs would bind these.
 *
 * Conventions: plain C, no torch / HIP types in any signature; every function returns 0 on
 * success or a negative FQG_ERR_* value and never calls exit(); record-level findings (format
 * errors in the data) are *results*, not failures.  Buffers tagged FQG_MEM_DEVICE are device
 * pointers owned by the caller (e.g. a torch tensor's data_ptr()); FQG_MEM_HOST buffers are
 * copied to the GPU by the library (pinned memory from fqg_host_alloc() copies fastest).
 * One fqg_ctx per process and GPU; a context is not thread-safe.
 */
#ifndef FQG_H
#define FQG_H

#include <stddef.h>
#include <stdint.h>

#include "fqg_codes.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FQG_ABI_VERSION 1

/* library-level failures (negative); record-level outcomes are enum fqg_code */
#define FQG_ERR_NO_DEVICE (-1)   /* no usable gfx950 device / HIP runtime failure at open */
#define FQG_ERR_HIP (-2)         /* a HIP call failed; see fqg_last_error() */
#define FQG_ERR_ARG (-3)         /* invalid argument */
#define FQG_ERR_NOMEM (-4)       /* device or pinned allocation failed */
#define FQG_ERR_STATE (-5)       /* call sequence violated (e.g. no frame to index) */

#define FQG_MEM_HOST 0
#define FQG_MEM_DEVICE 1
#define FQG_MEM_DEVICE_INDEXED 2 /* fqg_umi_count: the record stream AND the array of record offsets are device memory */

typedef struct fqg_ctx fqg_ctx;
typedef struct fqg_acc fqg_acc;

/* ---- context -------------------------------------------------------------------------- */
int fqg_open(int device_ordinal, fqg_ctx **ctx);
void fqg_close(fqg_ctx *ctx);
const char *fqg_last_error(const fqg_ctx *ctx);
int fqg_abi_version(void);
/* launch on a caller-provided hipStream_t (passed as void*); NULL restores the context's own */
int fqg_set_stream(fqg_ctx *ctx, void *hip_stream);
int fqg_synchronize(fqg_ctx *ctx);
/* pinned host memory for staging file contents */
void *fqg_host_alloc(fqg_ctx *ctx, size_t bytes);
void fqg_host_free(fqg_ctx *ctx, void *p);

/* ---- per-file state decided from the first record --------------------------------------
 * Mirrors the FASTQ_FILE fields that steer validation (src/fastq.h:125-130): the read-name
 * format and the colour-space flag are fixed by the first record of a file
 * (src/fastq.c:459-485) and is_pe by the program (src/fastq.c:88-90).  The host decides them
 * (fqg_probe_first_record) and passes them to every bulk call on that file. */
typedef struct {
  int32_t is_pe;           /* FASTQ_FILE.is_pe */
  int32_t readname_format; /* FQG_NAME_* */
  int32_t space;           /* FQG_SPACE_* */
  int32_t reserved;
} fqg_file_state;

/* Host-side, no GPU: src/fastq.c:459-485 + :666-754 on one header line (without the leading
 * '@', NUL-terminated) and one sequence line.  Returns the format / the space. */
int fqg_probe_readname_format(const char *hdr_after_at);
int fqg_probe_space(const char *seq_line);
/* Both probes on the first record of a host-resident image (first two lines, read with the
 * reference's gzgets limits, src/fastq.c:249-251).  Fills readname_format and space; is_pe is
 * copied through.  Returns 0, or FQG_ERR_ARG when the image has no complete first two lines. */
int fqg_probe_first_record(const void *host_image, uint64_t nbytes, int is_pe, fqg_file_state *out);

/* ---- statistics accumulator -------------------------------------------------------------
 * Device-resident counterpart of FASTQ_FILE's counters (src/fastq.h:116-122): min/max/last
 * read length, min/max quality byte, number of reads, and the rdlen_ctr[] histogram.
 * Replaces fastq_new_entry_stats() (src/fastq.c:97-110) and the quality-range update in
 * fastq_validate_entry() (src/fastq.c:372-378). */
typedef struct {
  uint64_t num_rds;  /* records counted */
  uint64_t min_rl;   /* FASTQ_FILE.min_rl: strlen(seq) units, i.e. including the '\n' */
  uint64_t max_rl;
  uint64_t min_qual; /* as the reference's unsigned long: bytes >= 0x80 read 0xFFFFFF80.. */
  uint64_t max_qual;
} fqg_file_stats;

int fqg_acc_create(fqg_ctx *ctx, fqg_acc **acc);
void fqg_acc_destroy(fqg_acc *acc);
int fqg_acc_reset(fqg_acc *acc);
int fqg_acc_read(fqg_acc *acc, fqg_file_stats *out);
/* count of reads whose length (strlen(seq) units) is `len`, i.e. FASTQ_FILE.rdlen_ctr[len] */
int fqg_acc_hist_nonzero(fqg_acc *acc, uint64_t *lens, uint64_t *counts, size_t cap, size_t *n);
/* median_rl(), src/fastq_info.c:39-55, over one or two accumulators (b may be NULL).
 * `weight` multiplies every count and num_rds (index mode counts each record twice, F7). */
int fqg_acc_median(fqg_acc *a, fqg_acc *b, uint64_t *median);
/* multi-GPU: element-wise merge of another accumulator's host-exported state */
int fqg_acc_export(fqg_acc *acc, void *buf, size_t cap, size_t *used);
int fqg_acc_merge(fqg_acc *acc, const void *buf, size_t used);

/* A context keeps the device buffers of its calls (framing tables of the largest image seen, capture records, the plan
 * and the output text of the tile kernels ...) so that the next call does not allocate again.  fqg_release_scratch gives
 * them back - between phases of a program that needs the memory for something else (the next phase allocates what it
 * needs again).  The context's current frame goes with them; retained frames, indexes, accumulators and censuses stay. */
int fqg_release_scratch(fqg_ctx *ctx);

/* ---- framing + validation ---------------------------------------------------------------
 * Replaces the read/validate loop: fastq_read_entry() (src/fastq.c:245-261) for every record
 * of the image followed by fastq_validate_entry() (src/fastq.c:300-392), as driven by
 * validate_single_fastq_file() (src/fastq_info.c:155-176) and the other fastq_info loops.
 *
 * `image` must start at a record boundary.  With final != 0 the image ends the file: a last
 * line without '\n' is a line, and an incomplete last record is FQG_E_TRUNCATED.  With
 * final == 0 an incomplete tail is not an error: `consumed` tells how many bytes were used and
 * the caller prepends the rest to its next image.
 *
 * The outcome reported is the FIRST one in file order, exactly as the serial loop would hit
 * it (record index, then the check order inside fastq_validate_entry).  Statistics are added
 * to `acc` for every record of the image; they are only meaningful when code == FQG_OK (the
 * reference exits at the first error and never prints them otherwise). */
typedef struct {
  uint64_t n_records; /* complete records framed (and validated) */
  uint64_t n_lines;   /* lines seen */
  uint64_t consumed;  /* bytes covered by the n_records records */
  uint64_t record;    /* 0-based index (within the image) of the first failing record */
  uint64_t aux0;      /* code-specific: offending byte / slen */
  uint64_t aux1;      /*                qlen */
  int32_t code;       /* enum fqg_code; FQG_OK if every record passed */
  int32_t stopped;    /* 1: a record starting with a NUL byte ended the file early (src/fastq.c:250) */
  int32_t path;       /* which device path ran: 1 = exact wave-per-record, 2 = tiled two-pass path,
                         3 = single-pass (streaming) path */
  int32_t tail_lines; /* final images: lines (1..3) of an incomplete last record, whether or not an
                         earlier record already failed; 0 if the image ends at a record boundary */
} fqg_validate_result;

/* checks bitmask */
#define FQG_VALIDATE_DEFAULT 0u
#define FQG_VALIDATE_FORCE_EXACT 1u /* always use the wave-per-record kernel */
#define FQG_VALIDATE_NO_STATS 2u    /* do not touch acc (acc may be NULL) */
#define FQG_VALIDATE_FRAME_ONLY 8u  /* build the line index only: no checks, no statistics (inputs that the
                                       caller vouches for, src/fastq_pre_barcodes.c:541-543) */
#define FQG_VALIDATE_TWO_PASS 16u    /* never take the single-pass (streaming) framing path */
#define FQG_VALIDATE_NAMES 32u      /* the names of this image go to an index next (fqg_index_insert_unique,
                                       fqg_index_match_delete, fqg_index_probe_delete on the frame this call leaves): the
                                       single-pass framing also copies every header line into a 64-byte record while the
                                       bytes are on the chip, and the index calls work from those records instead of going
                                       back to the image.  Results are the same with or without it. */
#define FQG_VALIDATE_NAME_DIGESTS 256u /* FQG_VALIDATE_NAMES for an index that will only be tested for uniqueness
                                       (fqg_index_expect_lookups(index, 0): fastq_info on ONE file, src/fastq_info.c:289-300):
                                       the single-pass framing canonicalises and hashes every header line itself (under the
                                       `state` of this call) and keeps 16 bytes per header instead of 64.  An insert into any
                                       other index, and a look-up, do not use digests: they read the names through the
                                       line index.  Results are the same with or without it. */
#define FQG_VALIDATE_REFRAMED 64u    /* the host has already cut the image at the reference's gzgets() limits
                                       (src/fastq.c:249-253; fastq_utils_amd/host/fq_input.h, Reframer): a piece that gzgets
                                       would return without its newline is followed by "\0\n" - the NUL the reference's
                                       buffer holds behind it, and a newline that only frames.  No line is held against the
                                       limits then (FQG_E_LINE_TOO_LONG is never reported). */
#define FQG_VALIDATE_INDEX 128u     /* the frame this call leaves will be used (fqg_frame_retain, fqg_frame_records, the name
                                       calls): write the whole line index now.  Without it the single-pass framing checks
                                       the records but stores only the end of the index (32 bytes per record that a call
                                       which only validates never reads back), and the first call that needs the frame
                                       writes the rest - same results, the line kernels run a second time. */
#define FQG_VALIDATE_COUNT_TWICE 4u /* every record counts twice in acc: the index loop runs
                                       fastq_new_entry_stats in both fastq_read_next_entry and
                                       fastq_validate_entry (src/fastq.c:415,432) */

int fqg_validate(fqg_ctx *ctx, fqg_acc *acc, const void *image, uint64_t nbytes, int mem,
                 int final, const fqg_file_state *state, uint32_t flags,
                 fqg_validate_result *out);

/* Record descriptors of the image framed by the last fqg_validate() call on this context.
 * One per record, 32 bytes: what FASTQ_ENTRY (src/fastq.h:97-108) holds besides the bytes. */
typedef struct {
  uint64_t offset;   /* FASTQ_ENTRY.offset: byte offset of the '@' line in the image */
  uint32_t hdr1_len; /* bytes in each of the four lines, '\n' included when present */
  uint32_t seq_len;
  uint32_t hdr2_len;
  uint32_t qual_len;
  uint32_t read_len; /* FASTQ_ENTRY.read_len = strlen(seq) */
  uint32_t reserved;
} fqg_record;

int fqg_frame_records(fqg_ctx *ctx, uint64_t first, uint64_t count, fqg_record *out, int mem);

/* ---- frames ------------------------------------------------------------------------------
 * fqg_validate() leaves a "frame" in the context: the image on the device plus its line index.
 * The next fqg_validate() call overwrites it.  fqg_frame_retain() moves the frame's buffers into
 * an object of its own so that it outlives later calls (images the caller passed as
 * FQG_MEM_DEVICE stay the caller's and must stay alive as long as the frame is used). */
typedef struct fqg_frame fqg_frame;
int fqg_frame_retain(fqg_ctx *ctx, fqg_frame **out);
void fqg_frame_release(fqg_frame *frame);
uint64_t fqg_frame_n_records(const fqg_frame *frame);
/* make a retained frame the context's current frame again (for the calls that work on "the current frame", e.g.
 * fqg_index_probe_delete of one file's records against the other file's index); it stays owned by whoever holds it */
int fqg_frame_make_current(fqg_ctx *ctx, const fqg_frame *frame);

/* ---- read-name index --------------------------------------------------------------------
 * Replaces hash.c as fastq.c uses it: new_hashtable() + the fastq_index_readnames() loop
 * (src/fastq.c:396-439: fastq_get_readname, fastq_index_lookup_header, new_indexentry) and the
 * lookup / fastq_index_delete loop over the second file (src/fastq_info.c:333-356).
 * Names are the canonical read names of fastq_get_readname() (src/fastq.c:488-512) under the
 * given file state; equality is decided on the name bytes. */
typedef struct fqg_index fqg_index;
typedef struct {
  uint64_t n_entries; /* hashtable.n_entries after the call */
  uint64_t index_mem; /* what the reference adds up in index_mem (src/fastq.c:609, src/fastq_info.c:293) */
  uint64_t record;    /* index, within the frame, of the first record with a finding */
  int32_t code;       /* FQG_OK, FQG_E_WRONG_HEADER, FQG_E_DUP_NAME, FQG_E_UNPAIRED, FQG_E_NAME_MISMATCH */
  int32_t reserved;
} fqg_index_result;

int fqg_index_create(fqg_ctx *ctx, uint64_t expected_names, fqg_index **out);
void fqg_index_destroy(fqg_index *index);
/* Before the first insert: yes = 0 tells the index that nobody will look names up in it (fastq_info on ONE file only
 * tests the names for uniqueness, src/fastq_info.c:289-300) - it then keeps no name record per entry (64 bytes each,
 * which the look-ups of a second file are decided on).  Look-ups still work without them, through the images. */
int fqg_index_expect_lookups(fqg_index *index, int yes);
/* Insert the name of every record of the context's current frame (the frame is retained by the
 * index).  Finding: the first record whose header does not start with '@' (FQG_E_WRONG_HEADER,
 * src/fastq.c:448) or whose name is already in the index (FQG_E_DUP_NAME, src/fastq.c:422),
 * whichever record comes first. */
int fqg_index_insert_unique(fqg_ctx *ctx, fqg_index *index, const fqg_file_state *state,
                            fqg_index_result *out);
/* For every record of the current frame: look its name up and delete the entry.  Finding: the
 * first record with a wrong header or without a partner (FQG_E_UNPAIRED, src/fastq_info.c:338). */
int fqg_index_match_delete(fqg_ctx *ctx, fqg_index *index, const fqg_file_state *state,
                           fqg_index_result *out);
/* The same pass with one answer per record (the lookup + fastq_index_delete of src/fastq_filterpair.c:150-170,
 * 196-216): match[r] (host, one per record of the current frame) = which inserted record's entry record r found
 * and took - its index in insertion order over all inserted frames - or UINT64_MAX when the name is not in the
 * index or an earlier record (of this or an earlier frame of the asking file) took it (FQG_NO_MATCH), or
 * FQG_MATCH_WRONG_HEADER for a record whose header does not start with '@' (src/fastq.c:448). */
#define FQG_NO_MATCH UINT64_MAX
#define FQG_MATCH_WRONG_HEADER (UINT64_MAX - 1)
int fqg_index_probe_delete(fqg_ctx *ctx, fqg_index *index, const fqg_file_state *state, uint64_t *match,
                           fqg_index_result *out);
/* alive[g] = 1 for every inserted record g (insertion order) whose entry nobody has taken yet: what a lookup of
 * the file's own names finds after the pairing loop (src/fastq_filterpair.c:196-216); cap >= records inserted */
int fqg_index_alive(fqg_ctx *ctx, fqg_index *index, uint8_t *alive, uint64_t cap);
/* Of the last fqg_index_insert_unique / _match_delete / _probe_delete call on this context: how many records had their
 * name taken straight from a capture record of the streaming pass (FQG_VALIDATE_NAMES); the rest went through the line
 * index.  A measurement / test aid: results do not depend on it. */
uint64_t fqg_index_names_captured(const fqg_ctx *ctx);
/* the frames an index has retained, in insertion order (borrowed: they live as long as the index) */
uint64_t fqg_index_n_frames(const fqg_index *index);
const fqg_frame *fqg_index_frame(const fqg_index *index, uint64_t k);
/* Pairwise name agreement without an index.  b == NULL: records 2k and 2k+1 of frame a
 * (interleaved input, src/fastq_info.c:81-91, finding FQG_E_UNPAIRED at pair k -> record 2k);
 * otherwise record k of a against record k of b (src/fastq_info.c:133-138, FQG_E_NAME_MISMATCH). */
int fqg_names_compare(fqg_ctx *ctx, const fqg_frame *a, const fqg_file_state *state_a, const fqg_frame *b,
                      const fqg_file_state *state_b, fqg_index_result *out);

/* ---- read names across GPUs ------------------------------------------------------------------
 * The unique-name test of fastq_index_readnames() (src/fastq.c:422-425) when the records of a file
 * are sharded over several GPUs.  Every rank turns the canonical names of its frames
 * into (64-bit fingerprint, GLOBAL record index) pairs bucketed by owner rank; the buckets travel
 * with one all-to-all (RCCL, done by the caller: fastq_utils_amd/dist.py); the owner sorts what
 * it received by fingerprint and lists, per value held more than once, its holders as CANDIDATE duplicates; candidates are
 * confirmed on the name bytes (fqg_frame_name on their home ranks), so equality is still decided on
 * the bytes and the finding is the one the serial loop would make over the concatenated shards. */
typedef struct {
  uint64_t fp, idx;
} fqg_fp;
typedef struct fqg_fpset fqg_fpset;
#define FQG_MAX_OWNERS 64
/* owner rank of a fingerprint */
uint32_t fqg_fp_owner(uint64_t fp, uint32_t n_owners);
/* Fingerprints of the records of a retained frame (NULL: the context's current frame), written to
 * `out` (DEVICE memory, room for the frame's records) as n_owners buckets back to back; counts[o]
 * (host) = size of bucket o.  record_base = global index of the frame's first record.  Records whose
 * header does not start with '@' are left out. */
int fqg_names_fingerprints(fqg_ctx *ctx, const fqg_frame *frame, const fqg_file_state *state, uint64_t record_base,
                           uint32_t n_owners, void *out_device, uint64_t *counts);
/* the same, and *name_bytes = sum over the frame's records of the `len` the reference accounts per indexed name
 * (src/fastq.c:609: what "Memory used in indexing" is made of) */
int fqg_names_fingerprints_acct(fqg_ctx *ctx, const fqg_frame *frame, const fqg_file_state *state, uint64_t record_base,
                                uint32_t n_owners, void *out_device, uint64_t *counts, uint64_t *name_bytes);
/* Device buffers for the exchange, and the exchange itself when the owners are contexts of ONE process (the drop-in
 * programs with FQGPU_DEVICES=0,1,..: fastq_utils_amd/host/fq_names_multi.h): a copy from a buffer of one context
 * into a buffer of another - over xGMI between two GPUs - that has arrived when the call returns.  Ranks in
 * different processes use RCCL instead (fastq_utils_amd/dist.py). */
void *fqg_device_alloc(fqg_ctx *ctx, uint64_t bytes);
void fqg_device_free(fqg_ctx *ctx, void *p);
int fqg_device_copy(fqg_ctx *dst_ctx, void *dst, fqg_ctx *src_ctx, const void *src, uint64_t bytes);
int fqg_fpset_create(fqg_ctx *ctx, uint64_t expected, fqg_fpset **out);
void fqg_fpset_destroy(fqg_fpset *set);
/* fps: DEVICE memory, n fqg_fp */
int fqg_fpset_insert(fqg_ctx *ctx, fqg_fpset *set, const void *fps_device, uint64_t n);
/* Pairing with the names themselves (src/fastq.c:577-587: the key match is followed by strcmp).
 * fqg_names_fingerprints_named writes, beside every pair, a record of FQG_NAME_REC_BYTES at the same place of a second
 * device array: the name's first 56 bytes, zero padded, and its length; the two arrays travel together, and a set whose
 * entries were all inserted with fqg_fpset_insert_named counts a holder and its asker as a pair only when the two names
 * are the same bytes (names beyond 56 bytes, and names that differ under equal fingerprints, are left to the caller's
 * resolution by record index like every run the fingerprints cannot decide). */
#define FQG_NAME_REC_BYTES 64
int fqg_names_fingerprints_named(fqg_ctx *ctx, const fqg_frame *frame, const fqg_file_state *state, uint64_t record_base,
                                 uint32_t n_owners, void *out_device, void *names_device, uint64_t *counts,
                                 uint64_t *name_bytes);
int fqg_fpset_insert_named(fqg_ctx *ctx, fqg_fpset *set, const void *fps_device, const void *names_device, uint64_t n);
/* Names by POSITION, tried before the exchange above (the file-2 loop of fastq_info, src/fastq_info.c:333-356, on mate
 * files that hold their reads in one order - the usual case - finds the partner of record i in the entry of record i):
 * fqg_frame_name_records writes the FQG_NAME_REC_BYTES records of records first .. first + n - 1 of a retained frame
 * (canonical names under `state`) to out_device; the caller copies them to the context that holds the same records of
 * the other file (fqg_device_copy) and fqg_frame_names_equal counts there the records whose name is the record's at
 * the same place (*n_equal) and those that agree in all a record holds but are longer than its 56 bytes
 * (*n_undecided).  n_equal == the records of both files (and file 1 without a repeated name): every read is paired. */
int fqg_frame_name_records(fqg_ctx *ctx, const fqg_frame *frame, const fqg_file_state *state, uint64_t first, uint64_t n,
                           void *out_device);
int fqg_frame_names_equal(fqg_ctx *ctx, const fqg_frame *frame, const fqg_file_state *state, uint64_t first, uint64_t n,
                          const void *recs_device, uint64_t *n_equal, uint64_t *n_undecided);
/* After every insert: pairs[2k], pairs[2k+1] (host) = (earliest holder, another holder) for every
 * fingerprint value held by more than one of the inserted records; *n_found may exceed cap. */
int fqg_fpset_candidates(fqg_ctx *ctx, fqg_fpset *set, uint64_t *pairs, uint64_t cap, uint64_t *n_found);
/* Pairing across GPUs (the file-2 loop of fastq_info, src/fastq_info.c:333-356): the set holds the
 * fingerprints of file 1 AND of file 2, file-2 entries with FQG_FP_FILE2 set in their index.  Runs of
 * equal fingerprints are classified on the device: a pair, names nobody asked for, askers without a
 * holder (first_unpaired = the smallest index among them, without the flag).  Runs of any other shape
 * (a name asked for twice, a hash collision) are exported as (run id, index) pairs for the exact
 * resolution on the name bytes; n_complex counts their entries (it may exceed cap). */
#define FQG_FP_FILE2 (1ull << 63)
typedef struct {
  uint64_t matched, leftover, unpaired, first_unpaired, n_complex;
} fqg_pair_summary;
int fqg_fpset_pair_runs(fqg_ctx *ctx, fqg_fpset *set, fqg_pair_summary *out, uint64_t *entries, uint64_t cap);
/* canonical read name (fastq_get_readname, src/fastq.c:488-512) of record `record` of a retained frame,
 * NUL-terminated into out[cap]; returns its length or a negative FQG_ERR_* */
int64_t fqg_frame_name(fqg_ctx *ctx, const fqg_frame *frame, const fqg_file_state *state, uint64_t record, char *out,
                       uint64_t cap);

/* ---- barcode extraction (fastq_pre_barcodes) ----------------------------------------------
 * Replaces the body of the main loop of fastq_pre_barcodes (src/fastq_pre_barcodes.c:594-727):
 * for iteration k, record first_record[x] + k*step (+1) of every input x must carry the same
 * read name (src/fastq_pre_barcodes.c:606-635); UMI / cell / sample are cut out (get_barcode,
 * :218-259) - a read too short or with a base below min_qual is discarded -; kept reads are written
 * with the tags in the read name (add_tags2readname :192-216, slice_read :160-190) as FASTQ text, or
 * as SAM lines (:657-710).  Inputs are retained frames; index 1..5 = read1, read2, index1..3 as in
 * the reference's READ_IDX.  The two references of --interleaved advance two records per iteration. */
typedef struct {
  int32_t present[6];
  int32_t interleaved[2];      /* {0,0}: none; else the two file references of --interleaved */
  int32_t umi_read, cell_read, sample_read; /* 1..5, -1 = not set */
  int32_t phred_encoding, min_qual;
  int32_t out_sam, tenx;       /* --sam, --10x */
  int32_t emit[3];             /* FASTQ mode: outfile1 / outfile2 wanted */
  int64_t umi_offset, umi_size, cell_offset, cell_size, sample_offset, sample_size; /* offset -1 = not set */
  int64_t read_offset[3], read_size[3];
} fqg_barcode_params;

typedef struct {
  uint64_t n_done;        /* iterations consumed: all of them, or up to a finding, or (interleaved input)
                             up to and including the first discarded read, after which the reference's
                             file pointers are out of step (src/fastq_pre_barcodes.c:653 vs :722) */
  uint64_t n_discarded;   /* among the n_done */
  uint64_t n_short;       /* of those, "Warning: Read too short - barcode not found" cases */
  uint64_t out_bytes[3];  /* [0] SAM text, [1] / [2] FASTQ text of outfile1 / outfile2 */
  uint64_t iteration;     /* finding: iteration (== n_done) */
  int32_t code;           /* FQG_OK, FQG_E_NAME_MISMATCH, FQG_E_WRONG_HEADER */
  int32_t file;           /* finding: which input (1..5) */
} fqg_barcode_result;

int fqg_barcodes_transform(fqg_ctx *ctx, const fqg_frame *const frames[6], const fqg_file_state states[6],
                           const uint64_t first_record[6], const fqg_barcode_params *params,
                           uint64_t n_iterations, uint64_t first_read_number, fqg_barcode_result *out);
/* copy output `which` (0 SAM, 1, 2) of the last transform to host memory */
int fqg_barcodes_output(fqg_ctx *ctx, int which, void *host_dst, uint64_t nbytes);
/* ... the same copy on a stream of its own: it returns at once and runs beside whatever the context does next in the
 * OTHER direction of the link (fqg_validate of the next piece of input: host to device).  host_dst must be pinned
 * (fqg_host_alloc) for the copy to be asynchronous.  fqg_barcodes_output_wait returns when every such copy has
 * landed; the next fqg_barcodes_transform / fqg_records_filter / fqg_records_gather of the context, which write
 * the same device buffers, wait for them by themselves. */
int fqg_barcodes_output_begin(fqg_ctx *ctx, int which, void *host_dst, uint64_t nbytes);
int fqg_barcodes_output_wait(fqg_ctx *ctx);

/* ---- FASTQ -> (cell, UMI) without the BAM round trip ---------------------------------------
 * In the reference's pipeline (sh/fastq2bam:116-273) the barcodes of a kept read travel inside its name
 * (add_tags2readname, src/fastq_pre_barcodes.c:192-216) through the aligner into a BAM file, where bam_add_tags parses
 * them out again (get_barcodes, src/bam_add_tags.c:43-99) as CR / RX tags, which bam_umi_count packs with char2uint_64
 * (src/bam_umi_count.c:364-382).  None of that depends on the alignment.  fqg_barcodes_census - called behind the
 * fqg_barcodes_transform of the same batch, with the same frames / states / first_record / params and that call's
 * n_done - appends, in device memory, one (cell, UMI) pair of packed values per read the transform kept and that
 * bam_umi_count would count: a read without a UMI has none (src/bam_umi_count.c:960), a read without a cell barcode
 * has the cell 0 (char2uint_64 of a missing tag), a value that holds a '_' ends get_barcodes (no tags: no pair).
 * fqg_census_finish sorts the pairs by (cell, UMI) and makes one line per cell, in ascending order of the packed cell:
 * its reads and its distinct UMIs - bam_umi_count's per-cell totals before a gene tag exists. */
typedef struct fqg_census fqg_census;
typedef struct {
  uint64_t cell, reads, umis;
} fqg_census_cell;
int fqg_census_create(fqg_ctx *ctx, fqg_census **out);
void fqg_census_destroy(fqg_census *census);
int fqg_barcodes_census(fqg_ctx *ctx, fqg_census *census, const fqg_frame *const frames[6], const fqg_file_state states[6],
                        const uint64_t first_record[6], const fqg_barcode_params *params, uint64_t n_done,
                        uint64_t *n_added);
int fqg_census_finish(fqg_ctx *ctx, fqg_census *census, uint64_t *n_pairs, uint64_t *n_cells);
/* the lines / the pairs (sorted once the census is finished) copied to host memory; at most cap of them */
int fqg_census_cells(fqg_ctx *ctx, fqg_census *census, fqg_census_cell *out, uint64_t cap);
int fqg_census_pairs(fqg_ctx *ctx, fqg_census *census, uint64_t *cells, uint64_t *umis, uint64_t cap);
/* the pairs where they live: device pointers to the packed cells (which = 0) / UMIs (1), n_pairs of each */
const void *fqg_census_device_pairs(const fqg_census *census, int which);

/* ---- whitelist membership of a barcode (BASELINE configs[2]: "known_cells whitelist") ------------------
 * The cell barcode that fastq_pre_barcodes cuts out of a read is what bam_umi_count later packs with char2uint_64
 * (src/bam_umi_count.c:364-382) and tests with valid_barcode (:523-535) against the table load_whitelist (:543-579)
 * filled from the --known_cells file.  fqg_whitelist_create takes the packed lines of such a file (fqg_pack_barcode
 * on every line that fgets returns non-empty, in any order; repeats are fine); fqg_barcodes_whitelist answers, for
 * records first_record, first_record + step, ... of a retained frame, whether the `size` characters at `offset` of
 * the sequence line are a member.  A read too short for them (get_barcode's bounds, src/fastq_pre_barcodes.c:232)
 * has no barcode: not valid, counted in n_short.  valid (host memory, one byte per record) may be NULL. */
typedef struct fqg_whitelist fqg_whitelist;
typedef struct {
  uint64_t n_records, n_valid, n_short;
} fqg_whitelist_result;
int fqg_whitelist_create(fqg_ctx *ctx, const uint64_t *packed, uint64_t n, fqg_whitelist **out);
void fqg_whitelist_destroy(fqg_whitelist *wl);
int fqg_barcodes_whitelist(fqg_ctx *ctx, const fqg_frame *frame, uint64_t first_record, uint64_t step, uint64_t n_records,
                           int64_t offset, int64_t size, const fqg_whitelist *wl, uint8_t *valid,
                           fqg_whitelist_result *out);

/* ---- per-record filters (fastq_filter_n, fastq_trim_poly_at) ------------------------------------
 * Replaces the record loops of fastq_filter_n (src/fastq_filter_n.c:75-91: count N/n in the sequence,
 * drop the record when there are more than read_len * max_n / 100) and of fastq_trim_poly_at
 * (src/fastq_trim_poly_at.c:77-119 trim_poly_at, :214-222 the min_len test and the counters).
 * Works on records [first_record, first_record + n_records) of a retained frame; the kept (and
 * trimmed) records are left as FASTQ text in device memory, fetched with fqg_records_filter_output.
 * read_len is strlen(seq) with its '\n', as in the reference (src/fastq.c:259). */
typedef struct {
  int32_t mode;             /* FQG_FILTER_N or FQG_FILTER_POLY_AT (fqg_codes.h) */
  uint32_t max_n_percent;   /* FILTER_N: -n (0: any N drops the record); values above 100 count as 100 */
  int64_t min_poly_at_len;  /* POLY_AT: --min_poly_at_len (<= 0: nothing is trimmed) */
  int64_t min_len;          /* POLY_AT: --min_len, compared as the reference does (unsigned) */
} fqg_filter_params;
typedef struct {
  uint64_t n_records, n_kept, n_trimmed, n_discarded;
  uint64_t out_bytes;
} fqg_filter_result;
int fqg_records_filter(fqg_ctx *ctx, const fqg_frame *frame, uint64_t first_record, uint64_t n_records,
                       const fqg_filter_params *params, fqg_filter_result *out);
int fqg_records_filter_output(fqg_ctx *ctx, void *host_dst, uint64_t nbytes);

/* ---- ordered gather of records (fastq_filterpair) -----------------------------------------------
 * The output side of fastq_filterpair (src/fastq_filterpair.c:150-216): records of a retained frame written in a
 * given order - what fastq_write_entry (src/fastq.c:265-272) / fastq_quick_copy_entry (:125-157) produce record by
 * record.  records[k] = index in the frame of the k-th record of the output; the text stays on the device until
 * fqg_records_gather_output copies it. */
int fqg_records_gather(fqg_ctx *ctx, const fqg_frame *frame, const uint64_t *records, uint64_t n, uint64_t *out_bytes);
int fqg_records_gather_output(fqg_ctx *ctx, void *host_dst, uint64_t nbytes);

/* ---- UMI counting (bam_umi_count) ---------------------------------------------------------------
 * Replaces the alignment loop of bam_umi_count (src/bam_umi_count.c:942-1060: filters, aux tags,
 * char2uint_64 :364-382, the label maps :143-260, process_entry :444-509) and the output decisions
 * of cell2MM (:666-705) / write2MM (:584-663).  Input: the inflated BAM stream of a file (the host
 * inflates BGZF; libbam does that inside bam_read1) and the offset of every alignment record in
 * it (fqg_bam_index_records).  The UMIs of a (cell, feature) are kept as the reference's RL_Tree keeps them
 * (src/range_list.c: a set except where its array insert overwrites a node, see fqg_umi_params.strict_set);
 * counters are float32 and are added in record order, as the reference's `float` fields are. */
typedef struct {
  char feat_tag[2], cell_tag[2], umi_tag[2];      /* -x / -X / RX or UB (--10x) */
  char reserved[2];
  int32_t sorted_by_cell, uniq_mapped_only;       /* --not_sorted_by_cell clears the first, --uniq_mapped sets the second */
  uint32_t max_cells, max_features, min_reads, min_umis;
  const uint64_t *known_umis, *known_cells;       /* packed (fqg_pack_barcode) whitelist lines in file order, or NULL */
  uint64_t n_known_umis, n_known_cells;
  int32_t defer_output;   /* 1: count only; the lines are made by fqg_umi_emit (shards of one file: the output
                             rules need the GLOBAL feature ids, known only after every shard has counted) */
  int32_t strict_set;     /* 0 (default): the UMIs of a (cell, feature) are kept as the reference's RL_Tree keeps them
                             (src/range_list.c), which loses / invents members in rare arrival orders - bit-exact with
                             the reference program; 1: the set src/range_list.h:150-162 documents (an extra) */
  /* Shards of one file (several GPUs): what the RL_Tree holds are the UMIs' IDS - dense, in order of first appearance
   * in the WHOLE file (blabel2id, src/bam_umi_count.c:225-260) - so a shard must number its UMIs as the file does.
   * umi_table_keys: packed UMIs (not in --known_umi), sorted ascending; umi_table_ids[i]: 1 + how many such UMIs the
   * file saw before keys[i] first appears.  NULL: number them in order of first appearance in this call.  A UMI of
   * the call that is missing from the table is FQG_ERR_ARG. */
  const uint64_t *umi_table_keys;
  const uint32_t *umi_table_ids;
  uint64_t n_umi_table;
  /* ... and db->tot_reads_obs / tot_umi_obs (src/bam_umi_count.c:490-507) are ONE float32 chain over the file: a shard
   * continues it from where the shards before it ended (db_start_*), beginning at its own first alignment (db_skip
   * alignments in front of that are history of earlier shards, see fastq_utils_amd/dist.py).  Only read when the call
   * has fractional increments (unit increments give min(count, 2^24)). */
  float db_start_reads, db_start_umi;
  uint64_t db_skip;
} fqg_umi_params;

typedef struct {
  uint64_t n_alignments, n_tags_found, n_umis_discarded, n_cells_discarded;
  uint64_t n_features, n_cells;
  uint64_t n_entries[2];  /* lines of the UMI-count / read-count matrix */
  uint64_t total[2];      /* sum of the truncated counts of those lines (third header field in sorted mode) */
  float tot_reads, tot_umi; /* db->tot_reads_obs / tot_umi_obs (:1086-1087) */
  int32_t code;           /* FQG_OK or FQG_E_UMI_*: the first finding in record order; nothing else is valid then */
  int32_t reserved;
  uint64_t record, aux;   /* alignment index of the finding, offending id */
  uint64_t n_counted, n_new; /* alignments that reached process_entry / that brought a new UMI */
  int32_t unit_increments;   /* 1: every increment was 1.0 (then tot_reads = min(n_counted, 2^24) as a float) */
  int32_t reserved3;
  /* RL_Tree replay (strict_set = 0): (cell, feature) sets replayed node for node because the reference's tree
   * overwrites a node there (src/range_list.c:287-301,338-339); alignments whose "new UMI" decision differs from a
   * set's; reads of tree memory the reference never wrote (its output then depends on its heap: 0 is read here);
   * 1 if a replay did not fit this library's limits (results then follow set semantics for that set) */
  uint64_t rl_replayed, rl_changed, rl_undefined, rl_unresolved;
} fqg_umi_result;

typedef struct {
  uint32_t row, col, value; /* feature id (0 in unsorted mode, as the reference prints), cell id, rounded count */
} fqg_umi_entry;

/* char2uint_64 (:364-382) on a NUL- or newline-terminated string; host side, no GPU */
uint64_t fqg_pack_barcode(const char *s);
/* uint_642char (:342-360); out needs 20 bytes */
void fqg_unpack_barcode(uint64_t v, char *out);
/* Host side, no GPU: walk an inflated BAM stream (magic, header text, references, then records).
 * Writes the offset of every alignment record (of its block_size field) to offsets[0..*n) and the
 * end of the last complete record to *used.  Returns 0, FQG_ERR_ARG for a stream that is not BAM or a
 * too small `cap` (*n then holds the number needed). */
int fqg_bam_index_records(const void *stream, uint64_t nbytes, uint64_t *offsets, uint64_t cap, uint64_t *n,
                          uint64_t *used);
/* mem: where `stream` lies.  offsets is host memory - 8 bytes per alignment to upload, 1 ms of a 6 ms call on
 * BASELINE configs[3] - unless mem is FQG_MEM_DEVICE_INDEXED (a caller that keeps the inflated records in HBM keeps
 * their index there too). */
int fqg_umi_count(fqg_ctx *ctx, const void *stream, uint64_t nbytes, int mem, const uint64_t *offsets,
                  uint64_t n_records, const fqg_umi_params *params, fqg_umi_result *out);
/* After fqg_umi_count(defer_output = 1): apply the output rules with feature ids mapped through
 * feat_remap[0..n_remap) (local id -> printed id; NULL: identity) and cell ids shifted by cell_offset.
 * Fills n_entries / total of *out; the lines are read with fqg_umi_entries as usual. */
int fqg_umi_emit(fqg_ctx *ctx, const uint32_t *feat_remap, uint64_t n_remap, uint32_t cell_offset,
                 fqg_umi_result *out);
/* After fqg_umi_count(defer_output = 1), for a file whose cells are sharded over several GPUs: feat[i] = feature id of
 * alignment i (0: not counted); flags[f] = 1 when a (cell, feature f) set was replayed as the reference's RL_Tree
 * behaves (strict_set = 0) - the tree of such a feature carries state from one cell to the next
 * (src/range_list.c:187-198, src/bam_umi_count.c:418-441), so a later shard must see that feature's earlier
 * alignments to reproduce it (fastq_utils_amd/dist.py: umi_count_sharded).  flags needs n_features + 1 bytes. */
int fqg_umi_record_features(fqg_ctx *ctx, uint32_t *feat, uint64_t cap);
int fqg_umi_replayed_features(fqg_ctx *ctx, uint8_t *flags, uint64_t cap);
/* results of the last fqg_umi_count on this context */
int fqg_umi_features(fqg_ctx *ctx, char *names, uint64_t cap);       /* n_features x 25 bytes, NUL padded, id order */
int fqg_umi_cells(fqg_ctx *ctx, uint64_t *packed, uint64_t cap);     /* n_cells packed barcodes, id order */
/* the packed UMIs the call saw that are not in --known_umi, in order of first appearance; *n = how many (call with
 * packed = NULL, cap = 0 for the count) */
int fqg_umi_umis(fqg_ctx *ctx, uint64_t *packed, uint64_t cap, uint64_t *n);
int fqg_umi_entries(fqg_ctx *ctx, int which, fqg_umi_entry *out, uint64_t cap); /* which: 0 UMI counts, 1 read counts */

/* ---- measurement ------------------------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by hipEvents on the launch stream. */
typedef struct {
  char name[48];
  uint64_t launches;
  double total_ms;
} fqg_kernel_time;

int fqg_profile_enable(fqg_ctx *ctx, int on);
int fqg_profile_reset(fqg_ctx *ctx);
int fqg_profile_read(fqg_ctx *ctx, fqg_kernel_time *out, size_t cap, size_t *n);

/* ---- synthetic data (bench / tests) -----------------------------------------------------
 * Fills a device buffer with `n_records` seeded FASTQ records of fixed geometry
 * (CASAVA-1.8 names unique by index, `read_len` bases uniform over ACGT with
 * 0.1% N, Phred 2..40 +33).  Every record has exactly fqg_synth_record_bytes(read_len) bytes.
 * Deterministic in (seed, first_index, i). */
uint64_t fqg_synth_record_bytes(uint32_t read_len);
int fqg_synth_fastq(fqg_ctx *ctx, void *device_out, uint64_t n_records, uint32_t read_len,
                    uint64_t first_index, uint64_t seed, int mate);

/* ---- bam_add_tags -----------------------------------------------------------------------------------
 * Replaces the alignment loop of bam_add_tags (src/bam_add_tags.c:250-294: get_barcodes :43-99 on the read name,
 * bam_aux_append of RX|UB / CR / BC, and with --tx of tx and GX) - the step between fastq_pre_barcodes and
 * bam_umi_count (sh/fastq2bam:116-273).  Input as for fqg_umi_count: the inflated BAM stream and the offset of every
 * alignment (fqg_bam_index_records).  Output: the alignment records with their new tags, back to back, in input
 * order; it stays on the device until fqg_bam_add_tags_output copies it.  The header (everything before the first
 * alignment) is not touched: the caller writes it as it read it (bam_header_write after bam_header_read).
 * The transcript -> gene map (--tx_2_gx, :203-232, get_gene :118-129) is resolved by the caller once per reference
 * of the header: gx_len[t] = FQG_NO_GENE when reference t has no gene. */
#define FQG_NO_GENE 0xFFFFFFFFu
typedef struct {
  int32_t tenx;           /* --10x: the UMI goes to UB instead of RX (src/sam_tags.h:40-47) */
  int32_t tx_tag;         /* --tx */
  uint32_t n_targets;     /* references of the header */
  uint32_t reserved;
  const uint32_t *tx_off; /* [n_targets] header->target_name[t] = names + tx_off[t], tx_len[t] bytes */
  const uint32_t *tx_len;
  const uint32_t *gx_off; /* [n_targets] its gene, or gx_len[t] = FQG_NO_GENE */
  const uint32_t *gx_len;
  const char *names;
  uint64_t names_bytes;
} fqg_bam_tags_params;
typedef struct {
  uint64_t n_alignments;
  uint64_t n_tagged;  /* alignments whose name get_barcodes accepted */
  uint64_t out_bytes; /* size of the output record stream */
  uint64_t record;    /* first alignment with a finding */
  int32_t code;       /* FQG_OK, FQG_E_TAGS_NAME, FQG_E_TAGS_TID */
  int32_t reserved;
} fqg_bam_tags_result;
int fqg_bam_add_tags(fqg_ctx *ctx, const void *stream, uint64_t nbytes, int mem, const uint64_t *offsets,
                     uint64_t n_records, const fqg_bam_tags_params *params, fqg_bam_tags_result *out);
int fqg_bam_add_tags_output(fqg_ctx *ctx, void *host_dst, uint64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
