/*
 * fqg_codes.h - record-level outcome codes shared by the bulk C-ABI (fqg.h), the host
 * programs and the test oracle.
 *
 * Every code names the reference check it stands for (paths relative to the reference
 * checkout, nunofonseca/fastq_utils 0.25.3).  The numeric order has no meaning; the order in
 * which checks are applied to one record is fixed by the validator itself.
 */
#ifndef FQG_CODES_H
#define FQG_CODES_H

enum fqg_code {
  FQG_OK = 0,
  /* src/fastq.c:254-257  a record with fewer than four lines, or whose 2nd/3rd/4th line
   * starts with a NUL byte: "file truncated", exit status 1 */
  FQG_E_TRUNCATED = 1,
  /* src/fastq.c:448-451  fastq_get_readname() on a header that does not start with '@'
   * (reached before validation in the index / pairing loops): "wrong header", exit 3 */
  FQG_E_WRONG_HEADER = 2,
  /* src/fastq.c:422-425  read name already present in the index: "duplicated sequence" */
  FQG_E_DUP_NAME = 3,
  /* src/fastq.c:306-309  "sequence identifier should start with an @" */
  FQG_E_HDR1_AT = 4,
  /* src/fastq.c:310-313  "sequence identifier should be longer than 1" */
  FQG_E_HDR1_SHORT = 5,
  /* src/fastq.c:319-325  "invalid character '%c'"; aux0 = the byte */
  FQG_E_SEQ_CHAR = 6,
  /* src/fastq.c:327-341  "read contains both U and T bases" */
  FQG_E_SEQ_UT = 7,
  /* src/fastq.c:346-349  "read length too small"; aux0 = slen */
  FQG_E_LEN_SMALL = 8,
  /* src/fastq.c:355-358  "header2 wrong" */
  FQG_E_HDR2_PLUS = 9,
  /* src/fastq.c:363-370  "header2 differs from header1" */
  FQG_E_HDR2_DIFF = 10,
  /* src/fastq.c:380-383  "sequence and quality don't have the same length"; aux0/aux1 = slen/qlen */
  FQG_E_QLEN = 11,
  /* src/fastq.c:385-390  colour space: "sequence and quality length don't match" */
  FQG_E_QLEN_CS = 12,
  /* src/fastq_info.c:88-91,338-342  "unpaired read" */
  FQG_E_UNPAIRED = 13,
  /* src/fastq_info.c:135-138  "Readnames do not match across files" */
  FQG_E_NAME_MISMATCH = 14,
  /* Not a reference outcome, and not the programs' last word.  src/fastq.c:249-253 reads lines
   * with gzgets() limits of MAX_LABEL_LENGTH (headers) and MAX_READ_LENGTH (sequence, quality);
   * a longer line is split there and shifts the framing of everything after it.  The library
   * reports the first record with such a line; the caller then hands the input over again cut
   * the way gzgets cuts it (FQG_VALIDATE_REFRAMED) and gets the reference's findings. */
  FQG_E_LINE_TOO_LONG = 15,
  /* src/fastq.c:249-250  a record whose first line starts with a NUL byte ends the file
   * silently (fastq_read_entry returns 0).  Reported so that the caller can re-run on the
   * prefix that precedes it. */
  FQG_STOP_NUL = 16,
  /* bam_umi_count, src/bam_umi_count.c:1004-1007  a cell seen again after another one in sorted mode:
   * "The BAM file does not seem to be sorted by CR", exit 1 */
  FQG_E_UMI_NOT_SORTED = 17,
  /* :1047  assert(len1+1 < FEAT_ID_MAX_LEN): a feature name of 24 characters or more aborts */
  FQG_E_UMI_FEATURE_NAME = 18,
  /* process_entry :451-453, :455-457, :459-461  "Too many umi barcodes / cells / features"; aux = the id */
  FQG_E_UMI_TOO_MANY_UMIS = 19,
  FQG_E_UMI_TOO_MANY_CELLS = 20,
  FQG_E_UMI_TOO_MANY_FEATURES = 21,
  /* Not reference outcomes.  bam_add_tags, src/bam_add_tags.c:43-99: get_barcodes scans for '_' without a bound and
   * copies into 50-byte arrays; a read name whose value runs to the end of the alignment record, or is 50
   * characters or longer, makes the reference read / write memory it does not own.  Refused. */
  FQG_E_TAGS_NAME = 22,
  /* :275-277  header->target_name[tid] with tid beyond the header's references (--tx).  Refused. */
  FQG_E_TAGS_TID = 23
};

/* read-name formats, src/fastq.h:25-28 (INTEGERNAME and NOP share the value 2) */
#define FQG_NAME_DEFAULT 0
#define FQG_FILTER_N 1       /* fqg_filter_params.mode */
#define FQG_FILTER_POLY_AT 2
#define FQG_NAME_CASAVA18 1
#define FQG_NAME_INTEGER 2
#define FQG_NAME_NOP 2
#define FQG_NAME_UNDEF (-1)

/* src/fastq.h:48 */
#define FQG_SPACE_SEQ 0
#define FQG_SPACE_COLOUR 1
#define FQG_SPACE_UNDEF (-1)

/* src/fastq.h:30-46 */
#define FQG_MAX_READ_LENGTH 2500000
#define FQG_MAX_LABEL_LENGTH 1000
#define FQG_MAX_PHRED_QUAL 126

#endif
