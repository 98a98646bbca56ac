/*
 * fq_oracle.h - TEST INFRASTRUCTURE.  C interface of the CPU restatement in fq_oracle.c
 * (loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only).
 */
#ifndef FQ_ORACLE_H
#define FQ_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fastq_info options, src/fastq_info.c:214-248 */
#define FQO_FLAG_R 1 /* -r skip the duplicated-name check */
#define FQO_FLAG_S 2 /* -s files have the same ordering */
#define FQO_FLAG_E 4 /* -e empty files are fine */
#define FQO_FLAG_Q 8 /* -q unknown quality encoding is fine */

/* what the optional second positional argument is, src/fastq_info.c:258-267 */
#define FQO_ARG2_NONE 0
#define FQO_ARG2_FILE 1
#define FQO_ARG2_PE 2 /* the literal "pe": interleaved input */

typedef struct {
  const unsigned char *buf1;
  size_t n1;
  const char *name1; /* file name as it appears in messages */
  const unsigned char *buf2;
  size_t n2;
  const char *name2;
  int arg2_kind;
  int flags;
} fqo_job;

typedef struct {
  int32_t code;    /* enum fqg_code of the first record-level error, 0 if none */
  int32_t file;    /* 0/1: record came from the first input, 2: from the second */
  uint64_t record; /* 0-based record index within its input */
  uint64_t line;   /* the line number the message prints */
  uint64_t aux0, aux1;
} fqo_outcome;

typedef struct {
  uint64_t num_reads;       /* "Number of reads" */
  uint64_t min_rl, max_rl;  /* strlen(seq) units: one more than the printed values */
  uint64_t median_rl;       /* as returned by median_rl(), before the -1 */
  uint64_t min_qual, max_qual;
  uint64_t num_rds_counted; /* FASTQ_FILE.num_rds of the first input (twice the reads in index mode) */
} fqo_summary;

typedef struct {
  int exit_status;
  fqo_outcome first;
  fqo_summary summary; /* valid when exit_status == 0 and reads were found */
  char *out;           /* captured stdout */
  size_t out_len;
  char *err;           /* captured stderr */
  size_t err_len;
} fqo_result;

int fqo_fastq_info(const fqo_job *job, fqo_result *res);
/* fastq_filterpair (src/fastq_filterpair.c:38-228) on two images; flags & FQO_FLAG_S = the argument "sorted" */
int fqo_fastq_filterpair(const fqo_job *job, fqo_result *res, char *out[3], size_t out_len[3]);
void fqo_result_free(fqo_result *res);
const char *fqo_qual_range_to_enc(unsigned int min_qual, unsigned int max_qual);

#ifdef __cplusplus
}
#endif
#endif
