/* TEST DRIVER: the C restatements under oracle/ (fq_oracle.c, rl_oracle.c) compiled together with this file under
 * -fsanitize=address,undefined (tests/test_sanitizers.py).  argv: mode file1 [file2]; files are DECOMPRESSED images.
 *   info  f1 [f2]   fastq_info in every flag combination the goldens use
 *   pair  f1 f2     fastq_filterpair, both modes
 *   rl              RL_Tree replay of a seeded stream with heavy UMI re-use */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/fq_oracle.h"

int orl_replay(uint64_t n, const uint32_t *tree_of, const uint32_t *umi, const uint32_t *epoch, const float *incr,
               uint32_t n_trees, uint8_t *is_new, uint64_t *stats);

static unsigned char *slurp(const char *path, size_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) {
    perror(path);
    exit(9);
  }
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *b = (unsigned char *)malloc(*n + 1);
  if (fread(b, 1, *n, f) != *n) exit(9);
  fclose(f);
  return b;
}

int main(int argc, char **argv) {
  if (argc < 2) return 9;
  if (!strcmp(argv[1], "rl")) {
    enum { N = 200000 };
    uint32_t *t = malloc(N * 4), *u = malloc(N * 4), *e = malloc(N * 4);
    float *inc = malloc(N * 4);
    uint8_t *nw = malloc(N);
    uint64_t st[3] = {0, 0, 0}, x = 88172645463325252ull;
    for (int i = 0; i < N; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      e[i] = 1 + (uint32_t)(i / 2000);
      t[i] = 1 + (uint32_t)(x % 7);
      u[i] = 1 + (uint32_t)((x >> 20) % ((i & 1) ? 300 : 1048576));
      inc[i] = 1.0f;
    }
    orl_replay(N, t, u, e, inc, 8, nw, st);
    printf("rl %llu %llu %llu\n", (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2]);
    free(t); free(u); free(e); free(inc); free(nw);
    return 0;
  }
  size_t n1 = 0, n2 = 0;
  unsigned char *b1 = slurp(argv[2], &n1), *b2 = argc > 3 ? slurp(argv[3], &n2) : NULL;
  if (!strcmp(argv[1], "info")) {
    const int flagsets[] = {0, FQO_FLAG_R, FQO_FLAG_Q, FQO_FLAG_E, FQO_FLAG_R | FQO_FLAG_S};
    for (unsigned k = 0; k < sizeof(flagsets) / sizeof(flagsets[0]); ++k)
      for (int kind = 0; kind < 3; ++kind) {
        if (kind == FQO_ARG2_FILE && !b2) continue;
        fqo_job job = {b1, n1, argv[2], kind == FQO_ARG2_FILE ? b2 : NULL, kind == FQO_ARG2_FILE ? n2 : 0,
                       kind == FQO_ARG2_FILE ? argv[3] : NULL, kind, flagsets[k]};
        fqo_result res;
        fqo_fastq_info(&job, &res);
        fqo_result_free(&res);
      }
  } else if (!strcmp(argv[1], "pair") && b2) {
    for (int sorted = 0; sorted < 2; ++sorted) {
      fqo_job job = {b1, n1, argv[2], b2, n2, argv[3], FQO_ARG2_FILE, sorted ? FQO_FLAG_S : 0};
      fqo_result res;
      char *out[3];
      size_t len[3];
      fqo_fastq_filterpair(&job, &res, out, len);
      fqo_result_free(&res);
      for (int i = 0; i < 3; ++i) free(out[i]);
    }
  }
  free(b1);
  free(b2);
  return 0;
}
