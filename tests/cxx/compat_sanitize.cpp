// TEST DRIVER: the container code of libfastq_gpu.so that needs no GPU - range_list_compat.cpp (src/range_list.h API) -
// compiled together with this file under -fsanitize=address,undefined (tests/test_sanitizers.py): trees of many
// ranges, inserts in and out of order, OUT, rl_all, copies, freeze, display.
#include <stdint.h>
#include <stdio.h>

#include "../../include/fastq_gpu_compat.h"

int main() {
  uint64_t x = 0x9E3779B97F4A7C15ull;
  const unsigned long ranges[] = {2, 17, 64, 100, 1000, 4096, 65536, 100000, 1048576};
  unsigned long members = 0;
  for (unsigned long mx : ranges)
    for (int round = 0; round < 40; ++round) {
      RL_Tree* t = new_rl(mx);
      for (int epoch = 0; epoch < 3; ++epoch) {
        for (int i = 0; i < 400; ++i) {
          x ^= x << 13; x ^= x >> 7; x ^= x << 17;
          const unsigned long v = 1 + (unsigned long)(x % mx);
          if (!in_rl(t, v)) set_in_rl(t, v, (x >> 40) % 9 ? IN : OUT);
          members += (unsigned long)in_rl(t, v);
          (void)rl_next_in_bigger(t, (unsigned long)((x >> 33) % (mx + 1)));
        }
        RL_Tree* c = copy_rl(t);
        freeze_rl(c);
        members += (unsigned long)in_rl(c, 1);
        free_rl(c);
        rl_all(t, epoch == 1 ? IN : OUT);
      }
      if (round == 0 && mx <= 100) display_tree(t);
      free_rl(t);
    }
  printf("members seen: %lu\n", members);
  return 0;
}
