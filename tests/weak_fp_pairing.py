"""Run by tests/test_gpu_dist_pairing.py::test_names_decide_where_fingerprints_collide with FQGPU_FP_WEAK_BITS=10 in the
environment (the library reads it once per process): pairing of two files with reads without a mate on both sides, over
1 and 3 virtual ranks, with the names beside the pairs and without, against the oracle's serial file-2 loop."""
import sys

import fastq_utils_amd as fq
from tests.test_gpu_dist_pairing import oracle_pairing, orphans_on_both_sides, virtual_pairing


def main():
    ctx = fq.Context(0)
    ok = {True: True, False: True}
    for n_shards in (1, 3):
        for seed in (1, 2):
            f1, f2 = orphans_on_both_sides(seed, n=4000)
            want = oracle_pairing(f1, f2)
            for named in (True, False):
                got = virtual_pairing(ctx, f1, f2, n_shards, named=named)
                # the first read of file 2 without a mate is the serial loop's finding; the pairs are the names both files hold
                truth_matched = len(set(names(f1)) & set(names(f2)))
                same = got[3] == want[0] and got[0] == truth_matched
                ok[named] = ok[named] and same
                print(n_shards, seed, "named" if named else "plain", got, want, truth_matched, flush=True)
    print("names_travel: " + ("as the serial loop" if ok[True] else "NOT as the serial loop"))
    print("fingerprints_alone: " + ("as the serial loop" if ok[False] else "NOT as the serial loop"))
    ctx.close()
    return 0 if ok[True] else 1


def names(img):
    return [ln.split(b" ")[0] for ln in img.split(b"\n")[0::4] if ln]


if __name__ == "__main__":
    sys.exit(main())
