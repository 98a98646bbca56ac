"""The exact resolution of the runs that fingerprints cannot classify (fastq_utils_amd.dist.resolve_pair_runs)
and the merge of per-owner results, against a literal serial loop (reference src/fastq_info.c:333-356:
look the name up, unpaired if absent, delete).  CPU only."""
import numpy as np

from fastq_utils_amd import dist as fdist

F2 = fdist.FP_FILE2


def serial(names1, names2):
    index = set(names1)
    first, unpaired, matched = None, 0, 0
    for r, nm in enumerate(names2):
        if nm in index:
            index.discard(nm)
            matched += 1
        else:
            unpaired += 1
            if first is None:
                first = r
    return matched, len(index), unpaired, first


def as_runs(names1, names2, n_buckets, rng):
    """group entries into 'runs' the way colliding fingerprints would: by a weak hash of the name"""
    entries = []
    salt = int(rng.integers(1, 1 << 30))
    for i, nm in enumerate(names1):
        entries.append(((hash((nm, salt)) % n_buckets), i))
    for i, nm in enumerate(names2):
        entries.append(((hash((nm, salt)) % n_buckets), i | F2))
    return entries


def test_resolution_equals_the_serial_loop_under_heavy_collisions():
    rng = np.random.default_rng(3)
    for trial in range(200):
        n1 = int(rng.integers(0, 40))
        names1 = [b"n%d" % i for i in rng.permutation(60)[:n1]]
        names2 = [b"n%d" % int(x) for x in rng.integers(0, 70, int(rng.integers(0, 50)))]
        entries = as_runs(names1, names2, int(rng.integers(1, 12)), rng)
        table = {i: nm for i, nm in enumerate(names1)}
        table.update({i | F2: nm for i, nm in enumerate(names2)})
        assert fdist.resolve_pair_runs(entries, table.__getitem__) == serial(names1, names2), trial


def test_merge_over_owners():
    parts = [(5, 1, 0, None), (7, 0, 2, 90), (0, 3, 1, 17), (1, 0, 0, None)]
    assert fdist.merge_pairing(parts) == (13, 4, 3, 17)
    assert fdist.merge_pairing([(2, 0, 0, None)]) == (2, 0, 0, None)
