"""The replay of the reference's RL_Tree that the HIP chain kernel executes (fastq_utils_amd/csrc/fqg_rl_sim.h:
overwrite detection from arrival order, node-for-node replay of flagged (cell, gene) sets, earlier cells' arrays
rebuilt from their sorted members) run on the CPU with a one-lane wavefront (tests/cxx/rl_sim_check.cpp) and
compared, record by record, with the oracle's restatement of src/range_list.c (oracle/rl_oracle.c, itself
checked against the reference's own range_list.c in test_oracle_rl_matches_reference_source)."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import loader
from tests.util import REPO

SRC = os.path.join(REPO, "tests", "cxx", "rl_sim_check.cpp")


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rl") / "librl_sim_check.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, SRC], check=True)
    L = C.CDLL(so)
    L.rl_sim_check.argtypes = [C.c_uint32] + [C.c_void_p] * 3 + [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

    def run(chain, epoch, umi, cap=16384, mcap=4096, history=1, from_overwrite=1, by_cell=0):
        chain, epoch, umi = (np.ascontiguousarray(a, dtype=np.uint32) for a in (chain, epoch, umi))
        out = np.zeros(chain.size, dtype=np.uint8)
        st = np.zeros(8, dtype=np.uint64)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        L.rl_sim_check(chain.size, P(chain), P(epoch), P(umi), cap, mcap, history, from_overwrite, by_cell, P(out), P(st))
        return out, dict(zip(("undefined", "overwrites", "wild_writes", "overflow", "changed", "flagged", "lookback"),
                             (int(x) for x in st)))
    return run


def synth(rng, n_cells, n_genes, n, space, blocks):
    cell = np.sort(rng.integers(0, n_cells, n)).astype(np.uint32) + 1
    gene = (rng.zipf(1.5, n) % n_genes).astype(np.uint32) + 1
    if blocks:  # concentrate the UMIs in a few 64-blocks: many overwrites
        umi = (rng.integers(0, blocks, n) * 64 + rng.integers(0, 64, n)) % 1048576 + 1
    else:
        umi = rng.integers(0, space, n) + 1
    return cell, gene, umi.astype(np.uint32)


@pytest.mark.parametrize("seed", range(60))
def test_sorted_mode_chains(sim, seed):
    rng = np.random.default_rng(seed)
    n_cells, n_genes, n = int(rng.integers(1, 30)), int(rng.integers(1, 12)), int(rng.integers(10, 3000))
    cell, gene, umi = synth(rng, n_cells, n_genes, n, int(rng.choice([256, 4096, 1048576])),
                            int(rng.choice([0, 0, 3, 20, 200])))
    want, ost = loader.rl_replay(gene, umi, cell, np.ones(n, np.float32), n_genes + 1)
    got, st = sim(gene, cell, umi)
    assert st["overflow"] == 0 and (want == got).all()
    assert st["overwrites"] == ost[2]  # every overwrite of the reference happens inside a flagged set
    got, st = sim(gene, cell, umi, from_overwrite=0)  # replaying flagged sets from their first record: the same
    assert st["overflow"] == 0 and (want == got).all()
    got, st = sim(gene, cell, umi, by_cell=1)  # runs = (cell, gene) pairs with stored members (the sorted-mode path)
    assert st["overflow"] == 0 and (want == got).all()


@pytest.mark.parametrize("seed", range(30))
def test_unsorted_mode_one_tree_per_pair(sim, seed):
    rng = np.random.default_rng(1000 + seed)
    n_cells, n_genes, n = int(rng.integers(1, 10)), int(rng.integers(1, 8)), int(rng.integers(10, 4000))
    cell, gene, umi = synth(rng, n_cells, n_genes, n, int(rng.choice([256, 4096, 1048576])), int(rng.choice([0, 3, 20, 200])))
    p = rng.permutation(n)
    cell, gene, umi = cell[p], gene[p], umi[p]
    _, pid = np.unique(cell.astype(np.int64) * (n_genes + 1) + gene, return_inverse=True)
    pid = (pid + 1).astype(np.uint32)
    zero = np.zeros(n, np.uint32)
    want, _ = loader.rl_replay(pid, umi, zero, np.ones(n, np.float32), int(pid.max()) + 1)
    got, st = sim(pid, zero, umi, history=0)
    assert st["overflow"] == 0 and (want == got).all()


@pytest.mark.parametrize("seed", range(6))
def test_large_sets_saturate_the_node_counts(sim, seed):
    rng = np.random.default_rng(2000 + seed)
    n, n_cells, n_genes = int(rng.integers(2000, 20000)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
    cell = np.sort(rng.integers(0, n_cells, n)).astype(np.uint32) + 1
    gene = rng.integers(0, n_genes, n).astype(np.uint32) + 1
    umi = (rng.integers(0, int(rng.choice([2000, 20000, 1048576])), n) + 1).astype(np.uint32)
    want, _ = loader.rl_replay(gene, umi, cell, np.ones(n, np.float32), n_genes + 1)
    got, st = sim(gene, cell, umi, cap=1 << 16, mcap=1 << 15)
    assert st["overflow"] == 0 and (want == got).all()


def test_limits_are_reported_not_hidden(sim):
    rng = np.random.default_rng(7)
    n = 4000
    cell = np.ones(n, np.uint32)
    gene = np.ones(n, np.uint32)
    umi = (rng.integers(0, 1048576 - 64, n) + 1).astype(np.uint32)   # ~30 000 nodes: more than cap below
    top = 1048576 - 64
    umi[-3:] = [top + 41, top + 2, top + 42]  # the last block: leaf 2, then leaf 0 in front of it -> flagged
    _, st = sim(gene, cell, umi, cap=1024, mcap=64)
    assert st["flagged"] >= 1 and st["overflow"] == 1


def test_oracle_rl_matches_reference_source():
    """oracle/rl_oracle.c against the reference's own range_list.c (oracle/_ref/librange_list_ref.so, built by
    oracle/Makefile from /root/reference/src): same answers and the same node array after every operation,
    until the reference reads memory it never wrote (where its behaviour is not defined by its input)."""
    so = os.path.join(REPO, "oracle", "_ref", "librange_list_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/librange_list_ref.so not built")
    ref = C.CDLL(so)
    ref.new_rl.restype = C.c_void_p
    ref.new_rl.argtypes = [C.c_ulong]
    ref.set_in_rl.argtypes = [C.c_void_p, C.c_ulong, C.c_int]
    ref.set_in_rl.restype = C.c_void_p
    ref.in_rl.argtypes = [C.c_void_p, C.c_ulong]
    ref.in_rl.restype = C.c_short
    ref.rl_all.argtypes = [C.c_void_p, C.c_int]

    class RL(C.Structure):
        _fields_ = [("root", C.POINTER(C.c_uint16)), ("size", C.c_ulong), ("mem", C.c_ulong), ("max", C.c_ulong),
                    ("root_i", C.c_ulong)]
    compared = overwrites = 0
    for seed in range(400):
        rng = random.Random(seed)
        a, b = ref.new_rl(1048576), loader.RLTree(1048576)
        ra = C.cast(a, C.POINTER(RL)).contents
        nvals, span = rng.choice([5, 20, 100, 1000]), rng.choice([64, 1000, 100000, 1048576])
        base = rng.randrange(1, 1048576 - span + 2)
        stop = False
        for _ in range(rng.randrange(1, 6)):
            for _ in range(nvals):
                v = base + rng.randrange(span)
                x, y = bool(ref.in_rl(a, v)), v in b
                if b.undefined_reads:
                    stop = True
                    break
                assert x == y
                if not x:
                    ref.set_in_rl(a, v, 1)
                    b.insert(v)
                if b.undefined_reads:
                    stop = True
                    break
                assert ra.size == b.size
                assert b.first_difference(ra.root) == -1
                compared += 1
            if stop:
                break
            ref.rl_all(a, 0)
            b.all_out()
        overwrites += b.overwrites
    assert compared > 20000 and overwrites > 100


def test_compat_range_list_matches_reference_source():
    """The range_list.h API exported by libfastq_gpu.so (fastq_utils_amd/compat/range_list_compat.cpp) against the
    reference's own range_list.c: same answers, sizes and node arrays for ranges of every shape, IN and OUT, rl_all,
    rl_next_in_bigger.  Sequences that insert out of order are followed while the oracle says the reference has not
    read memory it never wrote (two heaps hold different garbage there).  Needs no GPU."""
    so_ref = os.path.join(REPO, "oracle", "_ref", "librange_list_ref.so")
    so_mine = os.path.join(REPO, "fastq_utils_amd", "libfastq_gpu.so")
    if not (os.path.exists(so_ref) and os.path.exists(so_mine)):
        pytest.skip("needs oracle/_ref/librange_list_ref.so and libfastq_gpu.so")

    class RL(C.Structure):
        _fields_ = [("root", C.POINTER(C.c_uint16)), ("size", C.c_ulong), ("mem", C.c_ulong), ("max", C.c_ulong),
                    ("root_i", C.c_ulong)]
    libs = []
    for path in (so_ref, so_mine):
        L = C.CDLL(path)
        L.new_rl.restype = C.POINTER(RL)
        L.new_rl.argtypes = [C.c_ulong]
        L.set_in_rl.argtypes = [C.POINTER(RL), C.c_ulong, C.c_int]
        L.set_in_rl.restype = C.POINTER(RL)
        L.in_rl.argtypes = [C.POINTER(RL), C.c_ulong]
        L.in_rl.restype = C.c_short
        L.rl_all.argtypes = [C.POINTER(RL), C.c_int]
        L.rl_next_in_bigger.argtypes = [C.POINTER(RL), C.c_ulong]
        L.rl_next_in_bigger.restype = C.c_ulong
        libs.append(L)
    ref, mine = libs
    ops = 0
    for seed in range(600):
        rng = random.Random(seed)
        mx = rng.choice([2, 17, 64, 100, 1000, 4096, 65536, 100000, 1048576])
        a, b = ref.new_rl(mx), mine.new_rl(mx)
        assert (a.contents.size, a.contents.max, a.contents.root_i, a.contents.root[0]) == (
            b.contents.size, b.contents.max, b.contents.root_i, b.contents.root[0])
        span = rng.choice([20, mx])
        base = rng.randrange(1, max(2, mx - span + 2))
        top = min(mx, base + span)
        vals = sorted(rng.sample(range(base, top + 1), min(top - base + 1, rng.choice([3, 30, 300]))))
        monotone = rng.random() < 0.5
        guide = None
        if not monotone:
            rng.shuffle(vals)
            guide = loader.RLTree(mx)
        for v in vals:
            st = 1 if (guide is not None or rng.random() < 0.85) else 0
            if guide is not None:
                if v not in guide:
                    guide.insert(v)
                if guide.undefined_reads:
                    break
            assert ref.in_rl(a, v) == mine.in_rl(b, v)
            ref.set_in_rl(a, v, st)
            mine.set_in_rl(b, v, st)
            assert a.contents.size == b.contents.size
            if guide is None:
                assert all(a.contents.root[i] == b.contents.root[i] for i in range(a.contents.size))
                q = rng.randrange(0, mx + 2)
                assert ref.rl_next_in_bigger(a, q) == mine.rl_next_in_bigger(b, q)
            ops += 1
        if guide is None:
            for st in (1, 0):
                ref.rl_all(a, st)
                mine.rl_all(b, st)
                for v in rng.sample(range(1, mx + 1), min(mx, 20)):
                    assert ref.in_rl(a, v) == mine.in_rl(b, v)
    assert ops > 15000
