"""dist.umi_count_sharded with more than one rank (its rounds: the file's numbering, the file's UMI numbers, the
history of replayed features, the float32 chain of totals): two and three processes over gloo, all on GPU 0 (RCCL
refuses several ranks on one device), against the single-GPU result of the whole file - which tests/test_gpu_umi.py
pins on the oracle and the reference binary."""
import json
import os
import sys
import tempfile

import numpy as np
import pytest

import fastq_utils_amd as fq
from tests import bamgen
from tests.test_gpu_umi import split_at_cell_boundaries
from tests.util import REPO, free_port, run_group

pytestmark = pytest.mark.gpu


def run_ranks(shards):
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "shards.npz"), **{"shard%d" % k: np.frombuffer(s, dtype=np.uint8) for k, s in enumerate(shards)})
        out = os.path.join(tmp, "out.json")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(len(shards)),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(REPO, "tests", "umi_shard_worker.py"),
               os.path.join(tmp, "shards.npz"), out]
        p = run_group(cmd, 600, cwd=REPO)
        assert p.returncode == 0, p.stderr.decode("latin-1")[-3000:]
        return json.load(open(out))


@pytest.mark.parametrize("ranks", [2, 3])
def test_replayed_sets_across_ranks(ranks):
    rng = np.random.default_rng(31 + ranks)
    n = 30000
    n_cells, n_genes = 40, 9
    cell = np.sort(rng.integers(0, n_cells, n))
    gene = rng.zipf(1.5, n) % n_genes
    umi = (rng.integers(0, 300, n) * 7) % (4 ** 10)
    cells_code = rng.choice(np.uint64(1) << np.uint64(32), size=n_cells, replace=False).astype(np.uint64)
    rec = bamgen.fixed_records(cells_code[cell], gene, umi.astype(np.uint64))
    hdr = bamgen.header()
    with fq.Context(0) as ctx:
        whole = ctx.umi_count(hdr + rec.tobytes())
    assert whole["code"] == 0 and whole["rl_replayed"] >= 1
    starts = np.nonzero(np.diff(cell, prepend=-1))[0]
    cuts = [int(starts[(len(starts) * k) // ranks]) for k in range(ranks)] + [n]
    got = run_ranks([hdr + rec[cuts[k]:cuts[k + 1]].tobytes() for k in range(ranks)])
    assert [tuple(e) for e in got["entries_u"]] == whole["entries"][0]
    assert [tuple(e) for e in got["entries_r"]] == whole["entries"][1]
    assert (got["n_entries"], got["total"], got["tot_reads"], got["tot_umi"], got["rl_undefined"]) == (
        whole["n_entries"], whole["total"], whole["tot_reads"], whole["tot_umi"], whole["rl_undefined"])


def test_fractional_increments_across_ranks():
    rng = np.random.default_rng(77)
    bam, stream = bamgen.tagged_bam(rng, n_cells=24, genes=20, reads_per_cell=(20, 200), umi_len=4, nh=True, multi_gx=True)
    with fq.Context(0) as ctx:
        whole = ctx.umi_count(stream)
    assert whole["code"] == 0 and not whole["unit_increments"]
    got = run_ranks(split_at_cell_boundaries(stream, 3))
    assert [tuple(e) for e in got["entries_u"]] == whole["entries"][0]
    assert [tuple(e) for e in got["entries_r"]] == whole["entries"][1]
    assert (got["tot_reads"], got["tot_umi"]) == (whole["tot_reads"], whole["tot_umi"])
