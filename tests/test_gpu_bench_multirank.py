"""bench.py's multi-rank control flow (barriers, max-over-ranks time, statistics merge, the cross-rank
unique-name and pairing extra, the exit without a collective tear-down) on a ONE-GPU box: two ranks on
GPU 0 over gloo (FQGPU_BENCH_ONE_DEVICE=1; RCCL refuses two ranks on one device).  The numbers mean
nothing here; the bench line must be the last line of stdout, complete, with exit status 0."""
import json
import os
import sys

import pytest

from tests.util import REPO, free_port, run_group

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_two_ranks_on_one_device(ranks):
    """(8: the shape of the driver's eight-GPU run - every rank's file-2 shard holds the mates of the next rank's file-1
    shard, BASELINE configs[4] - at a million reads a rank)"""
    env = dict(os.environ, FQGPU_BENCH_ONE_DEVICE="1")
    reads = 1000000 if ranks == 8 else 2000000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--reads", str(reads)]
    check_line(run_group(cmd, 900, cwd=REPO, env=env), ranks, reads)


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with no launcher around it (the shape of the driver's 1-GPU command): bench.py
    starts the ranks itself, as children, before it touches the GPU."""
    env = dict(os.environ, FQGPU_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "2000000"]
    check_line(run_group(cmd, 600, cwd=REPO, env=env), 2)


def check_line(p, ranks, reads=2000000):
    assert p.returncode == 0, p.stderr.decode("latin-1")[-2000:]
    lines = [ln for ln in p.stdout.decode("latin-1").splitlines() if ln.startswith("{")]
    # the measured line once before the extra and once, complete, as the LAST line; the extra as a line of its own
    assert len(lines) == 3 and all(len(ln) <= 4096 for ln in (lines[0], lines[-1]))
    first, d = json.loads(lines[0]), json.loads(lines[-1])
    assert first["final"] is False and d["final"] is True and first["value"] == d["value"]
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["value"] > 0
    assert d["steps"] == 2 and d["warmup"] == 1 and d["roofline"]["frac"] > 0
    assert d["dedup"]["finding"] is None and d["dedup"]["pairing_ok"] is True
    x = json.loads(lines[1])
    assert x["extra"] == "dedup_extra"
    assert "error" not in x, x
    assert x["names_total"] == ranks * reads and x["finding"] is None
    assert x["pairing"]["ok"] and x["pairing"]["matched"] == ranks * reads and x["pairing"]["pairs_total"] == ranks * reads
