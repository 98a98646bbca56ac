"""GPU parity tests for bin/fastq_filterpair (reference src/fastq_filterpair.c:38-228): every golden
invocation of the reference binary (tests/golden/filterpair.json, tools/gen_golden.py filterpair) - the reference
suite's own cases (run_tests.sh:361-370), every _1/_2 fixture pair in both orders and both modes, seeded files with
mates in different orders, singletons on both sides, a name asked for twice, truncated / malformed second files.
Compared: exit status, stderr, and the DECOMPRESSED bytes of the three outputs."""
import gzip
import hashlib
import json
import os
import subprocess
import tempfile

import pytest

from tests.util import GOLD, REPO, SideBySide

pytestmark = pytest.mark.gpu
BIN = os.path.join(REPO, "bin", "fastq_filterpair")
GOLDEN = json.load(open(os.path.join(GOLD, "filterpair.json")))


def run_case(case, env=None):
    with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
        rel = os.path.relpath(tmp, GOLD)
        real = list(case["args"])
        if len(real) in (2, 3):
            real = real[:2] + [rel + "/p1.fastq.gz", rel + "/p2.fastq.gz", rel + "/up.fastq.gz"] + real[2:]
        real = [rel + "/" + a if a in ("O1", "O2") else a for a in real]
        p = subprocess.run(["fastq_filterpair"] + real, executable=BIN, cwd=GOLD, capture_output=True, timeout=300,
                           env=dict(os.environ, **(env or {})))
        files = {}
        if p.returncode == 0:
            for k in ("p1", "p2", "up"):
                path = os.path.join(tmp, k + ".fastq.gz")
                if os.path.exists(path):
                    files[k] = gzip.decompress(open(path, "rb").read())
        return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1").replace(rel + "/", "SCRATCH/"), files


# (the programs of all cases start side by side the first time one is asked for: tests/util.py)
GOLDEN_RUNS = SideBySide(lambda i: run_case(GOLDEN[i]), range(len(GOLDEN)))


@pytest.mark.parametrize("i", range(len(GOLDEN)), ids=[" ".join(c["args"])[:90] or "no-args" for c in GOLDEN])
def test_cli_matches_reference_binary(i):
    case = GOLDEN[i]
    rc, out, err, files = GOLDEN_RUNS.get(i)
    assert rc == case["exit"], err[-400:]
    assert out == case["stdout"]
    assert err == case["stderr"]
    assert set(files) == set(case["files"])
    for k, want in case["files"].items():
        assert len(files[k]) == want["len"] and hashlib.sha256(files[k]).hexdigest() == want["sha256"], k
