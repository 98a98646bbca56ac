"""Strengthen the oracle pins of the other programs: on the seeded, lightly damaged inputs that the differential campaign
(tools/fuzz_campaign_programs.py) feeds to the GPU programs, the Python restatement of fastq_pre_barcodes
(oracle/pre_barcodes_oracle.py) must print and write exactly what the reference binary does.  CPU only; skipped when
oracle/_ref is absent (it needs /root/reference to be built)."""
import gzip
import importlib
import os
import subprocess
import sys

import pytest

from oracle import pre_barcodes_oracle as pbo
from tests.util import REPO, strip_progress

REF = os.path.join(REPO, "oracle", "_ref", "fastq_pre_barcodes")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")


def campaign_case(seed):
    """the fastq_pre_barcodes case of a campaign seed: (args, files, mutations), or None when the seed has none"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    fc = importlib.import_module("fuzz_campaign_programs")
    got = {}

    def capture(name, args, files, outs, envs, seed_, what):
        if name == "fastq_pre_barcodes":
            got["case"] = (args, files, what)
        return []

    saved = fc.compare
    fc.compare = capture
    try:
        fc.one_case(seed)
    finally:
        fc.compare = saved
    return got.get("case")


@pytest.mark.parametrize("seed", [930035] + list(range(971000, 971060)))
def test_pre_barcodes_oracle_against_the_reference_binary(seed, tmp_path):
    case = campaign_case(seed)
    if case is None:
        pytest.skip("no fastq_pre_barcodes case for this seed")
    args, files, what = case
    if sum(len(v) for v in files.values()) > 400_000:
        pytest.skip("a large case (the restatement is a Python loop)")
    for name, img in files.items():
        (tmp_path / name).write_bytes(img)
    p = subprocess.run(["fastq_pre_barcodes"] + args, executable=REF, cwd=tmp_path, capture_output=True, timeout=120)
    if p.returncode < 0:
        pytest.skip("the reference dies of a signal on this input")
    want = pbo.run_pre_barcodes(args, lambda n: files[n])
    assert want["exit"] == p.returncode, (what, p.stderr.decode("latin-1")[-300:], want["stderr"][-300:])
    assert want["stdout"] == p.stdout.decode("latin-1"), what
    assert strip_progress(want["stderr"]) == strip_progress(p.stderr.decode("latin-1")), what
    out = tmp_path / "o.fastq.gz"
    if p.returncode == 0 and "o.fastq.gz" in args:
        raw = out.read_bytes()
        assert (gzip.decompress(raw) if raw else b"") == want["files"][1], what
