"""Pin the fastq_pre_barcodes oracle (oracle/pre_barcodes_oracle.py) on the golden vectors
captured from the reference binary (tests/golden/pre_barcodes.json, tools/gen_golden.py)."""
import json
import os

import pytest

from oracle import pre_barcodes_oracle as pbo
from tests.util import GOLD, read_image, strip_progress

GOLDEN = json.load(open(os.path.join(GOLD, "pre_barcodes.json")))


def reader(name):
    return read_image(os.path.join(GOLD, name))


def real_args(args):
    return [a.replace("OUT1", "SCRATCH/o1.fastq.gz").replace("OUT2", "SCRATCH/o2.fastq.gz") for a in args]


@pytest.mark.parametrize("case", GOLDEN, ids=[str(i) + ":" + " ".join(c["args"])[:70] for i, c in enumerate(GOLDEN)])
def test_oracle_matches_reference_binary(case):
    got = pbo.run_pre_barcodes(real_args(case["args"]), reader)
    assert got["exit"] == case["exit"]
    if "--help" in case["args"]:
        return
    assert got["stdout"] == case["stdout"]
    assert strip_progress(got["stderr"]) == strip_progress(case["stderr"])
    if case["exit"] == 0:
        for tag, idx in (("OUT1", 1), ("OUT2", 2)):
            if tag in case["files"]:
                assert got["files"][idx].decode("latin-1") == case["files"][tag]


def test_reference_suite_known_answers():
    """run_tests.sh:388-395: the three byte-exact goldens pre1/pre2/pre3.fastq.gz"""
    for k, name in ((2, "pre1"), (3, "pre2"), (4, "pre3")):
        want = read_image(os.path.join(GOLD, "data", name + ".fastq.gz")).decode("latin-1")
        assert GOLDEN[k]["files"]["OUT1"] == want
