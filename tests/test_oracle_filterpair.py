"""Pin the fastq_filterpair restatement (oracle/fq_oracle.c: fqo_fastq_filterpair, following
src/fastq_filterpair.c:38-228 and src/fastq.c:77-80,124-157) on the golden vectors captured from the reference
binary (tests/golden/filterpair.json, tools/gen_golden.py filterpair): the reference suite's own invocations
(run_tests.sh:361-370), every _1/_2 fixture pair in both orders and both modes, seeded files with mates in
different orders, singletons on both sides, a name asked for twice, truncated / malformed second files."""
import hashlib
import json
import os

import pytest

from oracle import loader as orc
from tests.util import GOLD, read_image

GOLDEN = json.load(open(os.path.join(GOLD, "filterpair.json")))


def run_oracle(args):
    """the program's argv handling (src/fastq_filterpair.c:47-61) and fastq_open's failures around the restatement"""
    head = "fastq_utils 0.25.3\n"
    n = len(args)
    if n in (2, 3):
        n += 3  # the three output names the tests add
    if n + 1 not in (6, 7):
        return {"exit": 1, "stdout": "", "stderr": head + "Usage: filterpair fastq1 fastq2 paired1 paired2 unpaired [sorted]\n",
                "files": None}
    for k in (0, 1):
        if not os.path.exists(os.path.join(GOLD, args[k])):
            return {"exit": 1, "stdout": "", "stderr": head + "%d\nERROR: Unable to open %s\n" % (n + 1, args[k]), "files": None}
    b1 = read_image(os.path.join(GOLD, args[0]))
    b2 = read_image(os.path.join(GOLD, args[1]))
    return orc.fastq_filterpair(b1, args[0], b2, args[1], sorted_mode=(len(args) == 3 and args[2] == "sorted"))


@pytest.mark.parametrize("case", GOLDEN, ids=lambda c: " ".join(c["args"])[:90] or "no-args")
def test_oracle_matches_reference_binary(case):
    got = run_oracle(case["args"])
    assert got["exit"] == case["exit"]
    assert got["stdout"] == case["stdout"]
    assert got["stderr"] == case["stderr"]
    if case["exit"] == 0:
        for k, name in enumerate(("p1", "p2", "up")):
            want = case["files"][name]
            assert len(got["files"][k]) == want["len"] and hashlib.sha256(got["files"][k]).hexdigest() == want["sha256"], name


def test_reference_suite_known_answers():
    """run_tests.sh:361-362: the paired outputs of a file against itself / of a_1 + a_2 are the inputs"""
    by = {" ".join(c["args"]): c for c in GOLDEN}
    c = by["data/test_2.fastq.gz data/test_2.fastq.gz"]
    assert c["exit"] == 0 and c["files"]["p1"]["sha256"] == hashlib.sha256(read_image(os.path.join(GOLD, "data/test_2.fastq.gz"))).hexdigest()
    c = by["data/a_1.fastq.gz data/a_2.fastq.gz"]
    assert c["files"]["p1"]["sha256"] == hashlib.sha256(read_image(os.path.join(GOLD, "data/a_1.fastq.gz"))).hexdigest()
    assert c["files"]["p2"]["sha256"] == hashlib.sha256(read_image(os.path.join(GOLD, "data/a_2.fastq.gz"))).hexdigest()
    assert by["data/c18_10000_1.fastq.gz data/casava.1.8_2.fastq.gz"]["exit"] != 0      # must_fail, :368
