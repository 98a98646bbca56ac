"""Strengthen the oracle pins of the other programs: on the seeded, lightly damaged inputs that the differential campaign
(tools/fuzz_campaign_programs.py) feeds to the GPU programs, the restatements of fastq_pre_barcodes
(oracle/pre_barcodes_oracle.py), fastq_filter_n / fastq_trim_poly_at (oracle/filter_oracle.py) and fastq_filterpair
(oracle/fq_oracle.c) must print and write exactly what the reference binaries do.  CPU only; skipped when oracle/_ref is
absent (it needs /root/reference to be built)."""
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


def campaign_case(seed, program="fastq_pre_barcodes"):
    """the case of a campaign seed for one program: (args, files, mutations), or None when the seed has none"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    fc = importlib.import_module("fuzz_campaign_programs")
    got = {}

    def capture(name, args, files, outs, envs, seed_, what):
        if name == program:
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


def _ref(program, args, files, tmp_path):
    for name, img in files.items():
        (tmp_path / name).write_bytes(img)
    return subprocess.run([program] + args, executable=os.path.join(REPO, "oracle", "_ref", program), cwd=tmp_path,
                          capture_output=True, timeout=120)


def _gunzip(path):
    raw = path.read_bytes() if path.exists() else b""
    return gzip.decompress(raw) if raw else b""


@pytest.mark.parametrize("seed", range(972000, 972040))
def test_filter_oracles_against_the_reference_binaries(seed, tmp_path):
    from oracle import filter_oracle as fo
    ran = 0
    for program in ("fastq_filter_n", "fastq_trim_poly_at"):
        case = campaign_case(seed, program)
        if case is None or sum(len(v) for v in case[1].values()) > 400_000:
            continue
        args, files, what = case
        p = _ref(program, args, files, tmp_path)
        if p.returncode < 0:
            continue
        if program == "fastq_filter_n":
            got = fo.filter_n(args, lambda n: files[n])
        else:
            got = fo.trim_poly_at(args, lambda n: files[n])
        assert got["exit"] == p.returncode, (program, what, p.stderr[-200:], got["stderr"][-200:])
        assert got["stdout"] == p.stdout, (program, what)
        assert got["stderr"].decode("latin-1") == strip_progress(p.stderr.decode("latin-1")), (program, what)
        if program == "fastq_trim_poly_at" and p.returncode == 0:
            assert _gunzip(tmp_path / "o.fastq.gz") == (got["out"] or b""), what
        ran += 1
    if not ran:
        pytest.skip("large cases only")


@pytest.mark.parametrize("seed", range(973000, 973040))
def test_filterpair_oracle_against_the_reference_binary(seed, tmp_path):
    from oracle import loader as orc
    case = campaign_case(seed, "fastq_filterpair")
    if case is None or sum(len(v) for v in case[1].values()) > 400_000:
        pytest.skip("a large case")
    args, files, what = case
    p = _ref("fastq_filterpair", args, files, tmp_path)
    if p.returncode < 0:
        pytest.skip("the reference dies of a signal on this input")
    got = orc.fastq_filterpair(files["a.fastq"], "a.fastq", files["b.fastq"], "b.fastq", sorted_mode=args[-1] == "sorted")
    assert got["exit"] == p.returncode, (what, p.stderr[-200:], got["stderr"][-200:])
    assert got["stdout"] == p.stdout.decode("latin-1"), what
    assert strip_progress(got["stderr"]) == strip_progress(p.stderr.decode("latin-1")), what
    if p.returncode == 0:
        for k, name in enumerate(("p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz")):
            assert _gunzip(tmp_path / name) == got["files"][k], (what, name)
