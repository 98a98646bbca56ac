"""The Python restatement of fastq_filter_n / fastq_trim_poly_at (oracle/filter_oracle.py) against
the golden invocations of the reference binaries (tests/golden/filters.json): exit status, stdout,
stderr (progress ticker stripped), and the decompressed output file.  CPU only."""
import hashlib
import json
import os

import pytest

from oracle import filter_oracle as fo
from tests.util import GOLD, strip_progress

GOLDEN = json.load(open(os.path.join(GOLD, "filters.json")))


def opener(path):
    return open(os.path.join(GOLD, path), "rb").read()


def same_text(packed, got):
    assert packed["len"] == len(got)
    assert packed["sha256"] == hashlib.sha256(got).hexdigest()
    if packed["text"] is not None:
        assert packed["text"] == got.decode("latin-1")


def ids(cases):
    return [str(i) + ":" + " ".join(c["args"])[-60:] for i, c in enumerate(cases)]


@pytest.mark.parametrize("case", GOLDEN["filter_n"], ids=ids(GOLDEN["filter_n"]))
def test_filter_n(case):
    got = fo.filter_n(case["args"], opener)
    assert got["exit"] == case["exit"], got["stderr"]
    same_text(case["stdout"], got["stdout"])
    assert got["stderr"].decode("latin-1") == strip_progress(case["stderr"])


@pytest.mark.parametrize("case", GOLDEN["trim_poly_at"], ids=ids(GOLDEN["trim_poly_at"]))
def test_trim_poly_at(case):
    got = fo.trim_poly_at(case["args"], opener, can_write=lambda p: not p.startswith("/xxx/"))
    assert got["exit"] == case["exit"], got["stderr"]
    assert got["stdout"].decode("latin-1") == case["stdout"]
    assert got["stderr"].decode("latin-1") == strip_progress(case["stderr"])
    if case["out"] is not None:
        same_text(case["out"], got["out"])


def test_reference_suite_golden_poly_at_len3():
    """run_tests.sh:199: --min_poly_at_len 3 on poly_at.fastq.gz gives poly_at_len3.fastq.gz"""
    got = fo.trim_poly_at(["--file", "data/poly_at.fastq.gz", "--outfile", "x", "--min_poly_at_len", "3"], opener)
    assert got["out"] == fo.read_input("data/poly_at_len3.fastq.gz", opener)


def test_trim_record_quality_length_quirks():
    # 3' cut lands inside / at the end of / behind a quality string of another length (:91-96)
    assert fo.trim_record(b"ACGTAAAA\n", b"IIIIIIII\n", 3) == (b"ACGT\n", b"IIII\n", 5, True)
    assert fo.trim_record(b"ACGTAAAA\n", b"IIII", 3) == (b"ACGT\n", b"IIII\n", 5, True)
    assert fo.trim_record(b"ACGTAAAA\n", b"II\n", 3) == (b"ACGT\n", b"II\n", 5, True)
    # 5' shift copies read_len - matched + 1 characters; a longer quality keeps its unshifted end (:107-113)
    assert fo.trim_record(b"TTTTACGT\n", b"12345678\n", 3) == (b"ACGT\n", b"5678\n", 5, True)
    assert fo.trim_record(b"TTTTACGT\n", b"123456789ab\n", 3) == (b"ACGT\n", b"56789a789ab\n", 5, True)
