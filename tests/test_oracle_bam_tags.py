"""The bam_add_tags oracle (oracle/bam_tags_oracle.py) against the reference program: every golden invocation of
tests/golden/bam_tags.json (oracle/_ref/bam_add_tags = src/bam_add_tags.c compiled from the reference's sources, run
by tools/gen_golden.py on the reference's own trans_small.bam / mapTrans2Gene.tsv - run_tests.sh:485-499 - and on a
seeded BAM whose read names leave get_barcodes at each of its exits) must give the same exit status, stderr and
inflated output BAM."""
import hashlib
import json
import os

import pytest

from oracle import bam_tags_oracle as bto
from tests.util import GOLD

GOLDEN = json.load(open(os.path.join(GOLD, "bam_tags.json")))


def reader(path):
    full = os.path.join(GOLD, path)
    return open(full, "rb").read() if os.path.exists(full) else None


def writable(path):
    return os.path.isdir(os.path.dirname(os.path.join(GOLD, path)) or GOLD)


@pytest.mark.parametrize("case", GOLDEN, ids=lambda c: " ".join(c["args"])[-70:] or "no arguments")
def test_golden(case):
    args = ["o.bam" if a == "OUT" else a for a in case["args"]]
    got = bto.run_bam_add_tags(args, reader, writable)
    assert got["exit"] == case["exit"]
    assert got["stderr"] == case["stderr"].replace("SCRATCH/", "")
    if "out_sha256" in case:
        data = got["stdout"] if case["stdout_is_bam"] else got["files"]["o.bam"]
        assert len(data) == case["out_bytes"]
        assert hashlib.sha256(data).hexdigest() == case["out_sha256"]
    if not case["stdout_is_bam"]:
        assert ("o.bam" in got["files"]) == case["out_created"]


def test_get_barcodes_exits():
    def gb(name):
        data = name + b"\0" + b"\x10\x00\x00\x00" + b"_tail_"
        return bto.get_barcodes(data, 0, len(data))
    assert gb(b"STAGS_CELL=AC_UMI=GG_SAMPLE=T_ETAGS_r1") == (True, b"AC", b"GG", b"T")
    assert gb(b"STAGS_CELL=_UMI=_SAMPLE=_ETAGS_r1") == (True, b"", b"", b"")
    assert gb(b"read1")[0] is False
    assert gb(b"STAGS_CELL=AC_UMX=GG_SAMPLE=T_")[0] is False
    assert gb(b"STAGS_CELL=AC_UMI=GG_SAMPLX=T_")[0] is False
    # the scan for '_' does not stop at the end of the name (it is a scan of memory): the value then holds the NUL
    ok, cell, umi, sample = gb(b"STAGS_CELL=AC_UMI=GG_SAMPLE=T")
    assert ok and sample == b"T\0\x10\x00\x00\x00"
    with pytest.raises(bto.Undefined):
        bto.get_barcodes(b"STAGS_CELL=ACGT", 0, 15)
    with pytest.raises(bto.Undefined):
        gb(b"STAGS_CELL=" + b"A" * 50 + b"_UMI=_SAMPLE=_")


def test_bench_stream_generator_and_its_size_formula():
    """bench.py's bam_add_tags extra checks the output size at full length with a formula: every alignment of
    tests/bamgen.tagged_name_records gains RX (14 bytes) and CR (20), mapped ones tx (19 with 15-character names)."""
    import numpy as np
    from tests import bamgen
    rng = np.random.default_rng(1)
    n = 3000
    refs = tuple((b"ENST%011d" % i, 1000) for i in range(300))
    tid = rng.integers(-1, len(refs), n).astype(np.int32)
    rec = bamgen.tagged_name_records(rng.integers(0, 1 << 32, n).astype(np.uint64), rng.integers(0, 1 << 20, n).astype(np.uint64), tid)
    stream = bamgen.header(refs) + rec.tobytes()
    out, m = bto.add_tags_stream(stream, tx_tag=True)
    assert m == n
    assert len(out) == len(stream) + 34 * n + 19 * int((tid >= 0).sum())
    ok, cell, umi, sample = bto.get_barcodes(stream, len(bamgen.header(refs)) + 36, len(stream))
    assert ok and len(cell) == 16 and len(umi) == 10 and sample == b""
