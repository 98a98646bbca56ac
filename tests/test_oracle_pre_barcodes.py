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


REF = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "fastq_pre_barcodes")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("open_end", [True, False], ids=["no_last_newline", "last_newline"])
@pytest.mark.parametrize("which", ["index_file_short", "read_file_short"])
def test_loop_condition_of_the_reference_binary(which, open_end, tmp_path):
    """fastq_files_eof (src/fastq_pre_barcodes.c:288-297, :594): the oracle's gzeof - true once a line without '\\n' has been
    read up to the end of the file - against the reference binary itself, on the four cases the GPU test
    test_loop_condition_ends_the_loop_before_a_truncated_record_is_read holds the program to"""
    import subprocess
    rec = lambda i, mate, n: (b"@SYN:1:FC:1:%d:%d:%d %d:N:0:ACGT\n" % (i % 97, i % 1013, i, mate) + b"ACGT" * n + b"\n+\n" + b"IIII" * n + b"\n")
    r1 = b"".join(rec(i, 1, 7) for i in range(5))    # index reads, 28 bp
    r2 = b"".join(rec(i, 2, 20) for i in range(5))   # cDNA reads
    if which == "index_file_short":
        short, long_ = r1, r2 + b"\n".join(r2.split(b"\n")[:2]) + b"\n"
        files = {"r1.fastq": short[:-1] if open_end else short, "r2.fastq": long_}
    else:
        short, long_ = r2, r1 + b"\n".join(r1.split(b"\n")[:2]) + b"\n"
        files = {"r1.fastq": long_, "r2.fastq": short[:-1] if open_end else short}
    args = ["--read1", "r2.fastq", "--index1", "r1.fastq", "--umi_read", "index1", "--umi_offset", "16", "--umi_size", "10",
            "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16", "--phred_encoding", "33", "--sam", "--outfile1", "-"]
    for name, img in files.items():
        (tmp_path / name).write_bytes(img)
    p = subprocess.run(["fastq_pre_barcodes"] + args, executable=os.path.abspath(REF), cwd=tmp_path, capture_output=True, timeout=60)
    got = pbo.run_pre_barcodes(args, lambda n: files[n])
    assert got["exit"] == p.returncode
    assert got["stdout"] == p.stdout.decode("latin-1")
    assert strip_progress(got["stderr"]) == strip_progress(p.stderr.decode("latin-1"))
