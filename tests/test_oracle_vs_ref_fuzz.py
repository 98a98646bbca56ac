"""Strengthen the oracle pin: on seeded, mutated FASTQ images (NUL bytes, stray CRs, dropped
lines, ...) the restatement must print exactly what the reference binary (oracle/_ref/fastq_info,
built from the reference sources by oracle/Makefile) prints.  Skipped when _ref is absent."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import loader as orc
from tests import fuzz
from tests.util import strip_progress

REF_BIN = os.path.join(orc.REF_DIR, "fastq_info")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref not built")


def run_ref(args, cwd):
    p = subprocess.run([REF_BIN] + args, cwd=cwd, capture_output=True, timeout=60)
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1")


def compare(tmp, img, flags_list=("-r", "")):
    path = os.path.join(tmp, "f.fastq")
    with open(path, "wb") as f:
        f.write(img)
    for fl in flags_list:
        args = ([fl] if fl else []) + ["f.fastq"]
        rc, out, err = run_ref(args, tmp)
        flags, _ = orc.parse_args(args)
        got = orc.fastq_info(img, "f.fastq", flags=flags)
        assert got["exit"] == rc, (args, err, got["stderr"])
        assert got["stdout"] == out
        assert strip_progress(got["stderr"]) == strip_progress(err)


@pytest.mark.parametrize("kind", fuzz.MUTATIONS)
def test_single_file_modes(kind):
    rng = np.random.default_rng(abs(hash("ref" + kind)) % 100000)
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(10):
            style = ["casava", "slash", "int", "nosuffix"][trial % 4]
            img = fuzz.make_fastq(rng, int(rng.integers(1, 60)), 1, 80, style, hdr2_names=bool(trial & 1),
                                  crlf=(trial % 5 == 4), rna=(trial % 6 == 5))
            for _ in range(int(rng.integers(1, 3))):
                img = fuzz.mutate(rng, img, kind)
            compare(tmp, img)


def test_paired_and_interleaved_modes():
    rng = np.random.default_rng(99)
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(12):
            style = ["casava", "slash"][trial % 2]
            n = int(rng.integers(1, 50))
            a = fuzz.make_fastq(np.random.default_rng(trial), n, 1, 60, style, mate=1)
            b = fuzz.make_fastq(np.random.default_rng(trial), n, 1, 60, style, mate=2)
            if trial % 3 == 1:
                b = fuzz.mutate(rng, b, fuzz.MUTATIONS[trial % len(fuzz.MUTATIONS)])
            if trial % 3 == 2:
                a = fuzz.mutate(rng, a, "drop_line")
            for name, img in (("a.fastq", a), ("b.fastq", b)):
                with open(os.path.join(tmp, name), "wb") as f:
                    f.write(img)
            for args in (["a.fastq", "b.fastq"], ["-r", "-s", "a.fastq", "b.fastq"], ["b.fastq", "a.fastq"]):
                rc, out, err = run_ref(args, tmp)
                flags, pos = orc.parse_args(args)
                imgs = {"a.fastq": a, "b.fastq": b}
                got = orc.fastq_info(imgs[pos[0]], pos[0], imgs[pos[1]], pos[1], orc.ARG2_FILE, flags)
                assert (got["exit"], got["stdout"]) == (rc, out), (args, err, got["stderr"])
                assert strip_progress(got["stderr"]) == strip_progress(err)
            # interleave a and b
            la, lb = a.split(b"\n"), b.split(b"\n")
            inter = []
            for k in range(0, min(len(la), len(lb)) - 1, 4):
                inter += la[k:k + 4] + lb[k:k + 4]
            img = b"\n".join(inter) + b"\n"
            with open(os.path.join(tmp, "i.fastq"), "wb") as f:
                f.write(img)
            rc, out, err = run_ref(["i.fastq", "pe"], tmp)
            got = orc.fastq_info(img, "i.fastq", None, "pe", orc.ARG2_PE, 0)
            assert (got["exit"], got["stdout"]) == (rc, out), (err, got["stderr"])
            assert strip_progress(got["stderr"]) == strip_progress(err)


def overlong_images():
    """lines beyond the reference's gzgets buffers (src/fastq.c:249-253: 1000 bytes for the header lines, 2 500 000 for
    sequence and quality): the reference reads them in pieces, so every later `line` is out of step"""
    rng = np.random.default_rng(4)
    ok = fuzz.make_fastq(rng, 5, 30, 60, "casava")
    long_hdr = b"@" + b"h" * 1500 + b" 1:N:0:A\nACGT\n+\nIIII\n"
    n = 2_600_000
    long_read = b"@ultra 1:N:0:A\n" + b"ACGT" * (n // 4) + b"\n+\n" + b"I" * n + b"\n"
    # crafted so that the PIECES are valid records (the reference accepts these files and prints its summary):
    # two lines that are each read as two fields ...
    name = b"n" * 998
    two_lines = b"@" + name + b"ACGTTGCA\n" + b"+" + name + b"IIIIHHHH\n"
    # ... and a sequence / quality pair one byte beyond the limit whose overhang is the next field
    m = 2_499_999
    at_the_limit = b"@big 1:N:0:A\n" + b"A" * m + b"+\n" + b"I" * m + b"@next 1:N:0:A\nACGT\n+\nIIII\n"
    return {"header_1500": ok + long_hdr + ok, "header_1500_first": long_hdr + ok, "read_2.6M": ok + long_read + ok,
            "hdr2_1200": b"@r 1:N:0:A\nACGT\n+" + b"x" * 1200 + b"\nIIII\n" + ok,
            "pieces_are_records": ok + two_lines + ok, "pieces_are_records_first": two_lines + two_lines + ok,
            "pieces_at_the_read_limit": at_the_limit + ok, "header_1500_last_no_newline": ok + long_hdr[:-1],
            "header_999_exactly": ok + b"@" + b"h" * 998 + b"\nACGT\n+\nIIII\n" + ok}


@pytest.mark.parametrize("which", sorted(overlong_images()))
def test_lines_beyond_the_gzgets_buffers(which):
    """pins what the REFERENCE does with such input (the oracle restates its gzgets calls); the GPU programs refuse it
    with FQG_E_LINE_TOO_LONG instead - tests/test_gpu_cli.py::test_overlong_lines_are_refused, DESIGN.md 7.1"""
    with tempfile.TemporaryDirectory() as tmp:
        compare(tmp, overlong_images()[which])
