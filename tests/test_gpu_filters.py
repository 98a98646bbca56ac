"""GPU parity for fastq_filter_n / fastq_trim_poly_at: the drop-in programs (C++ host + HIP kernels,
fqg_records_filter) against golden invocations of the reference binaries, against the reference
binaries themselves (oracle/_ref) on seeded inputs - at several tile sizes, with long reads whose
tiles take the direct path - and against the Python oracle."""
import gzip
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import filter_oracle as fo
from tests.util import GOLD, REPO, strip_progress, SideBySide

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(GOLD, "filters.json")))
BIN_N = os.path.join(REPO, "bin", "fastq_filter_n")
BIN_T = os.path.join(REPO, "bin", "fastq_trim_poly_at")
REF_N = os.path.join(REPO, "oracle", "_ref", "fastq_filter_n")
REF_T = os.path.join(REPO, "oracle", "_ref", "fastq_trim_poly_at")


def run(binary, argv0, args, cwd, env=None, stdin=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    p = subprocess.run([argv0] + args, executable=binary, cwd=cwd, capture_output=True, timeout=600, env=e, input=stdin)
    return p.returncode, p.stdout, p.stderr.decode("latin-1")


def same_text(packed, got):
    assert packed["len"] == len(got)
    assert packed["sha256"] == hashlib.sha256(got).hexdigest()


def pick(cases, every):
    """all the failing / special invocations, and every `every`-th of the rest"""
    return [c for i, c in enumerate(cases) if c["exit"] != 0 or i % every == 0 or "syn_filters" in " ".join(c["args"])]


def ids(cases):
    return [" ".join(c["args"])[-70:] or "(no arguments)" for c in cases]


FN_CASES = pick(GOLDEN["filter_n"], 9)
TP_CASES = pick(GOLDEN["trim_poly_at"], 13)


# (the programs of all cases start side by side the first time one is asked for: tests/util.py)
FN_RUNS = SideBySide(lambda i: run(BIN_N, "fastq_filter_n", FN_CASES[i]["args"], GOLD), range(len(FN_CASES)))


@pytest.mark.parametrize("i", range(len(FN_CASES)), ids=ids(FN_CASES))
def test_filter_n_golden(i):
    case = FN_CASES[i]
    rc, out, err = FN_RUNS.get(i)
    assert rc == case["exit"], err
    same_text(case["stdout"], out)
    assert strip_progress(err) == strip_progress(case["stderr"])


def tp_run(i):
    case = TP_CASES[i]
    with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
        rel = os.path.relpath(tmp, GOLD)
        args = [rel + "/o.fastq.gz" if a == "OUT" else a for a in case["args"]]
        rc, out, err = run(BIN_T, "fastq_trim_poly_at", args, GOLD)
        path = os.path.join(tmp, "o.fastq.gz")
        raw = open(path, "rb").read() if os.path.exists(path) else None
    return rc, out, err.replace(rel + "/", "SCRATCH/"), raw


TP_RUNS = SideBySide(tp_run, range(len(TP_CASES)))


@pytest.mark.parametrize("i", range(len(TP_CASES)), ids=ids(TP_CASES))
def test_trim_poly_at_golden(i):
    case = TP_CASES[i]
    rc, out, err, raw = TP_RUNS.get(i)
    assert rc == case["exit"], err
    assert out.decode("latin-1") == case["stdout"]
    assert strip_progress(err) == strip_progress(case["stderr"])
    if case["out"] is not None:
        same_text(case["out"], gzip.decompress(raw) if raw else b"")


def make_reads(rng, n, long_every=0):
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)
    out = []
    for i in range(n):
        ln = int(rng.integers(20, 151))
        if long_every and i % long_every == long_every - 1:
            ln = int(rng.integers(4000, 20000))
        s = bytearray(bases[rng.choice(5, ln, p=[0.245, 0.245, 0.245, 0.245, 0.02])].tobytes())
        r = rng.random()
        if r < 0.25:
            s += b"A" * int(rng.integers(1, 40))
        elif r < 0.45:
            s = bytearray(b"T" * int(rng.integers(1, 40))) + s
        elif r < 0.5:
            for j in range(0, len(s), 3):
                s[j] = ord("N")
        q = (rng.integers(2, 41, len(s)) + 33).astype(np.uint8).tobytes()
        out.append(b"@R%d some text\n" % i + bytes(s) + b"\n+\n" + q + b"\n")
    return b"".join(out)


@pytest.mark.skipif(not os.path.exists(REF_N), reason="oracle/_ref not built")
@pytest.mark.parametrize("lds", [None, "4096", "65536"], ids=["default_tiles", "tiny_tiles", "large_tiles"])
@pytest.mark.parametrize("flags", [[], ["-n", "5"], ["-n", "40"]], ids=["any_n", "n5", "n40"])
def test_filter_n_against_reference_binary(flags, lds):
    rng = np.random.default_rng(len(flags) + 11)
    img = make_reads(rng, 20000, long_every=97)
    env = {"FQGPU_CHUNK_MB": "1"}
    if lds:
        env["FQGPU_BC_LDS"] = lds
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "in.fastq"), "wb") as f:
            f.write(img)
        a = run(REF_N, "fastq_filter_n", flags + ["in.fastq"], d)
        b = run(BIN_N, "fastq_filter_n", flags + ["in.fastq"], d, env)
        assert a[0] == b[0] == 0
        assert a[1] == b[1]
        assert 0 < len(a[1]) < len(img)
        assert strip_progress(a[2]) == strip_progress(b[2])


@pytest.mark.skipif(not os.path.exists(REF_T), reason="oracle/_ref not built")
@pytest.mark.parametrize("lds", [None, "4096", "65536"], ids=["default_tiles", "tiny_tiles", "large_tiles"])
@pytest.mark.parametrize("flags", [[], ["--min_poly_at_len", "5", "--min_len", "30"], ["--min_poly_at_len", "1", "--min_len", "1"],
                                   ["--min_poly_at_len", "25", "--min_len", "100"]], ids=["defaults", "p5_l30", "p1_l1", "p25_l100"])
def test_trim_poly_at_against_reference_binary(flags, lds):
    rng = np.random.default_rng(len(flags) + 5)
    img = make_reads(rng, 20000, long_every=89)
    env = {"FQGPU_CHUNK_MB": "1"}
    if lds:
        env["FQGPU_BC_LDS"] = lds
    with tempfile.TemporaryDirectory() as d:
        with gzip.open(os.path.join(d, "in.fastq.gz"), "wb", compresslevel=1) as f:
            f.write(img)
        res = []
        for binary, e, o in ((REF_T, None, "a.gz"), (BIN_T, env, "b.gz")):
            rc, out, err = run(binary, "fastq_trim_poly_at", ["--file", "in.fastq.gz", "--outfile", o] + flags, d, e)
            res.append((rc, out, strip_progress(err), gzip.decompress(open(os.path.join(d, o), "rb").read())))
        assert res[0][0] == res[1][0] == 0
        assert res[0][1:] == res[1][1:]
        assert 0 < len(res[0][3]) < len(img)


def test_stdin_stdout_and_truncated_input_like_the_oracle():
    rng = np.random.default_rng(2)
    img = make_reads(rng, 3000)
    cut = img[: img.rindex(b"\n+\n") + 3]  # the last record loses its quality line
    with tempfile.TemporaryDirectory() as d:
        files = {"in.fastq": img, "cut.fastq": cut}
        for k, v in files.items():
            with open(os.path.join(d, k), "wb") as f:
                f.write(v)
        # run_tests.sh:205: stdin to stdout, gzipped
        rc, out, err = run(BIN_T, "fastq_trim_poly_at", ["--file", "-", "--outfile", "-", "--min_poly_at_len", "4"], d,
                           stdin=gzip.compress(img, 1))
        want = fo.trim_poly_at(["--file", "in.fastq", "--outfile", "x", "--min_poly_at_len", "4"], lambda p: files[p])
        assert rc == 0 and gzip.decompress(out) == want["out"]
        assert strip_progress(err) == want["stderr"].decode("latin-1")
        for binary, argv0, args, oracle in ((BIN_N, "fastq_filter_n", ["-n", "3", "cut.fastq"], fo.filter_n),
                                            (BIN_T, "fastq_trim_poly_at", ["--file", "cut.fastq", "--outfile", "o.gz"],
                                             fo.trim_poly_at)):
            rc, out, err = run(binary, argv0, args, d, {"FQGPU_CHUNK_MB": "1"})
            want = oracle(args, lambda p: files[p])
            assert rc == want["exit"] == 1
            assert out == want["stdout"]
            assert strip_progress(err) == want["stderr"].decode("latin-1")


def odd_inputs():
    rng = np.random.default_rng(31)
    base = make_reads(rng, 400)
    lines = base.split(b"\n")
    out = {}
    out["no_final_newline"] = base[:-1]
    out["crlf"] = base.replace(b"\n", b"\r\n")
    # empty sequence and quality lines, a one-base read, a read that is nothing but A, one of T, one of N
    special = (b"@e1\n\n+\n\n" b"@one\nA\n+\nI\n" b"@allA\n" + b"A" * 60 + b"\n+\n" + b"I" * 60 + b"\n"
               b"@allT\n" + b"T" * 33 + b"\n+\n" + b"5" * 33 + b"\n" b"@allN\n" + b"N" * 20 + b"\n+\n" + b"#" * 20 + b"\n"
               b"@lower\nacgtnnnnaaaaaaaaaaaa\n+\nIIIIIIIIIIIIIIIIIIII\n" b"@tlow\nttttttttttttacgtacgtacgt\n+\nIIIIIIIIIIIIIIIIIIIIIIII\n")
    out["special_records"] = special + base[:20000].rsplit(b"\n@R", 1)[0] + b"\n"
    out["special_at_the_end_unterminated"] = base[:30000].rsplit(b"\n@R", 1)[0] + b"\n" + special[:-1]
    return out


@pytest.mark.skipif(not os.path.exists(REF_N), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", sorted(odd_inputs()))
def test_odd_inputs_against_reference_binaries(name):
    img = odd_inputs()[name]
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "in.fastq"), "wb") as f:
            f.write(img)
        for flags in ([], ["-n", "20"]):
            a = run(REF_N, "fastq_filter_n", flags + ["in.fastq"], d)
            b = run(BIN_N, "fastq_filter_n", flags + ["in.fastq"], d, {"FQGPU_BC_LDS": "8192"})
            assert a[0] == b[0] and a[1] == b[1] and strip_progress(a[2]) == strip_progress(b[2]), (name, flags, b[2][-300:])
        for flags in (["--min_poly_at_len", "3", "--min_len", "1"], ["--min_poly_at_len", "12"], ["--min_poly_at_len", "1", "--min_len", "0"]):
            res = []
            for binary, o in ((REF_T, "a.gz"), (BIN_T, "b.gz")):
                rc, out, err = run(binary, "fastq_trim_poly_at", ["--file", "in.fastq", "--outfile", o] + flags, d)
                raw = open(os.path.join(d, o), "rb").read() if os.path.exists(os.path.join(d, o)) else None
                res.append((rc, out, strip_progress(err), gzip.decompress(raw) if raw else raw))
            assert res[0] == res[1], (name, flags, res[1][2][-300:])
