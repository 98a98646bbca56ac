"""The streaming pass in PARTS (round 6; k_stream_pass1_lines, fqg_stream_kernels.hip): pass 1 of a part of the image
shares its launch with the line workers of the part before, whose statistics wait in accumulators of the context until
the whole image has passed.  Results must be those of the one-launch pass (FQGPU_STREAM_PARTS=1) and of the oracle: the
finding, the statistics (every record counted once - or twice with COUNT_TWICE -, also when a flag in a LATER part sends
the image to the two-pass path after earlier parts have been counted), on reads of every shape the line workers accept or
leave to the general kernel.  Images here are 50 - 80 MB (the parts are cut at 16 MiB spans; FQGPU_STREAM_PARTS_MIN_SPANS=3
instead of the 24 of production)."""
import os

import numpy as np
import pytest

import fastq_utils_amd as fq
from oracle import loader as orc

pytestmark = pytest.mark.gpu
A = fq.abi
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def image(seed, n, lo, hi, name=b"SYN:1:FC:%d:%d 1:N:0:ACGT"):
    """n records with reads of lo..hi bases (numpy, no Python loop over bases)"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, n)
    total = int(lens.sum())
    seq = BASES[rng.integers(0, 4, total)]
    qual = (rng.integers(2, 41, total) + 33).astype(np.uint8)
    out, at = [], 0
    for i in range(n):
        L = int(lens[i])
        out.append(b"@" + (name % (i % 97, i)) + b"\n" + seq[at:at + L].tobytes() + b"\n+\n" + qual[at:at + L].tobytes() + b"\n")
        at += L
    return out


def run(ctx, img, parts, flags=0):
    os.environ["FQGPU_STREAM_PARTS"] = str(parts)
    st = A.probe_first_record(img, False)
    acc = ctx.accumulator()
    r = ctx.validate(img, acc, st, flags=flags)
    s = acc.read()
    med = acc.median()
    acc.close()
    return r, s, med


@pytest.fixture(scope="module", autouse=True)
def small_parts():
    os.environ["FQGPU_STREAM_PARTS_MIN_SPANS"] = "3"
    yield
    os.environ.pop("FQGPU_STREAM_PARTS_MIN_SPANS", None)
    os.environ.pop("FQGPU_STREAM_PARTS", None)


def same(a, b):
    ra, sa, ma = a
    rb, sb, mb = b
    keys = ("code", "record", "aux0", "aux1", "n_records", "n_lines", "consumed", "tail_lines", "stopped")
    assert {k: ra[k] for k in keys} == {k: rb[k] for k in keys}, (ra, rb)
    if ra["code"] == 0:
        assert sa == sb and ma == mb, (sa, sb, ma, mb)


def against_oracle(got, img):
    r, s, med = got
    want = orc.fastq_info(img, "p.fastq", flags=orc.FLAG_R)
    assert r["code"] == want["first"]["code"], (r, want["first"])
    if r["code"]:
        assert r["record"] == want["first"]["record"] and r["aux0"] == want["first"]["aux0"]
    else:
        w = want["summary"]
        assert (s["num_rds"], s["min_rl"], s["max_rl"], s["min_qual"], s["max_qual"], med) == (
            w["num_reads"], w["min_rl"], w["max_rl"], w["min_qual"], w["max_qual"], w["median_rl"]), (s, med, w)


@pytest.mark.parametrize("shape", ["150bp", "100_150bp", "30_150bp", "2_6kb"])
def test_parts_give_the_one_launch_result(shape):
    lo, hi, n = {"150bp": (150, 150, 200_000), "100_150bp": (100, 150, 230_000), "30_150bp": (30, 150, 300_000),
                 "2_6kb": (2000, 6000, 8_000)}[shape]
    img = b"".join(image(7, n, lo, hi))
    assert len(img) > 3 * (16 << 20)
    with fq.Context(0) as ctx:
        one = run(ctx, img, 1)
        assert one[0]["path"] == 3 and one[0]["code"] == 0
        for parts in (2, 3, 4):
            got = run(ctx, img, parts)
            assert got[0]["path"] == 3
            same(got, one)
        # the last part's share of the image (FQGPU_STREAM_LAST_PART_PCT: where the cuts lie, nothing else)
        for pct in ("1", "5", "40", "90"):
            os.environ["FQGPU_STREAM_LAST_PART_PCT"] = pct
            try:
                same(run(ctx, img, 4), one)
                same(run(ctx, img, 2), one)
            finally:
                os.environ.pop("FQGPU_STREAM_LAST_PART_PCT", None)
        against_oracle(run(ctx, img, 3), img)
        twice = run(ctx, img, 3, flags=A.VALIDATE_COUNT_TWICE)
        assert twice[1]["num_rds"] == 2 * one[1]["num_rds"] and twice[2] == one[2]
        # a second call on the same context: the accumulators of the parts were left clean
        same(run(ctx, img, 3), one)


@pytest.mark.parametrize("where", [0.1, 0.5, 0.97])
@pytest.mark.parametrize("what", ["bad_base", "short_qual", "no_plus", "no_at", "nul", "cr", "high"])
def test_findings_and_flags_in_every_part(what, where):
    recs = image(11, 200_000, 150, 150)
    k = int(len(recs) * where)
    r = recs[k]
    if what == "bad_base":
        p = r.index(b"\n")
        r = r[:p + 6] + b"X" + r[p + 7:]
    elif what == "short_qual":
        r = r[:-3] + b"\n"
    elif what == "no_plus":
        r = r.replace(b"\n+\n", b"\n-\n", 1)
    elif what == "no_at":
        r = b"X" + r[1:]
    elif what == "nul":      # NUL / CR / a byte >= 0x80 in a LATER part: the image goes to the two-pass path after the
        r = r[:-10] + b"\0" + r[-9:]   # line workers of the earlier parts have counted their records
    elif what == "cr":
        r = r[:-1] + b"\r\n"
    elif what == "high":
        r = r[:-10] + b"\xc3" + r[-9:]
    recs[k] = r
    img = b"".join(recs)
    with fq.Context(0) as ctx:
        one = run(ctx, img, 1)
        got = run(ctx, img, 3)
        same(got, one)
        against_oracle(got, img)
        # ... and the context is clean behind it: a valid image counts every record once
        clean = b"".join(image(12, 200_000, 150, 150))
        ok = run(ctx, clean, 3)
        assert ok[0]["code"] == 0 and ok[1]["num_rds"] == 200_000
        against_oracle(ok, clean)


def test_incomplete_tail_and_piecewise_input():
    """final = 0: the records of the image's last part that are complete count, the tail is the caller's"""
    recs = image(5, 210_000, 140, 150)
    img = b"".join(recs)
    cut = img[:len(img) - 200]
    os.environ["FQGPU_STREAM_PARTS"] = "3"
    with fq.Context(0) as ctx:
        st = A.probe_first_record(img, False)
        for parts in (1, 3):
            os.environ["FQGPU_STREAM_PARTS"] = str(parts)
            acc = ctx.accumulator()
            r = ctx.validate(cut, acc, st, final=False)
            assert r["code"] == 0 and r["n_records"] == 209_999 and r["consumed"] == len(img) - len(recs[-1])
            assert acc.read()["num_rds"] == 209_999
            acc.close()
            r = ctx.validate(cut, None, st, final=True, flags=A.VALIDATE_NO_STATS)
            assert r["code"] != 0 and r["record"] == 209_999  # the truncated last record


@pytest.mark.parametrize("mode", ["digests", "records"])
def test_name_modes_in_parts(mode):
    """FQG_VALIDATE_NAMES / _NAME_DIGESTS want the line index: the line workers of the parted pass store it as they go
    (room sized from the boot window).  The index calls behind such a frame give what they give behind a one-launch
    pass and through the line index alone: entries, accounted bytes, a duplicate, a wrong header; a second file finds
    every name of the first."""
    n = 200_000
    recs = image(21, n, 150, 150)
    clean = b"".join(recs)
    k = n // 3
    dup = b"".join(recs[:n - 9] + [recs[k]] + recs[n - 9:])
    wrong = b"".join(recs[:k] + [b"X" + recs[k][1:]] + recs[k + 1:])
    flags = A.VALIDATE_NAME_DIGESTS if mode == "digests" else A.VALIDATE_NAMES
    with fq.Context(0) as ctx:
        for img in (clean, dup, wrong):
            st = A.probe_first_record(img, False)
            got = []
            for parts, fl in ((4, flags), (1, flags), (1, 0)):
                os.environ["FQGPU_STREAM_PARTS"] = str(parts)
                acc = ctx.accumulator()
                r = ctx.validate(img, acc, st, flags=A.VALIDATE_COUNT_TWICE | fl)
                idx = ctx.name_index(n + 16)
                if mode == "digests":
                    idx.expect_lookups(False)
                ir = idx.insert_unique(st)
                cap = idx.names_captured()
                assert (cap > 0.9 * n) == (fl != 0), (cap, parts, fl)
                s = acc.read()
                mr = None
                if mode == "records" and img is clean:
                    ctx.validate(img, None, st, flags=A.VALIDATE_NO_STATS | fl)
                    m = idx.match_delete(st)
                    mr = (m["code"], m["n_entries"])
                got.append((r["code"], r["record"], r["n_records"], s["num_rds"], ir["code"], ir["record"], ir["n_entries"], ir["index_mem"], mr))
                idx.close()
                acc.close()
            assert got[0] == got[1] == got[2], got
            if img is clean:
                assert got[0][4] == 0 and got[0][6] == n and got[0][3] == 2 * n
                if mode == "records":
                    assert got[0][8] == (0, 0)
            else:
                assert got[0][4] != 0 or got[0][0] != 0


def test_the_program_on_pieces_that_are_cut_into_parts():
    """bin/fastq_info on a 140 MB file in pieces of 64 MiB, every piece through the parted pass (four spans each with
    FQGPU_STREAM_PARTS_MIN_SPANS=3): -r, the default mode (name digests + the parted pass storing the line index) and a
    pair, clean and with a finding late in the file - stdout, stderr and status of the oracle's serial loops"""
    import tempfile

    from tests.test_gpu_cli import compare_all
    n = 400_000
    recs = image(31, n, 150, 150, name=b"SYN:1:FC:%d:%d 1:N:0:ACGT")
    mates = [r.replace(b" 1:N:0:", b" 2:N:0:", 1) for r in recs]
    k = int(n * 0.9)
    p = recs[k].index(b"\n")
    files = {"a.fastq": b"".join(recs), "b.fastq": b"".join(mates),
             "bad.fastq": b"".join(recs[:k] + [recs[k][:p + 9] + b"X" + recs[k][p + 10:]] + recs[k + 1:]),
             "dup.fastq": b"".join(recs[:k] + [recs[5]] + recs[k:]),
             "m.fastq": b"".join(mates[:k] + mates[k + 1:])}
    env = {"FQGPU_STREAM_PARTS_MIN_SPANS": "3", "FQGPU_CHUNK_MB": "64", "FQGPU_STREAM_PARTS": "4"}
    with tempfile.TemporaryDirectory() as tmp:
        for name, img in files.items():
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(img)
        compare_all([(tmp, args, files, env) for args in (["-r", "a.fastq"], ["a.fastq"], ["-r", "bad.fastq"], ["bad.fastq"],
                                                           ["dup.fastq"], ["a.fastq", "b.fastq"], ["a.fastq", "m.fastq"])])
