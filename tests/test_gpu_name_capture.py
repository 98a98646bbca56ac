"""The read-name kernels work from the header records that the streaming pass captures (FQG_VALIDATE_NAMES:
k_stream_pass1<names> -> k_names_pass, fqg_index_kernels.hip) and go through the line index for every header the
capture cannot vouch for.  Both ways must give what the reference's serial loops give (src/fastq.c:396-439,
src/fastq_info.c:333-362, src/fastq_filterpair.c:108-216).  Small files do not reach the streaming pass on their own, so
everything here runs with FQGPU_STREAM_MIN=256; inputs are built to land on each seam of the capture: chunks whose
speculated line type is wrong, chunks with more headers than record slots, header lines that straddle a chunk, lines
longer than a record holds, names of 48 bytes and more (the bucket carries 48), duplicates whose copies arrive by
different ways, a name asked for twice across two pieces of the second file."""
import gzip
import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import fastq_utils_amd as fq
from oracle import loader as orc
from tests import fuzz
from tests.test_gpu_cli import GOLDEN, POOL, compare_all, compare_with_oracle, goes_to_several_devices, put, run_cli
from tests.util import GOLD, REPO, strip_progress, thinned

pytestmark = pytest.mark.gpu
STREAM = {"FQGPU_STREAM_MIN": "256"}
PIECES = {"FQGPU_STREAM_MIN": "256", "FQGPU_CHUNK_MB": "1"}
FILTERPAIR = os.path.join(REPO, "bin", "fastq_filterpair")
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def rec(name, rng, ln):
    s = BASES[rng.integers(0, 4, ln)].tobytes()
    q = (rng.integers(5, 40, ln) + 33).astype(np.uint8).tobytes()
    return b"@" + name + b"\n" + s + b"\n+\n" + q + b"\n"


def write(tmp, files):
    for k, v in files.items():
        with open(os.path.join(tmp, k), "wb") as f:
            f.write(v)


def test_golden_index_and_pairing_invocations_with_the_capture():
    """golden invocations that build an index (no -r), streamed: the interleaved ones, and of the others those that
    tests/test_gpu_cli.py does not run over several devices (goes_to_several_devices)"""
    cases = [c for c in GOLDEN if "-r" not in c["args"] and ("pe" in c["args"] or not goes_to_several_devices(c))]
    assert len(cases) > 100
    cases = thinned(cases)

    def one(case):
        rc, out, err = run_cli(case["args"], GOLD, STREAM)
        ok = (rc == case["exit"] and out == case["stdout"] and strip_progress(err) == strip_progress(case["stderr"]))
        return None if ok else (case["args"], rc, case["exit"], err[-400:], case["stderr"][-400:])

    with ThreadPoolExecutor(POOL) as ex:
        bad = [b for b in ex.map(one, cases) if b]
    assert not bad, f"{len(bad)} of {len(cases)} differ; first: {bad[:3]}"


@pytest.mark.parametrize("kind", fuzz.MUTATIONS)
def test_mutated_files(kind):
    rng = np.random.default_rng(abs(hash("cap" + kind)) % 100000)
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(4):
            style = ["casava", "slash", "int", "nosuffix"][trial % 4]
            img = fuzz.make_fastq(rng, int(rng.integers(40, 600)), 1, 120, style, hdr2_names=bool(trial & 1), rna=(trial == 2))
            img = fuzz.mutate(rng, img, kind)
            d = put(tmp, "t%d" % trial, {"f.fastq": img})
            jobs += [(d, args, {"f.fastq": img}, STREAM) for args in (["f.fastq"], ["f.fastq", "pe"])]
        compare_all(jobs)


def test_duplicates_and_pairs():
    rng = np.random.default_rng(55)
    jobs = []
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(10):
            style = ["casava", "slash"][trial % 2]
            n = int(rng.integers(200, 3000))
            a = fuzz.make_fastq(np.random.default_rng(trial), n, 1, 90, style, mate=1)
            b = fuzz.make_fastq(np.random.default_rng(trial), n, 1, 90, style, mate=2)
            la, lb = a.split(b"\n"), b.split(b"\n")
            k = int(rng.integers(0, n))
            if trial % 5 == 1:  # a record of file 1 three times ("earliest repeat")
                la = la[:-1] + la[4 * k:4 * k + 4] + la[4 * k:4 * k + 4] + [b""]
            if trial % 5 == 2:  # a record missing from file 2
                lb = lb[:4 * k] + lb[4 * k + 4:]
            if trial % 5 == 3:  # file 2 in another order (still paired)
                recs = [lb[4 * i:4 * i + 4] for i in range(n)]
                lb = [x for i in rng.permutation(n) for x in recs[i]] + [b""]
            if trial % 5 == 4:  # a name twice in file 2
                lb = lb[:-1] + lb[4 * k:4 * k + 4] + [b""]
            files = {"a.fastq": b"\n".join(la), "b.fastq": b"\n".join(lb)}
            d = put(tmp, "t%d" % trial, files)
            jobs += [(d, args, files, env) for args in (["a.fastq"], ["a.fastq", "b.fastq"], ["b.fastq", "a.fastq"])
                     for env in (STREAM, PIECES)]
        compare_all(jobs)


def names_of_every_kind(rng, n, mate=1):
    """records whose headers land on every branch of the capture: short and 50-70 byte names (the bucket holds 48, the
    record 60), Casava headers whose blank lies beyond the record, headers of several hundred bytes, reads of one base
    (their sequence line looks like a '+' line: the chunk's speculated type is wrong) and runs of very short records (more
    headers in a chunk than record slots)"""
    out = []
    for i in range(n):
        kind = i % 11
        if kind == 0:
            name = b"N%d:%s %d:N:0:A" % (i, b"x" * int(rng.integers(30, 70)), mate)   # name of 32..72 bytes, then the blank
        elif kind == 1:
            name = b"L%d:%s %d:N:0:A" % (i, b"y" * int(rng.integers(100, 400)), mate)  # beyond the register path too
        elif kind == 2:
            name = b"S%d %d:N:0:%s" % (i, mate, b"C" * int(rng.integers(0, 200)))      # short name, long comment
        elif kind == 3:
            name = b"T%d:%s/%d %d:N:0:A" % (i, b"z" * int(rng.integers(0, 50)), mate, mate)  # Casava with a /1 in front of the blank
        else:
            name = b"R:%d:%d %d:N:0:ACGT" % (i % 7, i, mate)
        ln = 1 if (i % 97) == 5 else (int(rng.integers(1, 6)) if (i // 200) % 3 == 1 else int(rng.integers(20, 160)))
        out.append(rec(name, rng, ln))
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_every_seam_of_the_capture(seed):
    rng = np.random.default_rng(seed)
    n = 6000
    r1 = names_of_every_kind(np.random.default_rng(seed), n, 1)
    r2 = names_of_every_kind(np.random.default_rng(seed), n, 2)
    dup = list(r1)
    k = int(rng.integers(0, n))
    dup.insert(int(rng.integers(k + 1, n)), r1[k])
    perm = rng.permutation(n)
    files = {"a.fastq": b"".join(r1), "b.fastq": b"".join(r2), "d.fastq": b"".join(dup),
             "s.fastq": b"".join(r2[i] for i in perm), "m.fastq": b"".join(r2[:k] + r2[k + 1:])}
    with tempfile.TemporaryDirectory() as tmp:
        write(tmp, files)
        compare_all([(tmp, args, files, env) for env in (STREAM, PIECES)
                     for args in (["a.fastq"], ["d.fastq"], ["a.fastq", "b.fastq"], ["a.fastq", "s.fastq"], ["a.fastq", "m.fastq"],
                                  ["m.fastq", "a.fastq"])])


@pytest.mark.parametrize("style", ["slash", "nosuffix", "int"])
def test_names_that_are_the_whole_line(style):
    """not Casava: the name runs to the end of the line, so a line longer than the record goes through the line index"""
    n = 4000
    pad = lambda i: b"p" * (0 if i % 3 else 40 + (i * 7919) % 50)  # noqa: E731  (the same for both mates of a read)

    def name(i, mate):
        if style == "slash":
            return b"read%s.%d/%d" % (pad(i), i, mate)
        if style == "int":
            return b"%d" % (10 ** 70 + i if i % 5 == 0 else i)
        return b"read%s_%d_x" % (pad(i), i)

    rngs = [np.random.default_rng(3), np.random.default_rng(3)]
    a = b"".join(rec(name(i, 1), rngs[0], int(rngs[0].integers(10, 120))) for i in range(n))
    b = b"".join(rec(name(i, 2), rngs[1], int(rngs[1].integers(10, 120))) for i in range(n))
    files = {"a.fastq": a, "b.fastq": b}
    with tempfile.TemporaryDirectory() as tmp:
        write(tmp, files)
        compare_all([(tmp, args, files, PIECES) for args in (["a.fastq"], ["a.fastq", "b.fastq"])])


def test_a_file2_name_asked_twice_across_two_pieces():
    """the second copy of a file-2 record arrives in a later 1 MiB piece than the first: the entry was taken by an asker
    of an earlier piece, so the later one is the record the serial loop stops at (src/fastq_info.c:338)"""
    a = fuzz.make_fastq(np.random.default_rng(1), 30000, 50, 150, "casava", mate=1)
    b = fuzz.make_fastq(np.random.default_rng(1), 30000, 50, 150, "casava", mate=2)
    lb = b.split(b"\n")
    twice = lb[4 * 100:4 * 100 + 4]
    d2 = b"\n".join(lb[:4 * 25000] + twice + lb[4 * 25000:])   # record 100 again as record 25000 (piece 6 or so)
    files = {"a.fastq": a, "d2.fastq": d2}
    with tempfile.TemporaryDirectory() as tmp:
        write(tmp, files)
        for env in ({"FQGPU_CHUNK_MB": "1"}, PIECES):
            compare_with_oracle(tmp, ["a.fastq", "d2.fastq"], files, env)
            want = orc.fastq_filterpair(a, "a.fastq", d2, "d2.fastq")
            p = subprocess.run([FILTERPAIR, "a.fastq", "d2.fastq", "p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"], cwd=tmp,
                               capture_output=True, timeout=300, env=dict(os.environ, **env))
            assert p.returncode == want["exit"], p.stderr[-400:]
            assert strip_progress(p.stderr.decode("latin-1")) == strip_progress(want["stderr"])
            for k, name in enumerate(("p1", "p2", "up")):
                assert gzip.decompress(open(os.path.join(tmp, name + ".fastq.gz"), "rb").read()) == want["files"][k], name


def test_the_capture_is_what_runs():
    """on regular reads nearly every name comes from a capture record (the exceptions: headers that straddle a chunk),
    and the finding is the oracle's either way"""
    os.environ["FQGPU_STREAM_MIN"] = "256"
    try:
        rng = np.random.default_rng(8)
        n = 20000
        img = fuzz.make_fastq(rng, n, 100, 150, "casava")
        lines = img.split(b"\n")
        dup = b"\n".join(lines[:-1] + lines[4 * 4321:4 * 4321 + 4] + [b""])
        with fq.Context(0) as ctx:
            for image, want_code in ((img, 0), (dup, 3)):  # 3 = FQG_E_DUP_NAME
                st = fq.abi.probe_first_record(image, False)
                for flags, captured in ((fq.abi.VALIDATE_NAMES, True), (0, False)):
                    acc = ctx.accumulator()
                    r = ctx.validate(image, acc, st, flags=fq.abi.VALIDATE_COUNT_TWICE | flags)
                    assert r["code"] == 0 and r["path"] == 3
                    idx = ctx.name_index(1024)  # (grows: the earlier frames are re-inserted through the line index)
                    ir = idx.insert_unique(st)
                    assert ir["code"] == want_code, ir
                    if want_code:
                        assert ir["record"] == n
                    else:
                        assert ir["n_entries"] == n
                    got = idx.names_captured()
                    assert (got > 0.95 * n) if captured else got == 0, (got, n)
                    # the same names as a second file: all of them found, nothing left
                    r2 = ctx.validate(img, None, st, flags=fq.abi.VALIDATE_NO_STATS | flags)
                    mr = idx.match_delete(st)
                    assert mr["code"] == 0 and mr["n_entries"] == 0, mr
                    idx.close()
                    acc.close()
    finally:
        os.environ.pop("FQGPU_STREAM_MIN", None)


@pytest.mark.parametrize("build", ["cas", "lds", "lds2"])
@pytest.mark.parametrize("style", ["casava", "slash", "int", "nosuffix"])
def test_digests_give_what_records_and_the_line_index_give(style, build):
    """FQG_VALIDATE_NAME_DIGESTS (the streaming pass canonicalises and hashes the headers itself, 16 bytes per header):
    an index that keeps no name records takes its names from them; finding, entries and accounted bytes are those of the
    capture records and of the line index - on regular names, on names of every kind, with a duplicate, with a wrong
    header - and an index that will be asked ignores the digests.  build: the digests through k_names_pass (one CAS per
    name), or the table built part by part in LDS (fqg_names_build_kernels.hip) with one level of buckets / two"""
    os.environ["FQGPU_STREAM_MIN"] = "256"
    os.environ["FQGPU_NAMES_BUILD"] = "0" if build == "cas" else "1"
    try:
        rng = np.random.default_rng(21)
        n = 12000
        expect = 2 * n if build != "lds2" else 1 << 21  # (2^22 slots and more: parts behind two levels of buckets)
        if style == "casava":
            recs = [r for r in names_of_every_kind(np.random.default_rng(4), n, 1)]
        else:
            img0 = fuzz.make_fastq(rng, n, 30, 150, style)
            l0 = img0.split(b"\n")
            recs = [b"\n".join(l0[4 * i:4 * i + 4]) + b"\n" for i in range(n)]
        clean = b"".join(recs)
        k = int(rng.integers(0, n - 10))
        dup = b"".join(recs[:n - 5] + [recs[k]] + recs[n - 5:])
        three = b"".join(recs[:n // 2] + [recs[k // 2]] + recs[n // 2:] + [recs[k // 2]])
        wrong = b"".join(recs[:k] + [b"X" + recs[k][1:]] + recs[k + 1:])
        with fq.Context(0) as ctx:
            for image in (clean, dup, three, wrong):
                st = fq.abi.probe_first_record(image, False)
                got = []
                for flags, lookups in ((fq.abi.VALIDATE_NAME_DIGESTS, False), (fq.abi.VALIDATE_NAMES, False), (0, False),
                                       (fq.abi.VALIDATE_NAME_DIGESTS, True), (fq.abi.VALIDATE_NAMES, True)):
                    r = ctx.validate(image, None, st, flags=fq.abi.VALIDATE_NO_STATS | flags)
                    idx = ctx.name_index(expect)
                    if not lookups:
                        idx.expect_lookups(False)
                    ir = idx.insert_unique(st)
                    cap = idx.names_captured()
                    if r["path"] == 3 and flags == fq.abi.VALIDATE_NAME_DIGESTS and not lookups:
                        assert cap > 0.5 * n, (cap, n)  # the digests are what ran
                    if flags == 0 or (lookups and flags == fq.abi.VALIDATE_NAME_DIGESTS):
                        assert cap == 0, cap  # (digests hold no bytes: an index that keeps names reads the lines)
                    got.append((ir["code"], ir["record"], ir["n_entries"], ir["index_mem"]))
                    if lookups and flags == fq.abi.VALIDATE_NAMES and image is clean:
                        # the table a build from the capture RECORDS left (keys + name records) answers a second file:
                        # every name found, on its bytes, nothing left
                        ctx.validate(image, None, st, flags=fq.abi.VALIDATE_NO_STATS | flags)
                        mr = idx.match_delete(st)
                        assert mr["code"] == 0 and mr["n_entries"] == 0, mr
                    idx.close()
                assert got[0] == got[1] == got[2] == got[3] == got[4], (style, got)
            # two frames into one index: the second one's parts are merged with what the table holds; a name of the
            # first frame again in the second is the second frame's finding
            half = b"".join(recs[:n // 2])
            rest_clean = b"".join(recs[n // 2:])
            rest_dup = b"".join(recs[n // 2:n - 7] + [recs[k // 2]] + recs[n - 7:])
            st = fq.abi.probe_first_record(half, False)
            for second in (rest_clean, rest_dup):
                got = []
                for flags in (fq.abi.VALIDATE_NAME_DIGESTS, 0):
                    idx = ctx.name_index(expect)
                    idx.expect_lookups(False)
                    out = []
                    for image in (half, second):
                        ctx.validate(image, None, st, final=True, flags=fq.abi.VALIDATE_NO_STATS | flags)
                        ir = idx.insert_unique(st)
                        out.append((ir["code"], ir["record"], ir["n_entries"], ir["index_mem"]))
                    got.append(out)
                    idx.close()
                assert got[0] == got[1], (style, got)
                assert (got[0][1][0] != 0) == (second is rest_dup), got
    finally:
        os.environ.pop("FQGPU_STREAM_MIN", None)
        os.environ.pop("FQGPU_NAMES_BUILD", None)
