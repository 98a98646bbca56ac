"""GPU parity for fastq_pre_barcodes: bin/fastq_pre_barcodes (C++ host + HIP kernels) against
the golden invocations captured from the reference binary, and against the reference binary itself
(oracle/_ref) / the Python oracle on seeded 10x-style inputs."""
import gzip
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import pre_barcodes_oracle as pbo
from tests.util import GOLD, REPO, SideBySide, read_image, strip_progress

pytestmark = pytest.mark.gpu
BIN = os.path.join(REPO, "bin", "fastq_pre_barcodes")
REF = os.path.join(REPO, "oracle", "_ref", "fastq_pre_barcodes")
GOLDEN = json.load(open(os.path.join(GOLD, "pre_barcodes.json")))


def run(binary, args, cwd, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    p = subprocess.run(["fastq_pre_barcodes"] + args, executable=binary, cwd=cwd, capture_output=True, timeout=600, env=e)
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1")


def gunzip_file(path):
    if not os.path.exists(path):
        return None
    raw = open(path, "rb").read()
    return gzip.decompress(raw).decode("latin-1") if raw else ""


def golden_run(job):
    """one golden invocation (index into GOLDEN, environment): what the program printed and wrote"""
    i, env = job
    case = GOLDEN[i]
    with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
        rel = os.path.relpath(tmp, GOLD)
        args = [a.replace("OUT1", rel + "/o1.fastq.gz").replace("OUT2", rel + "/o2.fastq.gz") for a in case["args"]]
        rc, out, err = run(BIN, args, GOLD, dict(env) if env else None)
        out, err = out.replace(rel + "/", "SCRATCH/"), err.replace(rel + "/", "SCRATCH/")
        files = {tag: gunzip_file(os.path.join(tmp, fn)) for tag, fn in (("OUT1", "o1.fastq.gz"), ("OUT2", "o2.fastq.gz"))}
    return rc, out, err, files


def golden_check(case, got, files_also_on_failure=False):
    rc, out, err, files = got
    assert rc == case["exit"], err
    assert out == case["stdout"]
    assert strip_progress(err) == strip_progress(case["stderr"])
    if case["exit"] == 0 or files_also_on_failure:
        for tag in ("OUT1", "OUT2"):
            if tag in case["files"]:
                assert files[tag] == case["files"][tag]


GOLDEN_IDS = [str(i) + ":" + " ".join(c["args"])[:60] for i, c in enumerate(GOLDEN)]
PLAIN_RUNS = SideBySide(golden_run, [(i, None) for i in range(len(GOLDEN))])


@pytest.mark.parametrize("i", range(len(GOLDEN)), ids=GOLDEN_IDS)
def test_golden_invocations(i):
    golden_check(GOLDEN[i], PLAIN_RUNS.get((i, None)))


def make_10x(rng, n, umi_q_low=0.05, short=0.01):
    """R1 = 16 bp cell + 10 bp UMI, R2 = 40..150 bp cDNA; same names before the blank."""
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)
    r1, r2 = [], []
    for i in range(n):
        name = b"SYN:1:FC:%d:%d:%d:%d" % (i % 8 + 1, i % 97, i % 1013, i)
        l1 = 26 if rng.random() > short else int(rng.integers(5, 26))
        s1 = bases[rng.integers(0, 4, l1)].tobytes()
        q1 = (rng.integers(12, 41, l1) + 33).astype(np.uint8)
        if rng.random() < umi_q_low and l1:
            q1[int(rng.integers(0, l1))] = 33 + int(rng.integers(0, 10))
        l2 = int(rng.integers(40, 151))
        s2 = bases[rng.integers(0, 5, l2)].tobytes()
        q2 = (rng.integers(2, 41, l2) + 33).astype(np.uint8).tobytes()
        r1.append(b"@" + name + b" 1:N:0:ACGT\n" + s1 + b"\n+\n" + q1.tobytes() + b"\n")
        r2.append(b"@" + name + b" 2:N:0:ACGT\n" + s2 + b"\n+\n" + q2 + b"\n")
    return b"".join(r1), b"".join(r2)


_TEN_X = {}


def ten_x_pairs_20000():
    """the 20 000 pairs of test_several_devices_against_reference_binary (made once: a second of Python per call)"""
    if not _TEN_X:
        _TEN_X["p"] = make_10x(np.random.default_rng(41), 20000)
    return _TEN_X["p"]


V2 = ["--read1", "r2.fastq", "--index1", "r1.fastq", "--umi_read", "index1", "--umi_offset", "16", "--umi_size", "10",
      "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16", "--phred_encoding", "33", "--min_qual", "10"]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
@pytest.mark.parametrize("extra", [["--outfile1", "o.fastq.gz"], ["--sam", "--outfile1", "-"], ["--sam", "--10x", "--outfile1", "-"],
                                   ["--outfile1", "o.fastq.gz", "--read1_offset", "3", "--read1_size", "50"]])
def test_10x_v2_against_reference_binary(extra):
    rng = np.random.default_rng(len(extra) * 7 + 1)
    r1, r2 = make_10x(rng, 20000)
    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        res = []
        for d, binary, env in ((a, REF, None), (b, BIN, {"FQGPU_CHUNK_MB": "1"})):
            for name, img in (("r1.fastq", r1), ("r2.fastq", r2)):
                with open(os.path.join(d, name), "wb") as f:
                    f.write(img)
            rc, out, err = run(binary, V2 + extra, d, env)
            res.append((rc, out, strip_progress(err), gunzip_file(os.path.join(d, "o.fastq.gz"))))
        assert res[0][0] == res[1][0] == 0
        assert res[0][1] == res[1][1]
        assert res[0][2] == res[1][2]
        assert res[0][3] == res[1][3]


def test_name_mismatch_and_oracle_agreement():
    rng = np.random.default_rng(3)
    r1, r2 = make_10x(rng, 3000, 0.0, 0.0)
    lines = r2.split(b"\n")
    lines[4 * 1234] = lines[4 * 1234].replace(b"SYN:", b"SYX:")
    r2bad = b"\n".join(lines)
    with tempfile.TemporaryDirectory() as d:
        files = {"r1.fastq": r1, "r2.fastq": r2bad}
        for name, img in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(img)
        args = V2 + ["--sam", "--outfile1", "-"]
        rc, out, err = run(BIN, args, d)
        want = pbo.run_pre_barcodes(args, lambda n: files[n])
        assert rc == want["exit"] == 3
        assert "Readnames do not match across files (read #1235)" in err
        assert out == want["stdout"]
        assert strip_progress(err) == strip_progress(want["stderr"])


# ---- the tiled kernels: every specialisation (set of input files), tiny tiles, tiles that do not fit ----
def make_long_mix(rng, n, long_every=37, long_len=(3000, 9000)):
    """Paired records with matching names; every long_every-th pair is a long read (its tile cannot be
    staged in LDS and takes the direct path)."""
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)
    r1, r2 = [], []
    for i in range(n):
        name = b"LR:%d:%d" % (i % 13, i)
        for out, mate in ((r1, b"1"), (r2, b"2")):
            ln = int(rng.integers(*long_len)) if i % long_every == long_every - 1 else int(rng.integers(30, 120))
            s = bases[rng.integers(0, 5, ln)].tobytes()
            q = (rng.integers(15, 41, ln) + 33).astype(np.uint8).tobytes()
            out.append(b"@" + name + b"/" + mate + b"\n" + s + b"\n+\n" + q + b"\n")
    return b"".join(r1), b"".join(r2)


FILE_SETS = {
    "read1_only_fastq": (["--read1", "r1.fastq", "--umi_read", "read1", "--umi_offset", "0", "--umi_size", "8",
                          "--read1_offset", "8", "--outfile1", "o1.fastq.gz"], False),
    "read1_only_sam": (["--read1", "r1.fastq", "--umi_read", "read1", "--umi_offset", "2", "--umi_size", "6",
                        "--sam", "--outfile1", "-"], False),
    "paired_fastq": (["--read1", "r1.fastq", "--read2", "r2.fastq", "--cell_read", "read2", "--cell_offset", "0",
                      "--cell_size", "12", "--umi_read", "read1", "--umi_offset", "4", "--umi_size", "7",
                      "--read2_offset", "12", "--outfile1", "o1.fastq.gz", "--outfile2", "o2.fastq.gz"], False),
    "paired_sam": (["--read1", "r1.fastq", "--read2", "r2.fastq", "--cell_read", "read2", "--cell_offset", "0",
                    "--cell_size", "12", "--umi_read", "read2", "--umi_offset", "12", "--umi_size", "8", "--min_qual", "16",
                    "--phred_encoding", "33", "--sam", "--outfile1", "-"], False),
    "paired_plus_index_sam": (["--read1", "r1.fastq", "--read2", "r2.fastq", "--index1", "r1.fastq", "--sample_read",
                               "index1", "--sample_offset", "1", "--sample_size", "5", "--sam", "--10x", "--outfile1", "-"],
                              False),
    "four_files_fastq": (["--read1", "r1.fastq", "--read2", "r2.fastq", "--index1", "r2.fastq", "--index2", "r1.fastq",
                          "--umi_read", "index2", "--umi_offset", "0", "--umi_size", "9", "--cell_read", "index1",
                          "--cell_offset", "3", "--cell_size", "10", "--outfile1", "o1.fastq.gz", "--outfile2",
                          "o2.fastq.gz"], False),
}


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
@pytest.mark.parametrize("lds", [None, "4096", "65536"], ids=["default_tiles", "tiny_tiles", "large_tiles"])
@pytest.mark.parametrize("name", sorted(FILE_SETS))
def test_file_sets_and_tile_sizes_against_reference_binary(name, lds):
    args, _ = FILE_SETS[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    r1, r2 = make_long_mix(rng, 3000)
    env = {"FQGPU_CHUNK_MB": "1"}
    if lds:
        env["FQGPU_BC_LDS"] = lds
    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        res = []
        for d, binary, e in ((a, REF, None), (b, BIN, env)):
            for fn, img in (("r1.fastq", r1), ("r2.fastq", r2)):
                with open(os.path.join(d, fn), "wb") as f:
                    f.write(img)
            rc, out, err = run(binary, args, d, e)
            res.append((rc, out, strip_progress(err), gunzip_file(os.path.join(d, "o1.fastq.gz")),
                        gunzip_file(os.path.join(d, "o2.fastq.gz"))))
        assert res[0][0] == res[1][0] == 0, res[1][2]
        for i in range(1, 5):
            assert res[0][i] == res[1][i], (name, i)
        assert (res[0][1] or res[0][3])  # something was written


TILE_CASES = [i for i, c in enumerate(GOLDEN) if c["exit"] == 0][::3]
TILE_RUNS = SideBySide(golden_run, [(i, (("FQGPU_BC_LDS", lds),)) for lds in ("4096", "12288") for i in TILE_CASES])


@pytest.mark.parametrize("lds", ["4096", "12288"])
@pytest.mark.parametrize("i", TILE_CASES, ids=[" ".join(GOLDEN[i]["args"])[:50] for i in TILE_CASES])
def test_golden_invocations_other_tile_sizes(i, lds):
    golden_check(GOLDEN[i], TILE_RUNS.get((i, (("FQGPU_BC_LDS", lds),))), files_also_on_failure=True)


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
@pytest.mark.parametrize("variant", ["no_final_newline", "crlf", "empty_lines"])
@pytest.mark.parametrize("extra", [["--sam", "--outfile1", "-"], ["--outfile1", "o.fastq.gz"]], ids=["sam", "fastq"])
def test_odd_inputs_against_reference_binary(variant, extra):
    rng = np.random.default_rng(17)
    r1, r2 = make_10x(rng, 1500)
    if variant == "no_final_newline":
        r1, r2 = r1[:-1], r2[:-1]
    elif variant == "crlf":
        r1, r2 = r1.replace(b"\n", b"\r\n"), r2.replace(b"\n", b"\r\n")
    else:  # a last pair with empty sequence and quality lines (the barcode cannot be cut: "read too short")
        r1 += b"@SYN:1:FC:9:9:9:1500 1:N:0:ACGT\n\n+\n\n"
        r2 += b"@SYN:1:FC:9:9:9:1500 2:N:0:ACGT\n\n+\n\n"
    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        res = []
        for d, binary, env in ((a, REF, None), (b, BIN, {"FQGPU_CHUNK_MB": "1", "FQGPU_BC_LDS": "8192"})):
            for fn, img in (("r1.fastq", r1), ("r2.fastq", r2)):
                with open(os.path.join(d, fn), "wb") as f:
                    f.write(img)
            rc, out, err = run(binary, V2 + extra, d, env)
            res.append((rc, out, strip_progress(err), gunzip_file(os.path.join(d, "o.fastq.gz"))))
        assert res[0] == res[1], (variant, res[1][2][-300:])


# ---- several devices (FQGPU_DEVICES): blocks of the same B records of every input, taken by whichever context is free ----
SEVERAL = {"FQGPU_DEVICES": "0,0,0"}


def several_env(per_block):
    return tuple(sorted(dict(SEVERAL, **({"FQGPU_BLOCK_RECORDS": per_block} if per_block else {})).items()))


SEVERAL_RUNS = SideBySide(golden_run, [(i, several_env(pb)) for pb in ("3", "50", None) for i in range(len(GOLDEN))])


@pytest.mark.parametrize("per_block", ["3", "50", None], ids=["blocks_of_3", "blocks_of_50", "one_block"])
@pytest.mark.parametrize("i", range(len(GOLDEN)), ids=GOLDEN_IDS)
def test_several_devices_golden_invocations(i, per_block):
    golden_check(GOLDEN[i], SEVERAL_RUNS.get((i, several_env(per_block))))


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
@pytest.mark.parametrize("extra", [["--outfile1", "o.fastq.gz"], ["--sam", "--outfile1", "-"]], ids=["fastq", "sam"])
@pytest.mark.parametrize("shape", ["same_length", "index_file_shorter", "read_file_shorter", "index_file_cut", "read_file_cut_at_block",
                                   "mismatch_late", "wrong_header_late", "gz_inputs", "nul_record_start_read_file",
                                   "nul_record_start_index_file"])
def test_several_devices_against_reference_binary(shape, extra):
    """20 000 10x-style pairs in blocks of ~3 000 (1 MiB pieces) and of 1 000 records over three contexts: outputs, messages
    and exit codes of the reference program - also when one file is shorter, ends inside a record, ends exactly at a block
    boundary with a record cut, or the first finding lies many blocks into the files"""
    r1, r2 = ten_x_pairs_20000()
    l1, l2 = r1.split(b"\n"), r2.split(b"\n")
    if shape == "index_file_shorter":
        r1 = b"\n".join(l1[:4 * 15555]) + b"\n"
    elif shape == "read_file_shorter":
        r2 = b"\n".join(l2[:4 * 7000]) + b"\n"
    elif shape == "index_file_cut":
        r1 = b"\n".join(l1[:4 * 12001 + 2]) + b"\n"
    elif shape == "read_file_cut_at_block":
        r2 = b"\n".join(l2[:4 * 5000 + 1]) + b"\n"
    elif shape == "mismatch_late":
        l2[4 * 17123] = l2[4 * 17123].replace(b"SYN:", b"SYX:")
        r2 = b"\n".join(l2)
    elif shape == "wrong_header_late":
        l1[4 * 9044] = b"#" + l1[4 * 9044][1:]
        r1 = b"\n".join(l1)
    elif shape == "nul_record_start_read_file":  # "no entry" for the reference (src/fastq.c:250): the loop ends there, cleanly
        l2[4 * 13007] = b"\0" + l2[4 * 13007][1:]
        r2 = b"\n".join(l2)
    elif shape == "nul_record_start_index_file":
        l1[4 * 2999] = b"\0" + l1[4 * 2999][1:]
        r1 = b"\n".join(l1)
    names = ("r1.fastq", "r2.fastq")
    args = list(V2)
    if shape == "gz_inputs":
        names = ("r1.fastq.gz", "r2.fastq.gz")
        args = [a + ".gz" if a.endswith(".fastq") else a for a in args]
    envs = [dict(SEVERAL, FQGPU_CHUNK_MB="1"), dict(SEVERAL, FQGPU_BLOCK_RECORDS="1000")]
    if shape.startswith("nul_"):
        envs += [None, {"FQGPU_CHUNK_MB": "1"}]  # the one-device loop, whole and in 1 MiB pieces: the same answer
    def one(job):
        binary, env = job
        with tempfile.TemporaryDirectory() as d:
            for fn, img in zip(names, (r1, r2)):
                with open(os.path.join(d, fn), "wb") as f:
                    f.write(gzip.compress(img, 1) if fn.endswith(".gz") else img)
            rc, out, err = run(binary, args + extra, d, env)
            return rc, out, strip_progress(err), gunzip_file(os.path.join(d, "o.fastq.gz")) if rc == 0 else None

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(6) as ex:  # the reference and the program's runs side by side
        res = list(ex.map(one, [(REF, None)] + [(BIN, e) for e in envs]))
    for got in res[1:]:
        assert got[0] == res[0][0], got[2][-400:]
        assert got[2] == res[0][2]
        assert got[1] == res[0][1]
        if res[0][0] == 0:  # (after a finding the reference leaves its gz stream unfinished)
            assert got[3] == res[0][3]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", sorted(FILE_SETS))
def test_several_devices_file_sets(name):
    args, _ = FILE_SETS[name]
    rng = np.random.default_rng(sum(map(ord, name)) + 1)
    r1, r2 = make_long_mix(rng, 3000)
    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        res = []
        for d, binary, e in ((a, REF, None), (b, BIN, dict(SEVERAL, FQGPU_BLOCK_RECORDS="211"))):
            for fn, img in (("r1.fastq", r1), ("r2.fastq", r2)):
                with open(os.path.join(d, fn), "wb") as f:
                    f.write(img)
            rc, out, err = run(binary, args, d, e)
            res.append((rc, out, strip_progress(err), gunzip_file(os.path.join(d, "o1.fastq.gz")),
                        gunzip_file(os.path.join(d, "o2.fastq.gz"))))
        assert res[0][0] == res[1][0] == 0, res[1][2]
        for i in range(1, 5):
            assert res[0][i] == res[1][i], (name, i)


@pytest.mark.parametrize("extra", [["--sam", "--outfile1", "-"], ["--sam", "--10x", "--outfile1", "-"], ["--outfile1", "o.fastq.gz"]],
                         ids=["sam", "sam10x", "fastq"])
@pytest.mark.parametrize("env", [None, {"FQGPU_BC_LDS": "4096"}, {"FQGPU_DEVICES": "0,0", "FQGPU_BLOCK_RECORDS": "7"}],
                         ids=["default", "tiny_tiles", "devices"])
def test_quality_line_shorter_than_the_barcode_range(extra, env):
    """a record of the barcode file whose quality line ends before offset + size (a damaged file: '+' twice, a quality
    line cut short): the reference copies the quality characters with strncpy from a buffer that still holds the bytes of
    EARLIER records behind the string's end; here and in the oracle the string ends where the line does
    (tools/fuzz_campaign_programs.py, seed 20226: the kernel used to print whatever LDS held there)"""
    rng = np.random.default_rng(20226)
    r1, r2 = make_10x(rng, 300, 0.0, 0.0)
    l1 = r1.split(b"\n")
    l1[4 * 40 + 3] = b"+"                    # the quality line of record 40 is one character
    l1[4 * 90 + 3] = l1[4 * 90 + 3][:18]     # ends inside the UMI range
    l1[4 * 120 + 3] = l1[4 * 120 + 3][:16]   # ends where the UMI range begins
    l1[4 * 150 + 3] = b""                    # empty
    r1 = b"\n".join(l1)
    files = {"r1.fastq": r1, "r2.fastq": r2}
    args = V2[:V2.index("--min_qual")] + ["--min_qual", "0"] + extra
    with tempfile.TemporaryDirectory() as d:
        for name, img in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(img)
        rc, out, err = run(BIN, args, d, env)
        want = pbo.run_pre_barcodes(args, lambda n: files[n])
        assert rc == want["exit"] == 0, err[-300:]
        assert out == want["stdout"]
        assert strip_progress(err) == strip_progress(want["stderr"])
        if "o.fastq.gz" in args:
            assert gunzip_file(os.path.join(d, "o.fastq.gz")) == want["files"][1].decode("latin-1")


# ---- the name checks (made by the emit kernels): what the word-wise agreement test accepts, and what it hands over to
# the exact comparison ----
def _pairs_with_names(rng, names):
    """two files of 26 bp / 60 bp reads whose k-th headers are names[k] = (header of file 1, header of file 2)"""
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    f1, f2 = [], []
    for h1, h2 in names:
        s1 = bases[rng.integers(0, 4, 26)].tobytes()
        s2 = bases[rng.integers(0, 4, 60)].tobytes()
        q1 = (rng.integers(20, 41, 26) + 33).astype(np.uint8).tobytes()
        q2 = (rng.integers(20, 41, 60) + 33).astype(np.uint8).tobytes()
        f1.append(b"@" + h1 + b"\n" + s1 + b"\n+\n" + q1 + b"\n")
        f2.append(b"@" + h2 + b"\n" + s2 + b"\n+\n" + q2 + b"\n")
    return b"".join(f1), b"".join(f2)


def _casava_names(n):
    """names of every length 1..70 (differences and blanks at every place of an 8-byte word), all agreeing"""
    out = []
    for i in range(n):
        stem = (b"N%dx" % i) + b"ABCDEFGHIJ" * 8
        stem = stem[: 1 + i % 70]
        kind = i % 7
        if kind == 0:
            out.append((stem + b" 1:N:0:AC", stem + b" 2:N:0:AC"))           # the usual pair
        elif kind == 1:
            out.append((stem + b" 1:N:0:AC", stem + b" 1:N:0:AC"))           # the same line
        elif kind == 2:
            out.append((stem + b" 1:N:0:AC", stem + b" 1:Y:0:AC"))           # they differ behind the first field
        elif kind == 3:
            out.append((stem + b" 1:N:0:AC", stem + b" 2"))                  # a short second header
        elif kind == 4:
            out.append((stem + b"/1 1:N:0:AC", stem + b"/2 2:N:0:AC"))       # "/x" in front of the blank is dropped
        elif kind == 5:
            out.append((stem + b" z 1:N:0", stem + b" y 1:N:0"))             # a blank, then a difference
        else:
            out.append((stem + b" 1:N:0:AC", stem + b"  2:N:0:AC"))          # two blanks
    return out


@pytest.mark.parametrize("env", [None, {"FQGPU_BC_LDS": "4096"}], ids=["default", "tiny_tiles"])
@pytest.mark.parametrize("extra", [["--sam", "--outfile1", "-"], ["--outfile1", "o.fastq.gz"]], ids=["sam", "fastq"])
def test_names_that_agree_in_every_shape(extra, env):
    rng = np.random.default_rng(77)
    names = _casava_names(1400)
    r2, r1 = _pairs_with_names(rng, names)  # (file of --read1 first)
    files = {"r1.fastq": r1, "r2.fastq": r2}
    args = V2 + extra
    with tempfile.TemporaryDirectory() as d:
        for name, img in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(img)
        rc, out, err = run(BIN, args, d, env)
        want = pbo.run_pre_barcodes(args, lambda n: files[n])
        assert rc == want["exit"] == 0, err[-300:]
        assert out == want["stdout"]
        assert strip_progress(err) == strip_progress(want["stderr"])
        if "o.fastq.gz" in args:
            assert gunzip_file(os.path.join(d, "o.fastq.gz")) == want["files"][1].decode("latin-1")


MISMATCHES = {
    "last_character_of_the_name": (b"RUN:7:12345 1:N:0:AC", b"RUN:7:12346 2:N:0:AC"),
    "first_character": (b"RUN:7:12345 1:N:0:AC", b"SUN:7:12345 2:N:0:AC"),
    "one_is_a_prefix": (b"RUN:7:12345 1:N:0:AC", b"RUN:7:1234 1:N:0:AC"),
    "longer_second_name": (b"RUN:7:12345 1:N:0:AC", b"RUN:7:123456 1:N:0:AC"),
    "ninth_character": (b"RUN:7:12345:AAAA 1:N:0:AC", b"RUN:7:12X45:AAAA 1:N:0:AC"),
    "slash_suffix_differs_in_the_name": (b"RUN:7:12345/1 1:N:0:AC", b"RUN:7:12355/2 2:N:0:AC"),
    "no_blank_in_either": (b"RUN:7:12345", b"RUN:7:12346"),
    # (a header without a blank keeps its '\n' in the name, src/fastq.c:502-511: it never equals one with a blank)
    "second_header_ends_with_the_name": (b"RUN:7:12345 1:N:0:AC", b"RUN:7:12345"),
    "wrong_header_second_file": (b"RUN:7:12345 1:N:0:AC", None),
}


@pytest.mark.parametrize("env", [None, {"FQGPU_BC_LDS": "4096"}], ids=["default", "tiny_tiles"])
@pytest.mark.parametrize("shape", sorted(MISMATCHES))
def test_names_that_do_not_agree(shape, env):
    """the first iteration whose names differ stops the program, discards in front of it are counted, what was printed
    in front of it is printed"""
    rng = np.random.default_rng(sum(map(ord, shape)))
    names = _casava_names(700)
    names[433] = MISMATCHES[shape] if MISMATCHES[shape][1] is not None else (MISMATCHES[shape][0], MISMATCHES[shape][0])
    names[600] = (b"LATER:1 1:N:0:AC", b"LATER:2 1:N:0:AC")  # a later one does not matter
    r2, r1 = _pairs_with_names(rng, names)
    if MISMATCHES[shape][1] is None:
        l1 = r1.split(b"\n")
        l1[4 * 433] = b"RUN:7:12345 1:N:0:AC"  # no '@'
        r1 = b"\n".join(l1)
    # a low base quality in a barcode in front of the finding (a discard) and behind it
    for rec in (100, 500):
        l1 = r1.split(b"\n")
        l1[4 * rec + 3] = b"!" + l1[4 * rec + 3][1:]
        r1 = b"\n".join(l1)
    files = {"r1.fastq": r1, "r2.fastq": r2}
    args = V2 + ["--sam", "--outfile1", "-"]
    with tempfile.TemporaryDirectory() as d:
        for name, img in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(img)
        rc, out, err = run(BIN, args, d, env)
        want = pbo.run_pre_barcodes(args, lambda n: files[n])
        assert rc == want["exit"] == 3, err[-300:]
        assert out == want["stdout"]
        assert strip_progress(err) == strip_progress(want["stderr"])


@pytest.mark.parametrize("env", [None, {"FQGPU_DEVICES": "0,0", "FQGPU_BLOCK_RECORDS": "3"}], ids=["default", "devices"])
@pytest.mark.parametrize("open_end", [True, False], ids=["no_last_newline", "last_newline"])
@pytest.mark.parametrize("which", ["r1", "r2"])
def test_loop_condition_ends_the_loop_before_a_truncated_record_is_read(which, open_end, env):
    """fastq_files_eof (src/fastq_pre_barcodes.c:288-297, :594): an input whose last line has no newline is at its end for
    gzeof once that line is read, so the incomplete record that follows in ANOTHER input is never read; with the newline
    the next iteration reads it and the file is truncated (tools/fuzz_campaign_programs.py, seed 930035)"""
    rng = np.random.default_rng(930035)
    r1, r2 = make_10x(rng, 7, 0.0, 0.0)
    extra = b"\n".join(r2.split(b"\n")[:2]) + b"\n"   # two lines of a record that never ends
    if which == "r1":
        short, long_ = r1, r2 + extra
    else:
        short, long_ = r2, r1 + b"\n".join(r1.split(b"\n")[:2]) + b"\n"
    if open_end:
        short = short[:-1]
    files = {"r1.fastq": short if which == "r1" else long_, "r2.fastq": long_ if which == "r1" else short}
    args = V2 + ["--sam", "--outfile1", "-"]
    with tempfile.TemporaryDirectory() as d:
        for name, img in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(img)
        rc, out, err = run(BIN, args, d, env)
        want = pbo.run_pre_barcodes(args, lambda n: files[n])
        assert rc == want["exit"], (err[-300:], want["stderr"][-300:])
        assert out == want["stdout"]
        assert strip_progress(err) == strip_progress(want["stderr"])


def test_fast_gzip_output_inflates_to_the_same_text():
    """FQGPU_GZIP_FAST=1 (host/fq_fastdeflate.h instead of zlib for the output members): another compressed file, the
    same text in it"""
    rng = np.random.default_rng(31)
    from tests import fuzz

    with tempfile.TemporaryDirectory() as tmp:
        n = 20000
        r1 = fuzz.make_fastq(np.random.default_rng(7), n, 60, 100, "casava", mate=1)
        i1 = fuzz.make_fastq(np.random.default_rng(7), n, 26, 26, "casava", mate=2)
        for name, img in (("r1.fastq", r1), ("i1.fastq", i1)):
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(img)
        args = ["--read1", "r1.fastq", "--index1", "i1.fastq", "--umi_read", "index1", "--umi_offset", "16", "--umi_size", "10",
                "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16", "--phred_encoding", "33", "--min_qual", "0"]
        texts, sizes = {}, {}
        for tag, env in (("zlib", {}), ("fast", {"FQGPU_GZIP_FAST": "1"})):
            out = "out_%s.fastq.gz" % tag
            rc, so, se = run(BIN, args + ["--outfile1", out], tmp, env)
            assert rc == 0, se[-500:]
            texts[tag] = gunzip_file(os.path.join(tmp, out))
            sizes[tag] = os.path.getsize(os.path.join(tmp, out))
        assert texts["fast"] == texts["zlib"] and len(texts["zlib"]) > 1000000
        assert sizes["fast"] != sizes["zlib"] and sizes["fast"] < 1.3 * sizes["zlib"]


def test_json_metrics_twin_of_the_summary():
    """FQGPU_JSON_METRICS (an extra): the machine-readable twin of "Reads processed / discarded" in a file of
    its own - stdout and stderr stay the reference's (compared with the oracle here)"""
    rng = np.random.default_rng(9)
    r1, r2 = make_10x(rng, 4000)
    with tempfile.TemporaryDirectory() as d:
        files = {"r1.fastq": r1, "r2.fastq": r2}
        for name, img in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(img)
        args = V2 + ["--sam", "--outfile1", "-"]
        jm = os.path.join(d, "m.json")
        rc, out, err = run(BIN, args, d, {"FQGPU_JSON_METRICS": jm})
        want = pbo.run_pre_barcodes(args, lambda n: files[n])
        assert rc == want["exit"] == 0 and out == want["stdout"] and strip_progress(err) == strip_progress(want["stderr"])
        m = json.load(open(jm))
        assert m["program"] == "fastq_pre_barcodes" and m["reads"] == 4000 and m["devices"] == 1
        assert "INFO:Reads discarded: %d" % m["discarded"] in err and m["seconds"] > 0 and m["input_bytes"] >= len(r1) + len(r2)
