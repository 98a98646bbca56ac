#!/usr/bin/env python3
"""Differential campaign, fresh seeds, for the other drop-in programs against the REFERENCE BINARIES (oracle/_ref):
fastq_filterpair, fastq_filter_n, fastq_trim_poly_at, fastq_pre_barcodes (one device and FQGPU_DEVICES) on seeded,
lightly damaged files.  `python tools/fuzz_campaign_programs.py <seed> <cases> [workers]` on the GPU box; prints the
cases that differ and a summary line.  (Inputs on which the reference itself dies of a signal are skipped; DESIGN 7.1
lists the inputs on which it reads memory it does not own.)"""
import gzip
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import fuzz  # noqa: E402
from tests.util import strip_progress  # noqa: E402

KINDS = ["drop_line", "dup_line", "truncate", "strip_last_nl", "del_byte", "empty_seq", "flip_seq", "bad_at", "empty_hdr", "ins_cr"]


def binaries(name):
    return os.path.join(REPO, "oracle", "_ref", name), os.path.join(REPO, "bin", name)


def run(binary, argv0, args, cwd, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    try:
        p = subprocess.run([argv0] + args, executable=binary, cwd=cwd, capture_output=True, timeout=120, env=e)
    except subprocess.TimeoutExpired:
        return ("timeout", b"", "")
    return (p.returncode, p.stdout, strip_progress(p.stderr.decode("latin-1")))


def gunzip(path):
    if not os.path.exists(path):
        return None
    raw = open(path, "rb").read()
    try:
        return gzip.decompress(raw) if raw else b""
    except Exception:
        return b"<broken gzip>"


def damaged(rng, img, p_none, kinds=None):
    kinds = kinds or KINDS
    what = []
    if rng.random() >= p_none:
        for _ in range(int(rng.integers(1, 3))):
            kind = kinds[int(rng.integers(0, len(kinds)))]
            what.append(kind)
            img = fuzz.mutate(rng, img, kind)
    return img, what


def in_pieces(rng, img, p):
    """Round 5: with probability p one more kind of damage - a NUL byte somewhere, or a header line beyond the gzgets
    limit (1000 - 3000 bytes: the reference reads it in pieces and every later line is out of step).  The programs
    reproduce both since round 5: a line is a C string."""
    if rng.random() >= p or not img:
        return img, []
    if rng.random() < 0.5:
        return fuzz.mutate(rng, img, "ins_nul"), ["ins_nul"]
    lines = img.split(b"\n")
    k = 4 * int(rng.integers(0, max(1, len(lines) // 4)))
    if k < len(lines) and lines[k]:
        lines[k] = lines[k][:1] + b"L" * int(rng.integers(1000, 3000)) + lines[k][1:]
    return b"\n".join(lines), ["long_hdr"]


def compare(name, args, files, outs, envs, seed, what):
    """run reference and product in fresh directories; outs: output files to compare when the exit status is 0"""
    ref, prod = binaries(name)
    bad = []
    with tempfile.TemporaryDirectory() as d:
        for fn, img in files.items():
            with open(os.path.join(d, fn), "wb") as f:
                f.write(img)
        want = run(ref, name, args, d)
        want_files = [gunzip(os.path.join(d, o)) if o.endswith(".gz") else None for o in outs]
    if want[0] == "timeout" or (isinstance(want[0], int) and want[0] < 0):
        return bad
    for tag, env in envs:
        env = dict(env)
        gz_inputs = env.pop("_gz_inputs", None)  # the product reads the same inputs gzip'd (under the same names: gzip is known by its magic)
        with tempfile.TemporaryDirectory() as d:
            for fn, img in files.items():
                with open(os.path.join(d, fn), "wb") as f:
                    f.write(gzip.compress(img, int(gz_inputs)) if gz_inputs else img)
            got = run(prod, name, args, d, env)
            got_files = [gunzip(os.path.join(d, o)) if o.endswith(".gz") else None for o in outs]
        same = got == want and (want[0] != 0 or got_files == want_files)
        if not same:
            bad.append((seed, name, tag, args, what, want[0], got[0], want[2][-250:], got[2][-250:],
                        [None if x is None else len(x) for x in want_files], [None if x is None else len(x) for x in got_files]))
    return bad


def one_case(seed):
    rng = np.random.default_rng(seed)
    bad = []
    big = rng.random() < 0.3
    n = int(rng.integers(8000, 14000)) if big else int(rng.integers(1, 300))
    style = ["casava", "slash", "int", "nosuffix"][int(rng.integers(0, 4))]
    sub = int(rng.integers(0, 1 << 30))
    pieces = [("default", {}), ("pieces", {"FQGPU_CHUNK_MB": "1"})] if big else [("default", {}), ("tiny_tiles", {"FQGPU_BC_LDS": "4096"})]
    # gzip'd inputs through the many-core gzip reader in chunks of 8 KiB, gzip output from the fast compressor (round 4)
    pieces.append(("gz_in_chunks_fast_gz_out", {"_gz_inputs": str(int(rng.integers(1, 10))), "FQGPU_PGZIP_MIN": "0", "FQGPU_PGZIP_CHUNK": "8192",
                                                  "FQGPU_HOST_THREADS": "3", "FQGPU_GZIP_FAST": "1", "FQGPU_CHUNK_MB": "1"}))
    # ---- fastq_filterpair
    a = fuzz.make_fastq(np.random.default_rng(sub), n, 20, 120, style, mate=1)
    b = fuzz.make_fastq(np.random.default_rng(sub), n, 20, 120, style, mate=2)
    la, lb = a.split(b"\n"), b.split(b"\n")
    keep_a = rng.random(n) < 0.9
    keep_b = rng.random(n) < 0.9
    order = rng.permutation(n) if rng.random() < 0.3 else np.arange(n)
    a2 = b"".join(b"\n".join(la[4 * i:4 * i + 4]) + b"\n" for i in range(n) if keep_a[i])
    b2 = b"".join(b"\n".join(lb[4 * i:4 * i + 4]) + b"\n" for i in order if keep_b[i])
    a2, w1 = damaged(rng, a2, 0.7)
    b2, w2 = damaged(rng, b2, 0.7)
    a2, x1 = in_pieces(rng, a2, 0.15)
    b2, x2 = in_pieces(rng, b2, 0.15)
    w1, w2 = w1 + x1, w2 + x2
    args = ["a.fastq", "b.fastq", "p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"]
    if rng.random() < 0.25:
        args.append("sorted")
    bad += compare("fastq_filterpair", args, {"a.fastq": a2, "b.fastq": b2}, ["p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"], pieces, seed, w1 + w2)
    # ---- fastq_filter_n / fastq_trim_poly_at on reads with N runs and poly-A/T ends
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 160))
        s = bytearray(bases[rng.choice(5, L, p=[0.24, 0.24, 0.24, 0.24, 0.04])].tobytes())
        r = rng.random()
        if r < 0.2:
            k = int(rng.integers(1, L + 1))
            s[L - k:] = b"A" * k
        elif r < 0.4:
            k = int(rng.integers(1, L + 1))
            s[:k] = b"T" * k
        q = (rng.integers(2, 41, L) + 33).astype(np.uint8).tobytes()
        recs.append(b"@r%d x\n" % i + bytes(s) + b"\n+\n" + q + b"\n")
    img, w = damaged(rng, b"".join(recs), 0.6)
    nflag = [[], ["-n", str(int(rng.integers(0, 60)))]][int(rng.integers(0, 2))]
    img_n, xn = in_pieces(rng, img, 0.25)
    bad += compare("fastq_filter_n", nflag + ["in.fastq"], {"in.fastq": img_n}, [], pieces, seed, w + xn)
    if not any(k in w for k in ("del_byte", "truncate", "drop_line", "dup_line", "ins_cr", "empty_hdr")):
        # (lines out of step make the reference print bytes of earlier records: DESIGN 7.1)
        tflags = ["--min_poly_at_len", str(int(rng.integers(1, 30))), "--min_len", str(int(rng.integers(0, 80)))]
        bad += compare("fastq_trim_poly_at", ["--file", "in.fastq", "--outfile", "o.fastq.gz"] + tflags, {"in.fastq": img}, ["o.fastq.gz"], pieces, seed, w)
    # ---- fastq_pre_barcodes, 10x-style
    r1, r2 = [], []
    for i in range(n):
        name = b"SYN:1:FC:%d:%d:%d:%d" % (i % 8 + 1, i % 97, i % 1013, i)
        l1 = 26 if rng.random() > 0.02 else int(rng.integers(1, 26))
        s1 = bases[rng.integers(0, 4, l1)].tobytes()
        q1 = (rng.integers(5, 41, l1) + 33).astype(np.uint8).tobytes()
        l2 = int(rng.integers(30, 151))
        s2 = bases[rng.integers(0, 5, l2)].tobytes()
        q2 = (rng.integers(2, 41, l2) + 33).astype(np.uint8).tobytes()
        r1.append(b"@" + name + b" 1:N:0:ACGT\n" + s1 + b"\n+\n" + q1 + b"\n")
        r2.append(b"@" + name + b" 2:N:0:ACGT\n" + s2 + b"\n+\n" + q2 + b"\n")
    # (the barcode file keeps its lines in step: a quality line shorter than the barcode range makes the reference copy
    #  bytes of earlier records that are still in its buffer - DESIGN 7.1; seed 20226 of an earlier version found it)
    i1, w1 = damaged(rng, b"".join(r1), 0.75, ["flip_seq", "bad_at", "empty_hdr", "strip_last_nl", "empty_seq"])
    i2, w2 = damaged(rng, b"".join(r2), 0.75)
    i2, x2 = in_pieces(rng, i2, 0.2)
    w2 = w2 + x2
    if rng.random() < 0.2:
        k = int(rng.integers(0, n + 1))
        i2 = b"".join(r2[:k])
        w2 = w2 + ["shorter"]
    v2 = ["--read1", "r2.fastq", "--index1", "r1.fastq", "--umi_read", "index1", "--umi_offset", "16", "--umi_size", "10",
          "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16", "--phred_encoding", "33", "--min_qual", str(int(rng.integers(0, 20)))]
    out = [["--outfile1", "o.fastq.gz"], ["--sam", "--outfile1", "-"], ["--sam", "--10x", "--outfile1", "-"]][int(rng.integers(0, 3))]
    envs = pieces + [("devices", {"FQGPU_DEVICES": "0,0,0", "FQGPU_BLOCK_RECORDS": str(int(rng.integers(1, 2000)))})]
    if not any(k in (w1 + w2) for k in ("ins_cr",)):
        bad += compare("fastq_pre_barcodes", v2 + out, {"r1.fastq": i1, "r2.fastq": i2}, ["o.fastq.gz"], envs, seed, w1 + w2)
    return bad


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    n_bad = 0
    with ThreadPoolExecutor(workers) as ex:
        for bad in ex.map(one_case, range(seed0, seed0 + cases)):
            for b in bad:
                n_bad += 1
                if n_bad <= 40:
                    print("DIFF", b, flush=True)
    print(f"campaign (programs) seeds {seed0}..{seed0 + cases - 1}: {n_bad} differing runs", flush=True)


if __name__ == "__main__":
    main()
