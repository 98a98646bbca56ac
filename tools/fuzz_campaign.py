#!/usr/bin/env python3
"""A differential campaign with fresh seeds: bin/fastq_info against the REFERENCE BINARY (oracle/_ref/fastq_info) on
seeded, mutated files - all four modes, one piece / 1 MiB pieces / several contexts, the streaming framing forced on
small images.  Not a test: run it on the GPU box (`python tools/fuzz_campaign.py <seed> <cases> [workers]`), it prints
every case that differs and a summary line.  Cases that differ become seeded tests."""
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

BIN = os.path.join(REPO, "bin", "fastq_info")
REF = os.path.join(REPO, "oracle", "_ref", "fastq_info")
ENVS = [("default", {}), ("pieces", {"FQGPU_CHUNK_MB": "1"}), ("stream", {"FQGPU_STREAM_MIN": "256"}),
        ("devices", {"FQGPU_DEVICES": "0,0", "FQGPU_CHUNK_MB": "1"}),
        ("devices_stream", {"FQGPU_DEVICES": "0,0,0", "FQGPU_CHUNK_MB": "1", "FQGPU_STREAM_MIN": "256"}),
        # (.gz inputs through the many-core gzip reader, host/fq_pgzip.h, in chunks of 8 KiB - files this small are one zlib thread's otherwise)
        ("gz_chunks", {"FQGPU_PGZIP_MIN": "0", "FQGPU_PGZIP_CHUNK": "8192", "FQGPU_HOST_THREADS": "3", "FQGPU_CHUNK_MB": "1"})]
if os.environ.get("CAMPAIGN_COMPAT") == "1":
    # the zero-change route: the reference's own fastq_info.c linked against libfastq_gpu.so (oracle/Makefile)
    BIN = os.path.join(REPO, "oracle", "_ref", "fastq_info_on_libfastq_gpu")
    ENVS = [("compat", {}), ("compat", {}), ("compat_stream", {"FQGPU_STREAM_MIN": "256"}), ("compat", {}), ("compat_stream", {"FQGPU_STREAM_MIN": "256"})]


def run(binary, args, cwd, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    try:
        p = subprocess.run(["fastq_info"] + args, executable=binary, cwd=cwd, capture_output=True, timeout=120, env=e)
    except subprocess.TimeoutExpired:
        return ("timeout", "", "")
    return (p.returncode, p.stdout.decode("latin-1"), strip_progress(p.stderr.decode("latin-1")))


def one_case(seed):
    rng = np.random.default_rng(seed)
    style = ["casava", "slash", "int", "nosuffix"][int(rng.integers(0, 4))]
    big = rng.random() < 0.35  # several 1 MiB pieces
    n = int(rng.integers(9000, 16000)) if big else int(rng.integers(1, 400))
    lo, hi = (30, 150) if big else (1, 120)
    hdr2 = bool(rng.random() < 0.3)
    crlf = bool(rng.random() < 0.1)
    rna = bool(rng.random() < 0.1)
    sub = int(rng.integers(0, 1 << 30))
    a = fuzz.make_fastq(np.random.default_rng(sub), n, lo, hi, style, hdr2_names=hdr2, crlf=crlf, rna=rna, mate=1)
    b = fuzz.make_fastq(np.random.default_rng(sub), n, lo, hi, style, hdr2_names=hdr2, crlf=crlf, rna=rna, mate=2)
    what = []
    for which in ("a", "b"):
        k = int(rng.integers(0, 4))  # 0-2 mutations per file, mostly none in the second
        if which == "b" and rng.random() < 0.6:
            k = 0
        for _ in range(min(k, 2)):
            kind = fuzz.MUTATIONS[int(rng.integers(0, len(fuzz.MUTATIONS)))]
            what.append(which + ":" + kind)
            if which == "a":
                a = fuzz.mutate(rng, a, kind)
            else:
                b = fuzz.mutate(rng, b, kind)
    # pairing shapes
    shape = int(rng.integers(0, 6))
    lb = b.split(b"\n")
    nb = len(lb) // 4
    if shape == 1 and nb > 2:  # shuffled, still paired
        recs = [lb[4 * i:4 * i + 4] for i in range(nb)]
        order = rng.permutation(nb)
        lb = [x for i in order for x in recs[i]] + lb[4 * nb:]
        b = b"\n".join(lb)
        what.append("b:shuffled")
    elif shape == 2 and nb > 2:  # a record dropped from file 2
        k = int(rng.integers(0, nb))
        b = b"\n".join(lb[:4 * k] + lb[4 * k + 4:])
        what.append("b:dropped")
    elif shape == 3 and nb > 2:  # a record twice in file 2
        k, at = int(rng.integers(0, nb)), int(rng.integers(0, nb))
        b = b"\n".join(lb[:4 * at] + lb[4 * k:4 * k + 4] + lb[4 * at:])
        what.append("b:twice")
    elif shape == 4:  # a record twice in file 1
        la = a.split(b"\n")
        na = len(la) // 4
        if na > 2:
            k, at = int(rng.integers(0, na)), int(rng.integers(0, na))
            a = b"\n".join(la[:4 * at] + la[4 * k:4 * k + 4] + la[4 * at:])
            what.append("a:twice")
    # an interleaved file from the (damaged) two: mates alternate as long as both have records
    la, lb2 = a.split(b"\n"), b.split(b"\n")
    ni = min(len(la), len(lb2)) // 4
    inter = b"".join(b"\n".join(la[4 * i:4 * i + 4]) + b"\n" + b"\n".join(lb2[4 * i:4 * i + 4]) + b"\n" for i in range(ni))
    if rng.random() < 0.3 and ni:
        kind = fuzz.MUTATIONS[int(rng.integers(0, len(fuzz.MUTATIONS)))]
        what.append("i:" + kind)
        inter = fuzz.mutate(rng, inter, kind)
    bad = []
    with tempfile.TemporaryDirectory() as ref_dir, tempfile.TemporaryDirectory() as gpu_dir:
        for d in (ref_dir, gpu_dir):
            for name, img in (("a.fastq", a), ("b.fastq", b), ("i.fastq", inter)):
                with open(os.path.join(d, name), "wb") as f:
                    f.write(img)
            for name, img, level in (("a.fastq.gz", a, 1), ("b.fastq.gz", b, 6), ("i.fastq.gz", inter, 9)):
                with open(os.path.join(d, name), "wb") as f:
                    f.write(gzip.compress(img, level))
        modes = [["-r", "a.fastq"], ["a.fastq"], ["i.fastq", "pe"], ["a.fastq", "b.fastq"], ["-s", "a.fastq", "b.fastq"],
                 ["a.fastq.gz", "b.fastq"], ["a.fastq.gz", "b.fastq.gz"], ["i.fastq.gz", "pe"]]
        if not big:
            modes += [["-r", "-s", "a.fastq", "b.fastq"], ["b.fastq", "a.fastq"], ["a.fastq", "pe"], ["-r", "a.fastq.gz"],
                      ["-e", "-q", "a.fastq"], ["-f", "a.fastq"]]
        envs = ENVS if big else [ENVS[0], ENVS[2], ENVS[4], ENVS[5]]
        if os.environ.get("CAMPAIGN_COMPAT") == "1":
            envs = [ENVS[0], ENVS[2]]
        for args in modes:
            want = run(REF, args, ref_dir)
            if want[0] in ("timeout", -11, -6):  # the reference itself crashed: not a case
                continue
            for tag, env in envs:
                got = run(BIN, args, gpu_dir, env)
                if got != want:
                    bad.append((seed, tag, args, what, want[0], got[0], want[2][-300:], got[2][-300:]))
    return bad


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    n_bad = 0
    with ThreadPoolExecutor(workers) as ex:
        for bad in ex.map(one_case, range(seed0, seed0 + cases)):
            for b in bad:
                n_bad += 1
                if n_bad <= 40:
                    print("DIFF", b, flush=True)
    print(f"campaign seeds {seed0}..{seed0 + cases - 1}: {n_bad} differing runs", flush=True)


if __name__ == "__main__":
    main()
