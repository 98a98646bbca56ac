#!/usr/bin/env python3
"""Wall time of the drop-in programs with one context against FQGPU_DEVICES with two and three contexts (on a one-GPU
box: the same GPU several times - it cannot be faster there, it must not be slower).  Legs: fastq_info -r, fastq_info
(default mode), fastq_info on a pair, fastq_pre_barcodes --sam.  Files of N million 150 bp reads (and their 26 bp index
reads) in /dev/shm.  usage: tools/multi_dev_legs.py [million reads] [repetitions] [part of a leg's name]"""
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ONLY = sys.argv[3] if len(sys.argv) > 3 else ""   # legs whose label holds this
D = "/dev/shm/fqg_multi_legs"
os.makedirs(D, exist_ok=True)


def write(path, n, read_len, seed, qlo=35):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        done = 0
        while done < n:
            m = min(1_000_000, n - done)
            w = 12 + 1 + read_len + 1 + 2 + read_len + 1
            rec = np.empty((m, w), dtype=np.uint8)
            names = np.char.zfill(np.arange(done, done + m).astype("U"), 10)
            rec[:, 0] = ord("@")
            rec[:, 1] = ord("r")
            rec[:, 2:12] = np.frombuffer("".join(names).encode(), dtype=np.uint8).reshape(m, 10)
            rec[:, 12] = 10
            rec[:, 13:13 + read_len] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (m, read_len))
            rec[:, 13 + read_len] = 10
            rec[:, 14 + read_len] = ord("+")
            rec[:, 15 + read_len] = 10
            rec[:, 16 + read_len:16 + 2 * read_len] = rng.integers(qlo, 74, (m, read_len), dtype=np.uint8)
            rec[:, 16 + 2 * read_len] = 10
            f.write(rec.tobytes())
            done += m


n = N * 1_000_000
write(f"{D}/a_1.fastq", n, 150, 1)
write(f"{D}/a_2.fastq", n, 150, 2)
write(f"{D}/i1.fastq", n, 26, 3, qlo=40)   # (a few index reads fall below --min_qual 10 + 33: the discards)
V2 = ["--read1", "a_1.fastq", "--index1", "i1.fastq", "--umi_read", "index1", "--umi_offset", "16", "--umi_size", "10",
      "--cell_read", "index1", "--cell_offset", "0", "--cell_size", "16", "--phred_encoding", "33", "--min_qual", "10"]
LEGS = [("fastq_info -r", ["fastq_info", "-r", "a_1.fastq"]),
        ("fastq_info (default)", ["fastq_info", "a_1.fastq"]),
        ("fastq_info pair", ["fastq_info", "a_1.fastq", "a_2.fastq"]),
        ("fastq_pre_barcodes --sam", ["fastq_pre_barcodes"] + V2 + ["--sam", "--outfile1", "-"])]
print(f"{N} M reads per file; best of {REPS} runs, seconds of wall time from process start to exit")
for label, cmd in LEGS:
    if ONLY not in label:
        continue
    exe = os.path.join(REPO, "bin", cmd[0])
    base = None
    for devs in ("", "0", "0,0", "0,0,0") if "pre_barcodes" in label else ("", "0,0", "0,0,0"):
        env = dict(os.environ)
        env.pop("FQGPU_DEVICES", None)
        if devs:
            env["FQGPU_DEVICES"] = devs
        times, says, rc = [], [], 0
        for rep in range(REPS):
            e = dict(env, FQGPU_TIMING="1") if rep == REPS - 1 else env
            t = time.perf_counter()
            p = subprocess.run(cmd, executable=exe, cwd=D, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
            times.append(time.perf_counter() - t)
            rc = rc or p.returncode
            if rep == REPS - 1:
                says = [ln[ln.find("fqgpu timing"):] for ln in p.stderr.decode("latin-1").splitlines() if "fqgpu timing" in ln]
        best = min(times)
        if not devs:
            base = best
        print(f"{label:26s} devices={devs or '-':6s} best {best:6.3f} s  ({N / best:6.1f} Mreads/s)  x{best / base:4.2f} of one context   "
              f"all {[round(x, 3) for x in times]} exit {rc}")
        for s in says:
            print("      " + s[:300])
for name in ("a_1.fastq", "a_2.fastq", "i1.fastq"):
    os.unlink(f"{D}/{name}")
os.rmdir(D)
