#!/usr/bin/env python3
"""The drop-in programs over several contexts of one device, many runs side by side: does every run end, with the same
status and the same bytes?  (A fuzz campaign found a run in a hundred dying inside the runtime when worker threads made
their first copies together: profiles/r05t_first_copy_race.txt.)  `python tools/stress_programs.py <runs> [workers]` on
the GPU box: seeded inputs large enough for several pieces, tools/segv_trace.so preloaded, a run that does not end in 60 s
is asked where its threads are and killed.  Prints, per invocation, how the runs ended and whether their outputs agree."""
import collections
import ctypes
import gzip
import hashlib
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import fuzz  # noqa: E402

TRACE = os.path.join(REPO, "tools", "segv_trace.so")
BIN = os.path.join(REPO, "bin")


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    if not os.path.exists(TRACE):
        subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", TRACE, os.path.join(REPO, "tools", "segv_trace.c")], check=True)
    work = tempfile.mkdtemp(prefix="stress_programs_")
    rng = np.random.default_rng(77)
    sub = 4242
    a = fuzz.make_fastq(np.random.default_rng(sub), 14000, 60, 150, "casava", mate=1)
    b = fuzz.make_fastq(np.random.default_rng(sub), 14000, 60, 150, "casava", mate=2)
    bad = fuzz.mutate(rng, a, "short_qual")
    for name, img in (("a.fastq", a), ("b.fastq", b), ("bad.fastq", bad)):
        with open(os.path.join(work, name), "wb") as f:
            f.write(img)
        with open(os.path.join(work, name + ".gz"), "wb") as f:
            f.write(gzip.compress(img, 1))
    several = {"FQGPU_DEVICES": "0,0,0", "FQGPU_CHUNK_MB": "1", "FQGPU_STREAM_MIN": "256"}
    blocks = {"FQGPU_DEVICES": "0,0,0", "FQGPU_BLOCK_RECORDS": "2000"}
    cases = [
        ("fastq_info -r, three contexts", ["fastq_info", "-r", "a.fastq.gz"], several, []),
        ("fastq_info pairs, three contexts", ["fastq_info", "a.fastq.gz", "b.fastq.gz"], several, []),
        ("fastq_info, a finding in file 1", ["fastq_info", "bad.fastq.gz", "b.fastq"], several, []),
        ("fastq_pre_barcodes, three contexts", ["fastq_pre_barcodes", "--read1", "a.fastq.gz", "--read2", "b.fastq.gz", "--umi_read", "read1",
                                                  "--umi_offset", "0", "--umi_size", "8", "--read1_offset", "8", "--outfile1", "o1.fastq.gz",
                                                  "--outfile2", "o2.fastq.gz"], blocks, ["o1.fastq.gz", "o2.fastq.gz"]),
        ("fastq_pre_barcodes --sam, nothing said (the record-block loop with one context, pinned output buffers)",
         ["fastq_pre_barcodes", "--read1", "a.fastq.gz", "--read2", "b.fastq.gz", "--umi_read", "read1", "--umi_offset", "0", "--umi_size", "8",
          "--sam", "--outfile1", "-"], {"FQGPU_BLOCK_RECORDS": "2000"}, []),
        ("fastq_filterpair (one context; many at a time)", ["fastq_filterpair", "a.fastq.gz", "b.fastq.gz", "p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"],
         {}, ["p1.fastq.gz", "p2.fastq.gz", "up.fastq.gz"]),
        ("fastq_filter_n (one context; many at a time)", ["fastq_filter_n", "-n", "5", "a.fastq.gz"], {"FQGPU_CHUNK_MB": "1"}, []),
    ]
    libc = ctypes.CDLL(None, use_errno=True)
    for label, argv, env_add, outs in cases:
        env = dict(os.environ)
        env.update(env_add)
        env["LD_PRELOAD"] = TRACE

        def once(i):
            d = os.path.join(work, "thread%d" % threading.get_ident())  # (a directory per worker thread: outputs do not meet)
            os.makedirs(d, exist_ok=True)
            for name in os.listdir(work):
                if name.endswith((".fastq", ".gz")) and not os.path.exists(os.path.join(d, name)):
                    os.symlink(os.path.join(work, name), os.path.join(d, name))
            with subprocess.Popen(argv, executable=os.path.join(BIN, argv[0]), cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) as p:
                try:
                    out, err = p.communicate(timeout=60)
                except subprocess.TimeoutExpired:
                    for tid in os.listdir(f"/proc/{p.pid}/task"):
                        libc.syscall(234, p.pid, int(tid), int(signal.SIGUSR1))  # tgkill
                        time.sleep(0.05)
                    time.sleep(1.0)
                    p.kill()
                    out, err = p.communicate()
                    return "hung", "", err.decode("latin-1")
            h = hashlib.sha1(out)
            for o in outs:
                path = os.path.join(d, o)
                raw = open(path, "rb").read() if os.path.exists(path) else b""
                h.update(gzip.decompress(raw) if raw else b"-")
                if os.path.exists(path):
                    os.remove(path)
            return p.returncode, h.hexdigest(), err.decode("latin-1")

        ends, sums, says = collections.Counter(), collections.Counter(), {}
        t0 = time.time()
        with ThreadPoolExecutor(workers) as ex:
            for rc, digest, err in ex.map(once, range(runs)):
                ends[rc] += 1
                sums[digest] += 1
                says.setdefault(rc, err)
        print(f"{label}: {runs} runs in {time.time() - t0:.0f} s: statuses {dict(ends)}, {len(sums)} different output(s)", flush=True)
        for rc, err in says.items():
            if rc not in (0, 1, 3):
                print(f"---- status {rc} ----\n{err[-4000:]}", flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
