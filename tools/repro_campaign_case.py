#!/usr/bin/env python3
"""One invocation of a tools/fuzz_campaign.py case, many times: for a difference that does not show on every run (a
process that died of a signal).  `python tools/repro_campaign_case.py <seed> <env tag> <runs> <workers> <args...>` on the
GPU box: writes the case's files once, runs bin/fastq_info on them `runs` times with tools/segv_trace.so preloaded (a
backtrace on stderr when the process dies of a signal) and prints how the runs ended, with the tail of stderr of every
kind of ending."""
import collections
import ctypes
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, REPO)
import fuzz_campaign as fc  # noqa: E402

TRACE = os.path.join(REPO, "tools", "segv_trace.so")


def main():
    seed, tag, runs, workers = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    args = sys.argv[5:]
    patience = float(os.environ.get("REPRO_PATIENCE", "60"))
    env = dict(os.environ)
    env.update(dict(fc.ENVS)[tag])
    if not os.path.exists(TRACE):
        subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", TRACE, os.path.join(REPO, "tools", "segv_trace.c")], check=True)
    env["LD_PRELOAD"] = TRACE
    work = tempfile.mkdtemp(prefix="repro_case_")

    def keep_files(binary, a, cwd, e=None):  # the first run of one_case: keep its files, run nothing
        if not os.listdir(work):
            for name in os.listdir(cwd):
                shutil.copy(os.path.join(cwd, name), os.path.join(work, name))
        return ("timeout", "", "")

    fc.run = keep_files
    fc.one_case(seed)
    want = subprocess.run(["fastq_info"] + args, executable=fc.REF, cwd=work, capture_output=True, timeout=300)
    print("reference:", want.returncode, want.stderr.decode("latin-1")[-200:].replace("\n", " | "), flush=True)

    libc = ctypes.CDLL(None, use_errno=True)

    def once(i):
        with subprocess.Popen(["fastq_info"] + args, executable=fc.BIN, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) as p:
            try:
                _, err = p.communicate(timeout=patience)
                return p.returncode, err.decode("latin-1")
            except subprocess.TimeoutExpired:
                # a run that does not end: every thread says where it is (segv_trace.c's SIGUSR1 handler), then it is killed
                for tid in os.listdir(f"/proc/{p.pid}/task"):
                    libc.syscall(234, p.pid, int(tid), int(signal.SIGUSR1))  # tgkill
                    time.sleep(0.05)
                time.sleep(1.0)
                p.kill()
                _, err = p.communicate()
                return "hung", err.decode("latin-1")

    ends = collections.Counter()
    says = {}
    with ThreadPoolExecutor(workers) as ex:
        for rc, err in ex.map(once, range(runs)):
            ends[rc] += 1
            says.setdefault(rc, err)
    print("runs:", dict(ends), flush=True)
    for rc, err in says.items():
        print(f"---- an ending with status {rc} ----\n{err[-6000:]}", flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
