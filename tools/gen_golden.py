#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run in the build container (needs /root/reference and oracle/_ref/, see oracle/Makefile):

    make -C oracle ref && python tools/gen_golden.py

What it writes
  tests/golden/data/*.fastq.gz    the reference's own FASTQ test fixtures (data files, copied
                                  byte for byte from <reference>/tests/)
  tests/golden/fastq_info.json    one entry per fastq_info invocation: argv, exit status, stdout,
                                  stderr, all captured from oracle/_ref/fastq_info (the reference
                                  program compiled unmodified)

The invocation list is the fastq_info section of the reference's run_tests.sh (:252-343) plus a
sweep of every fixture through the single-file modes, and every _1/_2 pair through the paired
modes.  Nothing here is needed at test time on the GPU box; only the outputs are.
"""
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FQ_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")
DATA = os.path.join(GOLD, "data")
REF_BIN = os.path.join(REPO, "oracle", "_ref", "fastq_info")


def run(args):
    p = subprocess.run([REF_BIN] + args, cwd=GOLD, capture_output=True, timeout=120)
    return {
        "args": args,
        "exit": p.returncode,
        "stdout": p.stdout.decode("latin-1"),
        "stderr": p.stderr.decode("latin-1"),
    }


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(DATA, exist_ok=True)
    names = []
    for src in sorted(glob.glob(os.path.join(REF, "tests", "*.fastq.gz"))):
        dst = os.path.join(DATA, os.path.basename(src))
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        names.append(os.path.basename(src))
    # an empty plain file, as run_tests.sh:298 creates with touch
    open(os.path.join(DATA, "empty.fastq"), "wb").close()

    d = lambda n: "data/" + n
    jobs = []
    singles = names + ["empty.fastq"]
    for n in singles:
        for flags in ([], ["-r"], ["-q"], ["-e"], ["-r", "-q"], ["-r", "-e"]):
            jobs.append(flags + [d(n)])
        jobs.append([d(n), "pe"])
    pairs = []
    for n in names:
        if "_1." in n:
            m = n.replace("_1.", "_2.")
            if m in names:
                pairs.append((n, m))
    # cross pairs exercised by run_tests.sh
    pairs += [
        ("test_e19_1.fastq.gz", "test_empty.fastq.gz"),
        ("test_empty.fastq.gz", "test_e19_1.fastq.gz"),
        ("test_empty.fastq.gz", "test_1.fastq.gz"),
        ("test_1.fastq.gz", "test_empty.fastq.gz"),
        ("pe_bug14.fastq.gz", "pe_bug14.fastq.gz"),
        ("casava.1.8_readname_trunc_1.err.fastq.gz", "casava.1.8_readname_trunc_2.fastq.gz"),
        ("casava.1.8_readname_trunc_2.fastq.gz", "casava.1.8_readname_trunc_1.err.fastq.gz"),
        ("casava.1.8_readname_trunc_1.err2.fastq.gz", "casava.1.8_readname_trunc_2.fastq.gz"),
        ("casava.1.8_readname_trunc_1.fastq.gz", "casava.1.8_2.fastq.gz"),
        ("test_1.fastq.gz", "test_2.fastq.gz"),
        ("test_2.fastq.gz", "test_1.fastq.gz"),
        ("test_21_1.fastq.gz", "test_1.fastq.gz"),
    ]
    for a, b in pairs:
        for x, y in ((a, b), (b, a)):
            for flags in ([], ["-r", "-s"], ["-s"], ["-r"], ["-q"]):
                jobs.append(flags + [d(x), d(y)])
    seen, out = set(), []
    for j in jobs:
        k = tuple(j)
        if k in seen:
            continue
        seen.add(k)
        out.append(run(j))
    with open(os.path.join(GOLD, "fastq_info.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    by = {}
    for o in out:
        by[o["exit"]] = by.get(o["exit"], 0) + 1
    print("fastq_info invocations:", len(out), "by exit status:", by)


if __name__ == "__main__":
    main()
