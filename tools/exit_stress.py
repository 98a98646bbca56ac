"""Do the drop-in programs and the compat library survive leaving through plain exit()?  (a
program once ended with status 139 after all of its output - the HIP runtime's teardown - and the programs have left
through _exit() since; a program that links libfastq_gpu.so under its own main() cannot be told to.)

  python3 tools/exit_stress.py RUNS [WORKERS] [what ...]     what: compat_info | umi | tags | info   (default: all)

Every run is a fresh process on a small golden input, with FQGPU_PLAIN_EXIT=1 (the programs' leave() then calls
exit(), not _exit()) and tools/segv_trace.so preloaded (a backtrace on stderr if the process dies of a signal).  Prints
per target: runs, statuses seen, and the stderr tail of the first runs that did not end with the expected status."""
import collections
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
TRACE = os.path.join(REPO, "tools", "segv_trace.so")
TARGETS = {
    # name: (executable, argv, expected status)
    "compat_info": (os.path.join(REPO, "oracle", "_ref", "fastq_info_on_libfastq_gpu"),
                    ["fastq_info", "data/test_21_1.fastq.gz", "data/test_21_2.fastq.gz"], 0),
    "compat_info_error": (os.path.join(REPO, "oracle", "_ref", "fastq_info_on_libfastq_gpu"),
                          ["fastq_info", "data/test_e9.fastq.gz"], 3),
    # exit() on the first finding while the library's reader thread is pinning / filling the next 512 MiB slot (BIG: a
    # 1.3 GB file made by main(), a bad base in its first record)
    "compat_info_exit_while_reading": (os.path.join(REPO, "oracle", "_ref", "fastq_info_on_libfastq_gpu"), ["fastq_info", "BIG"], 3),
    "info": (os.path.join(REPO, "bin", "fastq_info"), ["fastq_info", "data/c18_10000_1.fastq.gz", "data/c18_10000_2.fastq.gz"], 3),
    "umi": (os.path.join(REPO, "bin", "bam_umi_count"),
            ["bam_umi_count", "--bam", "data_umi/syn_few_umis.bam", "--ucounts", "OUT/u.mtx", "--known_umi", "data_umi/known_umis.txt"], 0),
    "tags": (os.path.join(REPO, "bin", "bam_add_tags"), ["bam_add_tags", "--inbam", "data_tags/trans_small.bam", "--outbam", "OUT/o.bam", "--tx"], 0),
}


def main():
    runs = int(sys.argv[1])
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    what = sys.argv[3:] or list(TARGETS)
    if not os.path.exists(TRACE):
        subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", TRACE, os.path.join(REPO, "tools", "segv_trace.c")], check=True)
    env = dict(os.environ, FQGPU_PLAIN_EXIT="1", LD_PRELOAD=TRACE)
    big = None
    for name in what:
        exe, argv, want = TARGETS[name]
        if "BIG" in argv:
            if big is None:
                big = tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, suffix=".fastq")
                rec = b"@SYN:1:FC:1:1:%d:%d 1:N:0:ACGT\n" + b"ACGT" * 37 + b"AC\n+\n" + b"I" * 150 + b"\n"
                block = b"".join(rec % (i % 97, i) for i in range(4000))
                big.write(block.replace(b"ACGTACGT", b"ACXTACGT", 1))
                for _ in range(1000):
                    big.write(block)
                big.flush()
            argv = [big.name if a == "BIG" else a for a in argv]
        if not os.path.exists(exe):
            print(f"{name}: {exe} is not built - skipped", flush=True)
            continue

        def one(_):
            with tempfile.TemporaryDirectory(dir=GOLD) as tmp:
                a = [x.replace("OUT", os.path.relpath(tmp, GOLD)) for x in argv]
                p = subprocess.run(a, executable=exe, cwd=GOLD, capture_output=True, env=env, timeout=300)
            return p.returncode, p.stderr.decode("latin-1")[-1500:]

        t0 = time.perf_counter()
        with ThreadPoolExecutor(workers) as ex:
            res = list(ex.map(one, range(runs)))
        seen = collections.Counter(rc for rc, _ in res)
        print(f"{name}: {runs} runs in {time.perf_counter() - t0:.0f} s, statuses {dict(seen)} (expected {want})", flush=True)
        for rc, err in [r for r in res if r[0] != want][:3]:
            print(f"--- status {rc}:\n{err}\n---", flush=True)


if __name__ == "__main__":
    main()
