"""How many drop-in programs can start side by side on one GPU box?  Runs the 908 golden fastq_info invocations with
8 / 16 / 32 / 64 worker threads and prints the wall time of each sweep (every invocation = process start + HIP
initialisation + a small file).  Used to size the thread pools of the golden sweeps in tests/."""
import os, subprocess, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import GOLD, REPO, load_fastq_info_golden, strip_progress
BIN = os.path.join(REPO, "bin", "fastq_info")
G = load_fastq_info_golden()
def one(case):
    p = subprocess.run([BIN] + case["args"], cwd=GOLD, capture_output=True, timeout=300)
    return (p.returncode == case["exit"] and p.stdout.decode("latin-1") == case["stdout"]
            and strip_progress(p.stderr.decode("latin-1")) == strip_progress(case["stderr"]))
t0 = time.perf_counter(); one(G[0]); print("one invocation, cold: %.2f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); one(G[0]); print("one invocation, warm: %.2f s" % (time.perf_counter() - t0), flush=True)
for w in [int(x) for x in (sys.argv[1:] or ["8", "16", "32", "64"])]:
    t0 = time.perf_counter()
    with ThreadPoolExecutor(w) as ex:
        ok = list(ex.map(one, G))
    print("%3d workers: %6.1f s for %d invocations, %d differ" % (w, time.perf_counter() - t0, len(G), ok.count(False)), flush=True)
