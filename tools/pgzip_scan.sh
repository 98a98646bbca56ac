#!/bin/bash
# How the many-core gzip reader (fastq_utils_amd/host/fq_pgzip.h) scales on this host: a synthetic single-member .gz
# (10 M reads of 150 bases, written as pigz writes) read with several thread counts and chunk sizes.  CPU only.
set -e
out=${1:-gpurun_out/pgzip_scan}
mkdir -p $out
g++ -O2 -std=c++17 -o $out/pgzip_check tests/cxx/pgzip_check.cpp -lz -pthread
g++ -O2 -std=c++17 -o $out/pgzip_parts tools/kbench/pgzip_parts.cpp -lz -pthread
python3 - "$out" <<'PY'
import sys, numpy as np
out = sys.argv[1]
n, L = 10_000_000, 150
rng = np.random.default_rng(7)
with open("/dev/shm/pgzip_scan.fastq", "wb") as f:
    for a in range(0, n, 1_000_000):
        m = min(1_000_000, n - a)
        R = 16 + L + 3 + L + 1
        rec = np.empty((m, R), dtype=np.uint8)
        rec[:, :5] = np.frombuffer(b"@SYN.", dtype=np.uint8)
        idx = np.arange(a, a + m)
        for d in range(9):
            rec[:, 5 + d] = 48 + (idx // 10 ** (8 - d)) % 10
        rec[:, 14:16] = np.frombuffer(b"/1", dtype=np.uint8)
        rec[:, 16] = 10
        rec[:, 17:17 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (m, L))]
        rec[:, 17 + L:20 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 20 + L:20 + 2 * L] = 33 + np.clip(rng.normal(34, 5, (m, L)), 2, 41).astype(np.uint8)
        rec[:, -1] = 10
        f.write(rec.tobytes())
PY
ls -la /dev/shm/pgzip_scan.fastq
python3 bench.py --gz-helper /dev/shm/pgzip_scan.fastq /dev/shm/pgzip_scan.gz $(stat -c %s /dev/shm/pgzip_scan.fastq)
ls -la /dev/shm/pgzip_scan.gz
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
$out/pgzip_parts /dev/shm/pgzip_scan.gz 2>&1 | tee $out/scan.txt
$out/pgzip_check /dev/shm/pgzip_scan.gz 16 2097152 134217728 2>&1 | tee -a $out/scan.txt
for t in 1 16 32; do for c in 2097152 4194304; do
  $out/pgzip_check /dev/shm/pgzip_scan.gz $t $c 134217728 timing 2>&1 | tee -a $out/scan.txt
done; done
# the symbol buffer's span between slides (KiB of symbols; FQGPU_PGZIP_SPAN)
for k in 68 512; do
  echo "span $k Ki symbols:" | tee -a $out/scan.txt
  FQGPU_PGZIP_SPAN=$k $out/pgzip_check /dev/shm/pgzip_scan.gz 16 4194304 134217728 timing 2>&1 | tee -a $out/scan.txt
done
rm -f /dev/shm/pgzip_scan.fastq /dev/shm/pgzip_scan.gz
