#!/bin/bash
# times bin/fastq_info -r on a synthetic 150 bp file in /dev/shm: one context against FQGPU_DEVICES with several
# contexts (on a one-GPU box: the same GPU several times).  usage: tools/multi_dev_time.sh [million reads]
set -e
N=${1:-40}
F=/dev/shm/multi_$N.fastq
python3 - "$N" "$F" <<'PY'
import sys, numpy as np
n = int(sys.argv[1]) * 1_000_000
rng = np.random.default_rng(1)
block = 1_000_000
with open(sys.argv[2], "wb") as f:
    done = 0
    while done < n:
        m = min(block, n - done)
        rec = np.empty((m, 12 + 1 + 151 + 2 + 151), dtype=np.uint8)
        names = np.char.zfill(np.arange(done, done + m).astype("U"), 10)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        rec[:, 2:12] = np.frombuffer("".join(names).encode(), dtype=np.uint8).reshape(m, 10)
        rec[:, 12] = 10
        rec[:, 13:163] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (m, 150))
        rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
        rec[:, 166:316] = rng.integers(35, 74, (m, 150), dtype=np.uint8)
        rec[:, 316] = 10
        f.write(rec.tobytes()); done += m
PY
ls -l $F
for devs in "" "0,0" "0,0,0"; do
  for rep in 1 2; do
    s=$(date +%s.%N)
    if [ -z "$devs" ]; then bin/fastq_info -r $F 2>/dev/null >/dev/null; else FQGPU_DEVICES=$devs bin/fastq_info -r $F 2>/dev/null >/dev/null; fi
    e=$(date +%s.%N)
    echo "devices='$devs' rep $rep: $(python3 -c "print(round($N/($e-$s),1))") Mreads/s"
  done
done
for devs in "" "0,0" "0,0,0"; do
  echo "--- FQGPU_TIMING, devices='$devs'"
  if [ -z "$devs" ]; then FQGPU_TIMING=1 bin/fastq_info -r $F 2>&1 >/dev/null | grep "fqgpu timing"; else FQGPU_TIMING=1 FQGPU_DEVICES=$devs bin/fastq_info -r $F 2>&1 >/dev/null | grep "fqgpu timing"; fi
done
FQGPU_DEVICES=0,0 bin/fastq_info -r $F 2>&1 | tail -6
bin/fastq_info -r $F 2>&1 | tail -6
rm -f $F
