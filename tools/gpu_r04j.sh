mkdir -p gpurun_out/r04j
O=gpurun_out/r04j/cpu.txt
( echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; cat /proc/loadavg; lscpu | grep -E 'Model name|Socket|Core|Thread|MHz' ) > $O 2>&1
python3 - <<'PY' >> $O 2>&1
import os, numpy as np
rng=np.random.default_rng(1)
a=rng.choice(np.frombuffer(b"ACGTNIIIHHFF@:+\n0123456789",dtype=np.uint8), 1<<30).astype(np.uint8)
open('/dev/shm/hp_in.txt','wb').write(a.tobytes())
PY
for t in 8 16 32 64 128 256; do
  /usr/bin/time -f "gz threads $t: %e s wall, %U user, %S sys" env FQGPU_HOST_THREADS=$t ./tools/kbench/hostpar gz /dev/shm/hp_in.txt /dev/shm/hp_out.gz 4 >> $O 2>&1
done
ls -la /dev/shm/hp_out.gz >> $O
rm -f /dev/shm/hp_in.txt /dev/shm/hp_out.gz
cat $O
timeout 300 ./tools/kbench/kbench 100000000 5 > gpurun_out/r04j/kbench.txt 2>&1
grep -n 'pass1' gpurun_out/r04j/kbench.txt
