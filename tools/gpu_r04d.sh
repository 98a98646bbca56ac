mkdir -p gpurun_out/r04d
timeout 400 ./tools/kbench/kbench 100000000 5 > gpurun_out/r04d/kbench.txt 2>&1
grep -n 'pass1\|v2\|\.\.\.' gpurun_out/r04d/kbench.txt
