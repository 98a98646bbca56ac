mkdir -p gpurun_out/r04l
timeout 400 ./tools/kbench/kbench 100000000 5 > gpurun_out/r04l/kbench.txt 2>&1
grep -n 'pass1\|stream:' gpurun_out/r04l/kbench.txt
