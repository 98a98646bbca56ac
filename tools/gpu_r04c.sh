mkdir -p gpurun_out/r04c
timeout 200 ./tools/kbench/valubench 2000 > gpurun_out/r04c/valubench.txt 2>&1
timeout 300 ./tools/kbench/kbench 100000000 5 > gpurun_out/r04c/kbench.txt 2>&1
grep -n 'pass1\|census\|count_nl' gpurun_out/r04c/kbench.txt
python -m pytest tests/test_gpu_pre_barcodes.py -x -q -k "several_devices_against_reference_binary" > gpurun_out/r04c/t_prebc.txt 2>&1
tail -3 gpurun_out/r04c/t_prebc.txt
