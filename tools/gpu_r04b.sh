mkdir -p gpurun_out/r04b
timeout 120 ./tools/kbench/valubench 2000 > gpurun_out/r04b/valubench.txt 2>&1
python -m pytest tests/test_gpu_cli.py -x -q -k "gzgets or long_line_late" > gpurun_out/r04b/t_gzgets.txt 2>&1
tail -5 gpurun_out/r04b/t_gzgets.txt
python -m pytest tests/test_gpu_pre_barcodes.py -x -q -k "several_devices_against_reference_binary" > gpurun_out/r04b/t_prebc.txt 2>&1
tail -3 gpurun_out/r04b/t_prebc.txt
python -m pytest tests/test_gpu_bench_multirank.py -x -q > gpurun_out/r04b/t_multirank.txt 2>&1
tail -3 gpurun_out/r04b/t_multirank.txt
