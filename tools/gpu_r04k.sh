mkdir -p gpurun_out/r04k
python -m pytest tests -q -m gpu -n 8 > gpurun_out/r04k/gpu_suite.txt 2>&1
tail -5 gpurun_out/r04k/gpu_suite.txt
