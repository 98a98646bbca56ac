mkdir -p gpurun_out/r04e
python -m pytest tests/test_gpu_stream.py tests/test_gpu_validate.py tests/test_gpu_name_capture.py tests/test_gpu_name_paths.py -x -q -n 4 > gpurun_out/r04e/t_val.txt 2>&1
tail -3 gpurun_out/r04e/t_val.txt
bash tools/val_quick.sh notest > gpurun_out/r04e/val_quick.txt 2>&1; cat gpurun_out/r04e/val_quick.txt | tail -4
