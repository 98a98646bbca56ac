mkdir -p gpurun_out/r04i
nproc > gpurun_out/r04i/nproc.txt; python -c "import os;print(len(os.sched_getaffinity(0)), os.cpu_count())" >> gpurun_out/r04i/nproc.txt; cat gpurun_out/r04i/nproc.txt
python -m pytest tests/test_gpu_stream.py tests/test_gpu_validate.py tests/test_gpu_cli.py tests/test_gpu_compat.py tests/test_gpu_pre_barcodes.py -x -q -n 6 > gpurun_out/r04i/t.txt 2>&1
tail -3 gpurun_out/r04i/t.txt
python bench.py --steps 5 --no-cpu-baseline --no-index-extra --no-dedup-extra --no-umi-extra --no-tags-extra --no-filterpair-extra --no-filters-extra > gpurun_out/r04i/bench.json 2> gpurun_out/r04i/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04i/bench.json'))
print('value', d['value'], d['roofline']['kernels_ms_per_step'])
for k,v in d.get('read_shapes_extra',{}).items(): print(k, v.get('ms_per_pass'), v.get('ok'), v.get('kernels_ms_per_pass'))
e=d.get('e2e',{})
print('cores', e.get('host_cores_usable'), e.get('host_cores'))
print('bgzf', e.get('cli_fastq_info_r_bgzf_file'))
pr=d.get('pre_barcodes_extra',{}).get('programs',{})
print('programs', json.dumps(pr.get('legs'), indent=1), pr.get('prefix_check'))
PY
