mkdir -p gpurun_out/r04f
python -m pytest tests/test_gpu_cli.py -x -q -k "bgzipped or gzgets_buffers and plain" > gpurun_out/r04f/t_cli.txt 2>&1
tail -3 gpurun_out/r04f/t_cli.txt
python -m pytest tests/test_gpu_stream.py tests/test_gpu_validate.py -x -q -n 4 > gpurun_out/r04f/t_val.txt 2>&1
tail -2 gpurun_out/r04f/t_val.txt
python bench.py --steps 5 --no-cpu-baseline --no-index-extra --no-dedup-extra --no-filters-extra --no-umi-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04f/bench.json'))
print('value', d['value'], d['roofline']['kernels_ms_per_step'])
print('host_fed', d.get('host_fed'))
e=d.get('e2e',{})
print('bgzf', e.get('cli_fastq_info_r_bgzf_file'))
print('cli', {k:v for k,v in (e.get('cli_fastq_info_r_tmpfs_file') or {}).items() if k!='variants'})
print('programs', json.dumps(d.get('pre_barcodes_extra',{}).get('programs'), indent=1))
PY
