mkdir -p gpurun_out/r04g
python -m pytest tests/test_gpu_filters.py tests/test_gpu_filterpair.py -x -q -n 4 > gpurun_out/r04g/t_filters.txt 2>&1
tail -3 gpurun_out/r04g/t_filters.txt
python bench.py --steps 3 --no-cpu-baseline --no-index-extra --no-dedup-extra --no-umi-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench.json'))
print('value', d['value'])
e=d.get('e2e',{})
print('bgzf', e.get('cli_fastq_info_r_bgzf_file'))
f=d.get('filters_extra',{})
print('filters', json.dumps({k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if 'ms' in a or 'ok' in a or 'identical' in a}) for k,v in f.items()}, indent=1)[:1500])
pr=d.get('pre_barcodes_extra',{}).get('programs',{})
print('programs', json.dumps(pr.get('legs'), indent=1))
PY
