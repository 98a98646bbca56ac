mkdir -p gpurun_out/r04n
python -m pytest tests/test_gpu_cli.py -x -q -n 4 > gpurun_out/r04n/cli.txt 2>&1
tail -3 gpurun_out/r04n/cli.txt
python -m pytest tests/test_pgzip.py -x -q > gpurun_out/r04n/pgzip.txt 2>&1
tail -2 gpurun_out/r04n/pgzip.txt
python bench.py --no-index-extra --no-barcodes-extra --no-filters-extra --no-umi-extra --no-tags-extra --no-filterpair-extra --no-shapes-extra --no-cpu-baseline > gpurun_out/r04n/bench.json 2> gpurun_out/r04n/bench.err
python - <<'PY'
import json
for l in open('gpurun_out/r04n/bench.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print(json.dumps(d['e2e'].get('cli_fastq_info_r_gz_file'),indent=1))
        print(d['e2e'].get('host_cores_usable'), d['host_fed'])
PY
