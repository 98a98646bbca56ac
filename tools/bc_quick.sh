#!/bin/bash
# quick look at fastq_pre_barcodes on the GPU: tests + the bench extra alone (output under gpurun_out/bc_quick)
mkdir -p gpurun_out/bc_quick
if [ "${1:-}" != "notest" ]; then
python -m pytest tests/test_gpu_pre_barcodes.py tests/test_gpu_filters.py -x -q > gpurun_out/bc_quick/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/bc_quick/pytest.txt | head -5
fi
python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-filters-extra --no-shapes-extra --no-filterpair-extra --extras-out gpurun_out/bc_quick/bench.json > gpurun_out/bc_quick/bench.out 2> gpurun_out/bc_quick/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bc_quick/bench.json"))
for k, v in d.items():
    if "barcode" in k:
        print(k, json.dumps({a: b for a, b in v.items() if a != "cpu_baseline"})[:1800])
PY
