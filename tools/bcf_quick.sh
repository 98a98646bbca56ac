#!/bin/bash
# quick look at the tile kernels on the GPU (fastq_pre_barcodes + the record filters): the two bench extras alone
mkdir -p gpurun_out/bc_quick
python bench.py --reads ${1:-100000000} --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-shapes-extra --no-filterpair-extra --no-tags-extra --extras-out gpurun_out/bc_quick/bench.json > gpurun_out/bc_quick/bench.out 2> gpurun_out/bc_quick/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bc_quick/bench.json"))
for k, v in d.items():
    if "barcode" in k or "filter" in k:
        v = {a: b for a, b in v.items() if a not in ("cpu_baseline", "whitelist_stage", "what")}
        print(k, json.dumps(v)[:1500])
PY
