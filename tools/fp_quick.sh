#!/bin/bash
# quick look at the fastq_filterpair bench extra (output under gpurun_out/fp_quick)
mkdir -p gpurun_out/fp_quick
python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-barcodes-extra --no-filters-extra --no-shapes-extra --no-tags-extra ${FP_ARGS:-} --extras-out gpurun_out/fp_quick/bench.json > gpurun_out/fp_quick/bench.out 2> gpurun_out/fp_quick/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/fp_quick/bench.json"))
print(json.dumps(d.get("filterpair_extra"), indent=0)[:2500])
PY
tail -3 gpurun_out/fp_quick/bench.err
