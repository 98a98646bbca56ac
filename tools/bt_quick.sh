#!/bin/bash
# quick look at bam_add_tags on the GPU: tests + the bench extra alone (output under gpurun_out/bt_quick)
mkdir -p gpurun_out/bt_quick
if [ "${1:-}" != "notest" ]; then
python -m pytest tests/test_gpu_bam_tags.py -x -q > gpurun_out/bt_quick/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/bt_quick/pytest.txt | head -5
fi
python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-barcodes-extra --no-filters-extra --no-shapes-extra --no-filterpair-extra --extras-out gpurun_out/bt_quick/bench.json > gpurun_out/bt_quick/bench.out 2> gpurun_out/bt_quick/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bt_quick/bench.json"))
print(json.dumps(d.get("bam_add_tags_extra"), indent=0)[:2500])
PY
