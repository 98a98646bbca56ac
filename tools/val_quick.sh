#!/bin/bash
# quick look at the validate path on the GPU: its tests + the headline bench alone (output under gpurun_out/val_quick)
mkdir -p gpurun_out/val_quick
if [ "${1:-}" != "notest" ]; then
python -m pytest tests/test_gpu_stream.py tests/test_gpu_validate.py -x -q -n 4 > gpurun_out/val_quick/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/val_quick/pytest.txt | head -5
fi
python bench.py --steps 5 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-barcodes-extra --no-filters-extra --no-umi-extra --no-tags-extra --no-filterpair-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('value', round(d['value'],1), 'all', round(r['all_kernels_ms_per_step'],3), json.dumps({k: round(v,3) for k,v in r['kernels_ms_per_step'].items()}))
for k,v in d.get('read_shapes_extra',{}).items(): print(k, round(v['ms_per_pass'],3), v.get('ok'), json.dumps({a: round(b,3) for a,b in v['kernels_ms_per_pass'].items()}))"
