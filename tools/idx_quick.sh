#!/bin/bash
# quick look at the name kernels on the GPU: their tests + the default-mode bench extra (output under gpurun_out/idx_quick)
mkdir -p gpurun_out/idx_quick
if [ "${1:-}" != "notest" ]; then
python -m pytest tests/test_gpu_name_capture.py tests/test_gpu_name_paths.py tests/test_gpu_cli.py tests/test_gpu_filterpair.py tests/test_gpu_dist_names.py tests/test_gpu_dist_pairing.py -x -q -n 4 > gpurun_out/idx_quick/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/idx_quick/pytest.txt | head -5
fi
python bench.py --steps 3 --no-cpu-baseline --no-e2e --no-barcodes-extra --no-filters-extra --no-umi-extra --no-tags-extra --no-filterpair-extra --no-shapes-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(d.get('default_mode_extra'), indent=1)); print(json.dumps(d.get('dedup_extra'))[:600])" | tee gpurun_out/idx_quick/bench.txt
