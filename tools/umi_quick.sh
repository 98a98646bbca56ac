#!/bin/bash
# quick look at the bam_umi_count kernels on configs[3]: tests + the bench extra alone (output under gpurun_out/)
mkdir -p gpurun_out/umi_quick
if [ "${1:-}" != "notest" ]; then
python -m pytest tests/test_gpu_umi.py -x -q > gpurun_out/umi_quick/pytest.txt 2>&1
grep -n "passed\|failed\|rror" gpurun_out/umi_quick/pytest.txt | head -5
fi
FQGPU_RL_DEBUG=1 python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-barcodes-extra --no-filters-extra --no-shapes-extra --no-filterpair-extra --extras-out gpurun_out/umi_quick/bench.json > gpurun_out/umi_quick/bench.out 2> gpurun_out/umi_quick/bench.err
grep "^\[rl\]" gpurun_out/umi_quick/bench.err | tail -2
python - <<'PY'
import json
d = json.load(open("gpurun_out/umi_quick/bench.json"))
u = d["umi_count_extra"]
print(u["kernels_ms"], json.dumps(u["kernels_ms_breakdown"]))
print(u["matrix_identical_to_reference_program"], u["rl_tree_replay"])
PY
