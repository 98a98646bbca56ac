#!/bin/bash
# NEEDS a measurement build of the library: make -C fastq_utils_amd/csrc clean && make -C fastq_utils_amd/csrc MEASURE=1 (the shipped library ignores the ablation variables)
# k_stream_lines under its ablation switch (FQGPU_LINES_ABL: 1 = no line-index stores; the switch 2, no staged-entry loads, went with the batched requests of round 3)
for abl in 0 1; do
  FQGPU_LINES_ABL=$abl python bench.py --steps 3 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-barcodes-extra --no-filters-extra --no-umi-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms_per_step']; print('abl $abl', 'lines', round(k['k_stream_lines'],3), 'pass1', round(k['k_stream_pass1'],3), 'all', round(d['roofline']['all_kernels_ms_per_step'],3))"
done
