#!/bin/bash
# NEEDS a measurement build of the library: make -C fastq_utils_amd/csrc clean && make -C fastq_utils_amd/csrc MEASURE=1 (the shipped library ignores the ablation variables)
# k_bc_emit_tile under its ablation switches (FQGPU_BC_ABL: 1 = no line is written, 2 = no flush, 4 = no name check,
# 8 = no landing of the spans; results are then wrong)
for abl in ${ABLS:-0 1 2 4 3 15}; do
  FQGPU_BC_ABL=$abl python bench.py --reads 4000000 --steps 2 --barcode-pairs ${1:-100000000} --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-filters-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['pre_barcodes_extra']; print('abl $abl', {k: round(v,2) for k,v in b['kernels_ms_breakdown'].items()})"
done
