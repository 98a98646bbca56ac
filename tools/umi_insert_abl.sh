#!/bin/bash
# NEEDS a measurement build of the library: make -C fastq_utils_amd/csrc clean && make -C fastq_utils_amd/csrc MEASURE=1 (the shipped library ignores the ablation variables)
# k_umi_insert under its ablation switches (results are wrong with them; only the kernel's time is of interest)
for abl in 0 1 2 4 7; do
  FQGPU_UMI_INSERT_ABL=$abl python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-barcodes-extra --no-filters-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['umi_count_extra']; print('abl $abl', u.get('kernels_ms_breakdown',{}).get('k_umi_insert'), u.get('error'))"
done
