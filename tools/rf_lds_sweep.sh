#!/bin/bash
# record filters (fastq_filter_n / fastq_trim_poly_at) under different LDS budgets per wavefront (FQGPU_BC_LDS)
for lds in 12288 16384 20480 28672 40960; do
  FQGPU_BC_LDS=$lds python bench.py --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-barcodes-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['filters_extra']; print('lds $lds', 'filter_n', round(f['filter_n']['kernels_ms'],2), {k: round(v,2) for k,v in f['filter_n']['kernels_ms_breakdown'].items()}, 'trim', round(f['trim_poly_at']['kernels_ms'],2), f['filter_n']['first_2000_records_identical_to_oracle'])"
done
