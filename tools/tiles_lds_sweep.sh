#!/bin/bash
# both kinds of tile kernels (fastq_pre_barcodes on ${1:-100000000} pairs, the record filters on 100 M reads) under LDS
# budgets per wavefront around the default (FQGPU_BC_LDS)
for lds in ${LDS:-16384 18432 20480 22528 24576}; do
  FQGPU_BC_LDS=$lds python bench.py --steps 2 --barcode-pairs ${1:-100000000} --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['filters_extra']; b=d['pre_barcodes_extra']
print('lds $lds', 'barcodes', round(b['kernels_ms'],2), {k: round(v,2) for k,v in b['kernels_ms_breakdown'].items()}, 'filter_n', round(f['filter_n']['kernels_ms'],2), {k: round(v,2) for k,v in f['filter_n']['kernels_ms_breakdown'].items()}, 'trim', round(f['trim_poly_at']['kernels_ms'],2))"
done
