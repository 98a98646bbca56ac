#!/bin/bash
# fastq_pre_barcodes bench extra under different LDS budgets per wavefront (FQGPU_BC_LDS)
for lds in 12288 16384 20480 28672 40960 65536; do
  FQGPU_BC_LDS=$lds FQGPU_BC_DEBUG=1 python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-filters-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra 2>/tmp/bc_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['pre_barcodes_extra']; print('lds $lds', round(b['kernels_ms'],2), {k: round(v,2) for k,v in b['kernels_ms_breakdown'].items()}, b.get('first_2000_pairs_identical_to_oracle'))"
  grep "fqgpu barcodes" /tmp/bc_err.txt | tail -1
done
