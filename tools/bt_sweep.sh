#!/bin/bash
# bam_add_tags bench extra with smaller tiles (FQGPU_BT_T = records per wavefront)
for t in 64 48 32 24 16; do
  FQGPU_BT_T=$t python bench.py --reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-barcodes-extra --no-filters-extra --no-shapes-extra --no-filterpair-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['bam_add_tags_extra']; print('T $t', round(b['kernels_ms'],3), {k: round(v,3) for k,v in b['kernels_ms_breakdown'].items()}, b['first_2000_alignments_identical_to_oracle'])"
done
