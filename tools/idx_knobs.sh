#!/bin/bash
# name kernels under ablations / launch shapes (measurement only): tools/idx_knobs.sh
run() {
  echo "== $*"
  env "$@" timeout 150 python bench.py --steps 2 --reads ${READS:-100000000} --no-cpu-baseline --no-e2e --no-barcodes-extra --no-filters-extra --no-umi-extra --no-tags-extra --no-filterpair-extra --no-shapes-extra 2>&1 | python -c "
import json,sys
txt=sys.stdin.read()
try:
  d=json.loads(txt[txt.index('{\"metric'):])['default_mode_extra']
  k=d['kernels_ms']
  print('insert', round(k.get('k_names_insert',0),2), 'pair-insert', round(d.get('pair_first_file',{}).get('insert_ms',0),2), 'rest', round(k.get('k_names_rest',0),2), 'pass1n', round(k.get('k_stream_pass1(names)',0),2), d.get('file2_loop',{}).get('kernels_ms',{}).get('k_names_match'))
except Exception as e:
  print('failed', repr(e), txt[-300:])"
}
run A=0
run FQGPU_NAMES_NT=0
run FQGPU_NAMES_DYN_LDS=30000
run FQGPU_NAMES_BLOCKS_PER_CU=6
run FQGPU_NAMES_BLOCKS_PER_CU=32
