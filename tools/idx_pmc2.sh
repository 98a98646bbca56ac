#!/bin/bash
# NEEDS a measurement build of the library: make -C fastq_utils_amd/csrc clean && make -C fastq_utils_amd/csrc MEASURE=1 (the shipped library ignores the ablation variables)
# instruction-fetch counters of the capture-fed name kernel with and without the decode (FQGPU_NAMES_ABL 1 / 9)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/idx_pmc2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --reads 50000000 --no-cpu-baseline --no-e2e --no-barcodes-extra --no-filters-extra --no-umi-extra --no-tags-extra --no-filterpair-extra --no-shapes-extra --no-dedup-extra"
for abl in 1 9; do
  export FQGPU_NAMES_ABL=$abl
  timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/a$abl -o pmc -- python3 $R/bench.py $ARGS > $O/run$abl.json 2> $O/run$abl.err
  find $O/a$abl -name '*kernel_trace.csv' -delete
  echo "ABL=$abl"; python3 $R/tools/pmc_sum.py $O/a$abl k_names_pass
  find $O/a$abl -name '*counter_collection.csv' -delete
done
