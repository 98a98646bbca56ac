#!/bin/bash
# SQ counters of the name kernels (default-mode extra of bench.py); summaries under gpurun_out/idx_pmc
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/idx_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-barcodes-extra --no-filters-extra --no-umi-extra --no-tags-extra --no-filterpair-extra --no-shapes-extra --no-dedup-extra"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python3 $R/bench.py $ARGS > $O/run$i.json 2> $O/run$i.err
  find $O/p$i -name '*kernel_trace.csv' -delete
done
python3 $R/tools/pmc_sum.py $O k_names > $O/names_counters.json
python3 $R/tools/pmc_sum.py $O k_index_ >> $O/names_counters.json
python3 $R/tools/pmc_sum.py $O k_stream_pass1 >> $O/names_counters.json
find $O -name '*counter_collection.csv' -delete
cat $O/names_counters.json
