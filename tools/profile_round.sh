#!/bin/bash
# The rocprofv3 evidence of a round, collected on the GPU box: tools/profile_round.sh <tag> [bench|traffic|umi ...]
#   bench    the judged bench line as the driver runs it + `--kernel-trace --stats` of the same workload
#   traffic  FETCH_SIZE / WRITE_SIZE in separate passes on the headline workload (+ a --two-pass run: k_count_nl, the
#            kernel with a known byte count the read correction is checked on) -> <tag>_traffic_100M_150bp.json
#   umi      FETCH_SIZE / WRITE_SIZE / SQ counters of the bam_umi_count kernels on configs[3]
#   index    FETCH_SIZE / WRITE_SIZE of the default-mode extra (the name kernels) -> traffic_index_100M.json
#   pass1sq  SQ counters of the streaming kernels (headline workload) -> pass1_sq_counters.json
#   micro    the random-access and decode microbenchmarks -> rmwbench.txt, decbench.txt
#   tilessq  SQ counters of the tile kernels of fastq_pre_barcodes (50 M pairs) -> tiles_sq_counters.json
#   tiles    FETCH_SIZE / WRITE_SIZE of the tile kernels (fastq_pre_barcodes on 50 M pairs, the record filters on
#            100 M reads) -> traffic_tile_kernels.json
# Counter passes never carry another trace domain than --kernel-trace.  Summaries land in gpurun_out/<tag>/; copy what
# is to be judged into profiles/.
set -u
TAG=$1; shift
WHAT=${*:-bench traffic umi}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ONLY_HEADLINE="--no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-barcodes-extra --no-filters-extra --no-umi-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra"
ONLY_UMI="--reads 4000000 --steps 2 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-barcodes-extra --no-filters-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra"
for w in $WHAT; do
  case $w in
  bench)
    python3 $R/bench.py --extras-out $O/bench.json > $O/bench_stdout.txt 2> $O/bench.err  # bench.json: headline + every extra as one document; the last line of bench_stdout.txt is the judged line
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e > $O/under_rocprof.json 2> $O/under_rocprof.err
    find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
    find $O/stats -name '*kernel_trace.csv' -delete
    # the headline workload alone (what bench.py's roofline block times): its average launch durations are the ones to
    # hold against roofline.avg_launch_ms
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -o stats -- python3 $R/bench.py $ONLY_HEADLINE > $O/headline_under_rocprof.json 2> $O/headline_under_rocprof.err
    find $O/stats_headline -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_headline.csv \;
    find $O/stats_headline -name '*kernel_trace.csv' -delete
    ;;
  traffic)
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 2 $ONLY_HEADLINE > $O/pmc_$c.json 2> $O/pmc_$c.err
    done
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib -o pmc -- python3 $R/bench.py --steps 2 --two-pass $ONLY_HEADLINE > $O/pmc_calib.json 2> $O/pmc_calib.err
    python3 $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv') $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv') \
        34900000000 $(find $O/pmc_calib -name '*counter_collection.csv') > $O/traffic_100M_150bp.json
    find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_calib -name '*kernel_trace.csv' -delete
    ;;
  index)
    # HBM bytes of the name kernels (default mode: validate + k_index_insert over 100 M unique names)
    ONLY_INDEX="--steps 2 --no-cpu-baseline --no-e2e --no-dedup-extra --no-barcodes-extra --no-filters-extra --no-umi-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra"
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/idx_pmc_$c -o pmc -- python3 $R/bench.py $ONLY_INDEX > $O/idx_pmc_$c.json 2> $O/idx_pmc_$c.err
      find $O/idx_pmc_$c -name '*kernel_trace.csv' -delete
    done
    python3 $R/tools/pmc_traffic.py $(find $O/idx_pmc_FETCH_SIZE -name '*counter_collection.csv') $(find $O/idx_pmc_WRITE_SIZE -name '*counter_collection.csv') \
        34900000000 > $O/traffic_index_100M.json 2>$O/traffic_index.err || true
    ;;
  pass1sq)
    # issue-side counters of the dominant kernel (k_stream_pass1<0>): is it the VALU that bounds it?
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
               "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
      i=$((i+1))
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p1sq/p$i -o pmc -- python3 $R/bench.py --steps 2 $ONLY_HEADLINE > $O/p1sq_$i.json 2> $O/p1sq_$i.err
      find $O/p1sq/p$i -name '*kernel_trace.csv' -delete
    done
    python3 $R/tools/pmc_sum.py $O/p1sq k_stream > $O/pass1_sq_counters.json
    find $O/p1sq -name '*counter_collection.csv' -delete
    ;;
  micro)
    # what one random access costs on this GPU (tools/kbench/rmwbench.hip), and what decoding a capture record costs
    (cd $R/tools/kbench && timeout 300 ./rmwbench 100 30 35) > $O/rmwbench.txt 2>&1
    (cd $R/tools/kbench && timeout 120 ./decbench) > $O/decbench.txt 2>&1
    ;;
  tilessq)
    ONLY_BC="--reads 4000000 --steps 2 --barcode-pairs 50000000 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-filters-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra"
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
               "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
      i=$((i+1))
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/bcsq/p$i -o pmc -- python3 $R/bench.py $ONLY_BC > $O/bcsq_$i.json 2> $O/bcsq_$i.err
      find $O/bcsq/p$i -name '*kernel_trace.csv' -delete
    done
    python3 $R/tools/pmc_sum.py $O/bcsq k_bc_ > $O/tiles_sq_counters.json
    find $O/bcsq -name '*counter_collection.csv' -delete
    ;;
  tiles)
    ONLY_TILES="--steps 2 --barcode-pairs 50000000 --no-cpu-baseline --no-e2e --no-index-extra --no-dedup-extra --no-umi-extra --no-shapes-extra --no-tags-extra --no-filterpair-extra"
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/tiles_pmc_$c -o pmc -- python3 $R/bench.py $ONLY_TILES > $O/tiles_pmc_$c.json 2> $O/tiles_pmc_$c.err
      find $O/tiles_pmc_$c -name '*kernel_trace.csv' -delete
    done
    python3 $R/tools/pmc_traffic.py $(find $O/tiles_pmc_FETCH_SIZE -name '*counter_collection.csv') $(find $O/tiles_pmc_WRITE_SIZE -name '*counter_collection.csv') \
        34900000000 > $O/traffic_tile_kernels.json 2>$O/traffic_tiles.err || true
    ;;
  umistats)
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/umi_stats -o stats -- python3 $R/bench.py $ONLY_UMI > $O/umi_under_rocprof.json 2> $O/umi_under_rocprof.err
    find $O/umi_stats -name '*kernel_stats.csv' -exec cp {} $O/umi_kernel_stats.csv \;
    find $O/umi_stats -name '*kernel_trace.csv' -delete
    ;;
  umi)
    i=0
    for set in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
               "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
      i=$((i+1))
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/umi_pmc/p$i -o pmc -- python3 $R/bench.py $ONLY_UMI > $O/umi_pmc_$i.json 2> $O/umi_pmc_$i.err
      find $O/umi_pmc/p$i -name '*kernel_trace.csv' -delete
    done
    python3 $R/tools/pmc_sum.py $O/umi_pmc k_umi > $O/umi_counters.json
    python3 $R/tools/pmc_sum.py $O/umi_pmc k_rl >> $O/umi_counters.json
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/umi_stats -o stats -- python3 $R/bench.py $ONLY_UMI > $O/umi_under_rocprof.json 2> $O/umi_under_rocprof.err
    find $O/umi_stats -name '*kernel_stats.csv' -exec cp {} $O/umi_kernel_stats.csv \;
    find $O/umi_stats -name '*kernel_trace.csv' -delete
    ;;
  esac
done
du -sh $O; ls $O
