#!/bin/bash
# What one program start costs in user and kernel time on the GPU box (the GPU test suite is bounded by program starts:
# profiles/r05a_golden_concurrency.txt): the bare runtime (tools/kbench/hipinit, with and without a kernel launch)
# against the drop-in programs on small inputs, 20 starts each, one at a time.
cd "$(dirname "$0")/.."
printf '@r1\nACGTACGTAC\n+\nIIIIIIIIII\n%.0s' $(seq 1 100) > /tmp/small.fastq
gzip -c /tmp/small.fastq > /tmp/small.fastq.gz
t() { local label="$1"; shift; local s=$(date +%s.%N); local tm; tm=$( { time -p (for i in $(seq 1 20); do "$@" > /dev/null 2>&1; done) ; } 2>&1 ); echo "$label: $(echo $tm | awk '{printf "real %.0f ms, user %.0f ms, sys %.0f ms per start", $2*50, $4*50, $6*50}')"; }
t "hipinit (no kernel)          " tools/kbench/hipinit
t "hipinit + one empty kernel   " tools/kbench/hipinit k
t "fastq_info -r small.fastq    " bin/fastq_info -r /tmp/small.fastq
t "fastq_info small.fastq       " bin/fastq_info /tmp/small.fastq
t "fastq_info -r small.fastq.gz " bin/fastq_info -r /tmp/small.fastq.gz
t "fastq_filter_n small.fastq   " bin/fastq_filter_n /tmp/small.fastq
t "reference fastq_info -r      " oracle/_ref/fastq_info -r /tmp/small.fastq
FQGPU_TIMING=1 bin/fastq_info -r /tmp/small.fastq 2>&1 | tail -5
