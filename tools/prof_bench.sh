#!/bin/bash
# rocprofv3 kernel summary of the default bench command (full C4, 1 GPU) + the bench line itself
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/bench_prof
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bench_prof -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/bench_prof/log.txt 2>&1
grep '^{' $R/gpurun_out/bench_prof/log.txt > $R/gpurun_out/bench_prof/bench.json
rm -f $R/gpurun_out/bench_prof/*kernel_trace.csv
cut -c1-400 $R/gpurun_out/bench_prof/bench.json
