#!/bin/bash
# Round-2 evidence run (full C4, 1 GPU): rocprofv3 kernel summary of the bench command, three SQ counter passes and
# the two HBM counter passes (each --pmc pass on its own, counters only with --kernel-trace).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-c4}
O=$R/gpurun_out/prof_r02_$W
mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --workload $W --steps 5 --warmup 1 --no-cpu > $O/bench_under_rocprof.log 2>&1
find $O/stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats
cd $R
tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ${W}_sq1 $W > $O/sq1.txt 2>&1
tools/pmc_pass.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" ${W}_sq2 $W > $O/sq2.txt 2>&1
tools/pmc_pass.sh "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" ${W}_sq3 $W > $O/sq3.txt 2>&1
tools/pmc_pass.sh "FETCH_SIZE" ${W}_fetch $W > $O/fetch.txt 2>&1
tools/pmc_pass.sh "WRITE_SIZE" ${W}_write $W > $O/write.txt 2>&1
cp $R/gpurun_out/pmc/${W}_*.json $O/
ls $O
