#!/bin/bash
# Final evidence run of round 2 (1 GPU): tests, every bench workload, A/B runs, parity report, rocprofv3 summaries,
# SQ / HBM counters, micro-benchmarks (the binaries under scratch/ub are built with: hipcc -O3 --offload-arch=gfx950
# tools/ubench/X.hip -o scratch/ub/X).  Results land in gpurun_out/final; profiles/README.md says which go where.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
for w in c2 c3_zi c5_sparse c4_eighth; do
  timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
ORIANA_PASS_IMPL=r1 timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_c4_r1kernels.json 2>/dev/null
ORIANA_PASS_IMPL=r1 timeout 900 python3 bench.py --workload c4_eighth --steps 20 --warmup 5 --no-cpu > $O/bench_c4_eighth_r1kernels.json 2>/dev/null
ORIANA_ZI_EXACT=1 timeout 900 python3 bench.py --workload c3_zi --steps 20 --warmup 5 --no-cpu > $O/bench_c3_zi_float64.json 2>/dev/null
ORIANA_ZI_MATRIX=f32 timeout 900 python3 bench.py --workload c3_zi --steps 20 --warmup 5 --no-cpu > $O/bench_c3_zi_f32mfma.json 2>/dev/null
ORIANA_SPARSE_ROWS=split ORIANA_SPARSE_COLS=split timeout 900 python3 bench.py --workload c5_sparse --steps 20 --warmup 5 --no-cpu > $O/bench_c5_sparse_split.json 2>/dev/null
timeout 900 python3 tools/parity_report.py $O/parity_errors.json > $O/parity.txt 2>&1
bash tools/prof_r02.sh c4 > $O/prof_c4.txt 2>&1
export TMPDIR=/tmp
for w in c3_zi c5_sparse c4_eighth; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1)
  find $O/stats_$w -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$w.csv \;
  rm -rf $O/stats_$w
done
bash tools/pmc_zi_dense.sh "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" zi_a > $O/zi_a.txt 2>&1
bash tools/pmc_zi_dense.sh "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" zi_b > $O/zi_b.txt 2>&1
for u in mfma_f32 mfma_valu_overlap mfma_bf16x3 lane_row; do timeout 120 scratch/ub/$u > $O/ubench_$u.txt 2>&1; done
timeout 300 python3 tools/perf_zi_dense.py --f64 2>/dev/null > $O/perf_zi_dense.txt
cp gpurun_out/pmc/zi_a.json gpurun_out/pmc/zi_b.json $O/
cp gpurun_out/prof_r02_c4/* $O/ 2>/dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], round(d['value'],2), d.get('ms_per_step_median'), d['roofline']['frac'], {k:round(v,2) for k,v in d['roofline']['kernel_ms'].items()})"; done
cat $O/parity.txt | tail -20
