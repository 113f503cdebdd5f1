#!/bin/bash
# Evidence run of round 3, part 3 (end of the round, 1 GPU): GPU tests, smoke, every bench workload, the configs[1] sweep
# timeline, rocprofv3 kernel summaries of the workloads whose kernels changed after part 1, the ZI contractions per K,
# ZI-pCMF at the C4 shape.  Results -> gpurun_out/r03b; tools/assemble_profiles_r03b.py files them under profiles/r03_*.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python3 bench.py --workload c2 --steps 200 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err
for w in c3_zi c5_sparse c4_eighth; do
  timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
for w in c2 c3_zi c4; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1)
  find $O/stats_$w -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$w.csv \;
  rm -rf $O/stats_$w
done
(cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 20 --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/$O/trace_c2.log 2>&1)
python3 tools/trace_sweep.py $O/trace_c2/c2_results.db > $O/c2_sweep_timeline.txt 2>&1; rm -rf $O/trace_c2
python3 tools/host_vs_gpu.py > $O/c2_host_vs_gpu.txt 2>&1
python3 tools/perf_small.py > $O/perf_small.txt 2>&1
ORIANA_PASS_IMPL=r2 python3 tools/perf_small.py > $O/perf_small_r2kernels.txt 2>&1
python3 tools/perf_zi_per_k.py 100000 20000 40 48 50 64 65 80 84 96 100 128 > $O/zi_dense_per_k.txt 2>&1
ORIANA_ZI_DN_MINK=1000 python3 tools/perf_zi_per_k.py 100000 20000 40 48 50 80 100 > $O/zi_dense_per_k_round2_kernels.txt 2>&1
timeout 1500 python3 tools/run_models.py ZIGaP 1000000 30000 100 > $O/zigap_c4shape_k100.txt 2>&1
timeout 600 python3 tools/parity_report.py $O/parity_errors.json > $O/parity.txt 2>&1
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items()}, d.get('uninstrumented_ms_per_step'), d.get('parity_slab'))"; done
cat $O/c2_sweep_timeline.txt; tail -2 $O/c2_host_vs_gpu.txt; grep "n=" $O/zi_dense_per_k.txt | cut -c1-150; tail -7 $O/zigap_c4shape_k100.txt; tail -5 $O/parity.txt
