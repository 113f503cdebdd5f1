#!/bin/bash
# Evidence run of round 3 (1 GPU), part 1: GPU tests, every bench workload (hybrid default and sliced-only A/B), the
# threshold sweep of the hybrid layout at full C4, parity report, rocprofv3 kernel summaries.  Results -> gpurun_out/r03;
# tools/assemble_profiles_r03.py copies them into profiles/r03_*.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --dense-density 0 > $O/bench_c4_sliced.json 2>/dev/null
for w in c2 c3_zi c5_sparse c4_eighth c4_eighth_z05; do
  timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 900 python3 bench.py --workload c4_eighth --steps 20 --warmup 5 --no-cpu --dense-density 0 > $O/bench_c4_eighth_sliced.json 2>/dev/null
timeout 1200 python3 tools/perf_dense_pass.py 1000000 30000 100 0.1 0 0.3 0.25 0.2 0.15 0.1 > $O/dense_threshold_c4.txt 2>/dev/null
timeout 900 python3 tools/parity_report.py $O/parity_errors.json > $O/parity.txt 2>&1
for w in c4 c4_eighth c3_zi c5_sparse; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1)
  find $O/stats_$w -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$w.csv \;
  rm -rf $O/stats_$w
done
timeout 120 scratch/mfma_mix > $O/ubench_mfma_mix.txt 2>&1
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:round(v,2) for k,v in d['roofline']['kernel_ms'].items()}, d.get('parity_slab'))"; done
tail -20 $O/parity.txt
