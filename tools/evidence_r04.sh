#!/bin/bash
# Evidence run of round 4 (1 GPU): GPU tests, the bench line as the driver runs it (headline + the secondary workloads in the
# same line), the other workloads with their A/B switches, rocprofv3 kernel summaries, the parity report, the RCCL
# rehearsal.  Results -> gpurun_out/r04; tools/assemble_profiles_r04.py copies them into profiles/r04_*.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 1200 python3 bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --dense-density 0 > $O/bench_c4_sliced.json 2>/dev/null
# the round-of-the-chip splits (row pass, dense row kernel, column work list) switched off, same box
ORIANA_ROW_SPLIT_ROUNDS=off ORIANA_DN_TAIL=off ORIANA_COL_ROUNDS=off timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_c4_norounds.json 2>/dev/null
for w in c3_zi c5_sparse; do
  ORIANA_ROW_SPLIT_ROUNDS=off ORIANA_COL_ROUNDS=off timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $O/bench_${w}_norounds.json 2>/dev/null
done
# the den threshold of the shifted form (DESIGN 10 m): the constant 1e-10 against the bound from the row statistics, 20 and 60 sweeps
ORIANA_DEN_THRESHOLD=fixed timeout 900 python3 bench.py --workload c3_zi --steps 20 --warmup 5 --no-cpu > $O/bench_c3_zi_fixedden.json 2>/dev/null
timeout 900 python3 bench.py --workload c3_zi --steps 60 --warmup 5 --no-cpu > $O/bench_c3_zi_60.json 2>/dev/null
ORIANA_DEN_THRESHOLD=fixed timeout 900 python3 bench.py --workload c3_zi --steps 60 --warmup 5 --no-cpu > $O/bench_c3_zi_60_fixedden.json 2>/dev/null
for w in c2 c3_zi c5_sparse c4_eighth; do
  timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
# the two-lane kernels for 33 <= Kp <= 64 against round 3's (ORIANA_PASS_IMPL=r3), same box
for w in c3_zi c5_sparse; do
  ORIANA_PASS_IMPL=r3 timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $O/bench_${w}_r3kernels.json 2>/dev/null
done
# the RCCL path on one rank (nccl backend, every collective of the sharded sweep issued)
ORIANA_BENCH_FORCE_PG=1 timeout 900 python3 bench.py --workload c4_eighth --steps 20 --warmup 5 --no-cpu > $O/bench_c4_eighth_rccl1.json 2> $O/bench_c4_eighth_rccl1.err
# ZI-pCMF with a gene count that is not a multiple of 4 (inert-gene padding) at K = 100 and K = 50
for spec in "ZIGaP 100000 20002 100" "ZIGaP 100000 20000 100" "ZIGaP 100000 20002 50" "ZIGaP 100000 20000 50"; do
  echo "== $spec" >> $O/zigap_gene_count.txt
  ORIANA_DENSE_DENSITY=0 timeout 600 python3 tools/run_models.py $spec 2>&1 | grep -a "sweep" >> $O/zigap_gene_count.txt
done
timeout 900 python3 tools/parity_report.py $O/parity_errors.json > $O/parity.txt 2>&1
for w in c4 c3_zi c5_sparse; do
  (cd /tmp && ORIANA_BENCH_SECONDARY=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1)
  find $O/stats_$w -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$w.csv \;
  rm -rf $O/stats_$w
done
for f in $O/bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:round(v,2) for k,v in d['roofline']['kernel_ms'].items()})
except Exception as e:
    print('$f', 'unreadable', e)"; done
tail -12 $O/parity.txt; cat $O/zigap_gene_count.txt
