#!/bin/bash
# usage: tools/evidence.sh <round tag, e.g. r06>     (1 GPU; run through gpurun)
# The evidence run of a round on the end-of-round state of the code: GPU tests; the bench line as the driver runs it (headline +
# the secondary workloads, each in its own child process, in the same line); every workload on its own with its CPU leg; the
# sliced-layout A/B of the headline; each rank's share of configs[3] at 2 / 4 / 8 GPUs on this one GPU (through the SHARDED code
# path: ORIANA_BENCH_FORCE_PG=1) and the PROJECTED scaling table made from them; SURVEY 8(d)'s second regime (z = 0.5);
# rocprofv3 kernel summaries; the NMF-start trace; the parity report; the dense ZI kernels alone.  A round's own A/B runs go into
# tools/evidence_extra.sh (sourced at the end if present: $O and $RND are set).  Results -> gpurun_out/<round>;
# `python tools/assemble_profiles.py <round>` files them under profiles/<round>_*.  (One parametrised pair since round 6; the
# per-round scripts of rounds 2-5 are in the history: git log -- tools/evidence_r05.sh.)
RND=${1:?usage: tools/evidence.sh <round tag>}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$RND
mkdir -p $O
export TMPDIR=/tmp
echo "tools/evidence.sh $RND" > $O/command.txt
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 1800 python3 bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --dense-density 0 > $O/bench_c4_sliced.json 2>/dev/null
for w in c4_half c4_quarter c4_eighth; do
  ORIANA_BENCH_FORCE_PG=1 timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err
done
python3 tools/scaling_projection.py $O/bench_c4.json $O/bench_c4_half.json $O/bench_c4_quarter.json $O/bench_c4_eighth.json $O/scaling_projection.json > $O/scaling_projection.md
cat $O/scaling_projection.md
for w in c2 c3_zi c5_sparse c4_eighth_z05 c3_zi_z05 c5_sparse_z05; do
  timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 900 python3 bench.py --workload c3_zi_nmf --steps 26 --warmup 0 --cpu-rows 400 > $O/bench_c3_zi_nmf.json 2> $O/bench_c3_zi_nmf.err
INIT=nmf SWEEPS=40 timeout 600 python3 tools/zi_slow_path_trace.py > $O/zi_trace_nmf.txt 2>&1
timeout 300 python3 tools/perf_zi_dense.py 100000 20000 50 > $O/perf_zi_dense_k50.txt 2>&1
timeout 300 python3 tools/perf_zi_dense.py 100000 20000 100 > $O/perf_zi_dense_k100.txt 2>&1
timeout 300 python3 tools/perf_gamma.py > $O/perf_gamma.txt 2>&1
timeout 900 python3 tools/parity_report.py $O/parity_errors.json > $O/parity.txt 2>&1
for w in c4 c3_zi c5_sparse; do
  (cd /tmp && ORIANA_BENCH_SECONDARY=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1)
  find $O/stats_$w -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$w.csv \;
  rm -rf $O/stats_$w
done
[ -f tools/evidence_extra.sh ] && source tools/evidence_extra.sh
for f in $O/bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1])
    ks=d['roofline']['kernel_ms']
    print('$f'.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['frac_step'],4), 'fixed', round(d['ms_per_step']-sum(ks.values()),3), {k:round(v,2) for k,v in ks.items()})
except Exception as e:
    print('$f', 'unreadable', e)"; done
tail -12 $O/parity.txt
