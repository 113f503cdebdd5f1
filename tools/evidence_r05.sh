#!/bin/bash
# Evidence run of round 5 (1 GPU): GPU tests; the bench line as the driver runs it (headline + the secondary workloads, each
# in its own child process, in the same line); each rank's share of configs[3] at 2 / 4 / 8 GPUs on this one GPU and the
# PROJECTED scaling table made from them; A/B of this round's changes on the same box (cell-side preparation folded into the
# Gamma update: ORIANA_FUSED_PREP=off; round 4's Gamma update kernel: ORIANA_GU_KERNEL=r4); rocprofv3 kernel summaries;
# counters of the kernels touched this round; the NMF-start trace; sparse pCMF at K = 100; the parity report; the RCCL
# rehearsal.  Results -> gpurun_out/r05; tools/assemble_profiles_r05.py copies them into profiles/r05_*.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
ORIANA_FUSED_PREP=off timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_c4_unfused_prep.json 2>/dev/null
ORIANA_FUSED_PREP=off ORIANA_GU_KERNEL=r4 timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_c4_r4_updates.json 2>/dev/null
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --dense-density 0 > $O/bench_c4_sliced.json 2>/dev/null
for w in c4_half c4_quarter c4_eighth; do
  timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err
done
python3 tools/scaling_projection.py $O/bench_c4.json $O/bench_c4_half.json $O/bench_c4_quarter.json $O/bench_c4_eighth.json $O/scaling_projection.json > $O/scaling_projection.md
cat $O/scaling_projection.md
for w in c2 c3_zi c5_sparse; do
  timeout 900 python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 900 python3 bench.py --workload c3_zi_nmf --steps 26 --warmup 0 --cpu-rows 400 > $O/bench_c3_zi_nmf.json 2> $O/bench_c3_zi_nmf.err
ORIANA_BENCH_FORCE_PG=1 timeout 900 python3 bench.py --workload c4_eighth --steps 20 --warmup 5 --no-cpu > $O/bench_c4_eighth_rccl1.json 2> $O/bench_c4_eighth_rccl1.err
# the slow path after the reference's default NMF start (same tool and seed as profiles/r04_zigap_slow_path_trace.txt)
INIT=nmf SWEEPS=40 timeout 600 python3 tools/zi_slow_path_trace.py > $O/zi_trace_nmf.txt 2>&1
for spec in "SparseGaP 500000 25000 100" "GaP 500000 25000 100" "SparseGaP 500000 25000 64"; do
  echo "== $spec" >> $O/sparse_k100.txt
  timeout 600 python3 tools/run_models.py $spec 2>&1 | grep -a "sweep" >> $O/sparse_k100.txt
done
timeout 300 python3 tools/perf_gamma.py > $O/perf_gamma.txt 2>&1
ORIANA_GU_KERNEL=r4 timeout 300 python3 tools/perf_gamma.py > $O/perf_gamma_r4kernel.txt 2>&1
timeout 900 python3 tools/parity_report.py $O/parity_errors.json > $O/parity.txt 2>&1
for w in c4 c3_zi c5_sparse; do
  (cd /tmp && ORIANA_BENCH_SECONDARY=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1)
  find $O/stats_$w -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$w.csv \;
  rm -rf $O/stats_$w
done
# HBM bytes of the kernels outside the pass at the headline configuration (separate --pmc passes, counters only)
F="k_gamma_update|k_row_stats|k_factor_prep|k_mstep|k_fixup"
tools/pmc_cmd.sh "FETCH_SIZE" r05_c4_updates_fetch "$F" bench.py --steps 3 --warmup 1 --no-cpu > $O/fetch_c4_updates.txt 2>&1
tools/pmc_cmd.sh "WRITE_SIZE" r05_c4_updates_write "$F" bench.py --steps 3 --warmup 1 --no-cpu > $O/write_c4_updates.txt 2>&1
cp gpurun_out/pmc/r05_*.json $O/ 2>/dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1])
    ks=d['roofline']['kernel_ms']
    print('$f'.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), 'fixed', round(d['ms_per_step']-sum(ks.values()),3), {k:round(v,2) for k,v in ks.items()})
except Exception as e:
    print('$f', 'unreadable', e)"; done
tail -12 $O/parity.txt; cat $O/sparse_k100.txt; cat $O/fetch_c4_updates.txt $O/write_c4_updates.txt | cut -c1-300
