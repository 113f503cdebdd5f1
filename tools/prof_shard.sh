#!/bin/bash
# per-kernel time of one rank's share of C4 (what each rank runs at 8 GPUs)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/shard_prof
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/shard_prof -o s -- python3 $R/bench.py --workload c4_eighth --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/shard_prof/log.txt 2>&1
rm -f $R/gpurun_out/shard_prof/*kernel_trace.csv
python3 - <<PY
import csv, glob
for f in glob.glob('$R/gpurun_out/shard_prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = 0
    for r in rows:
        if 'oriana' in r['Name'] and 'pack' not in r['Name']:
            per_sweep = float(r['TotalDurationNs']) / 1e6 / 12
            tot += per_sweep
            print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%8.3f ms/call %8.3f ms/sweep' % (float(r['AverageNs']) / 1e6, per_sweep))
    print('sum of oriana kernels per sweep: %.3f ms' % tot)
PY
grep '^{' $R/gpurun_out/shard_prof/log.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
