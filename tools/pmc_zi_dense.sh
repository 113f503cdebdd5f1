#!/bin/bash
# usage: pmc_zi_dense.sh "<counters>" TAG
# Hardware counters of the dense kernels of a ZI sweep (tools/perf_zi_dense.py at configs[2]), one rocprofv3 --pmc pass
# (counters only with --kernel-trace).  Per-kernel averages -> gpurun_out/pmc/TAG.json.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/pmc/raw_$2
mkdir -p $D
cd /tmp && rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $D -- python3 $R/tools/perf_zi_dense.py --shipped-only > $D/log.txt 2>&1
python3 - "$D" "$R/gpurun_out/pmc/$2.json" "$1" <<'PY'
import csv, glob, collections, json, sys
d, out, counters = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if any(s in k for s in ('k_dropout_sweep', 'k_dt_times_factor')):
            k = k.split('::')[-1].split('<')[0]
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in acc.items()}
for k, dd in res.items():
    dd['_dispatches'] = max(len(v) for v in acc[k].values())
json.dump({'workload': 'perf_zi_dense 100000 x 20000 K=50', 'counters': counters.split(), 'per_dispatch_mean': res}, open(out, 'w'), indent=1)
for k, dd in res.items():
    print(k, {c: '%.4g' % v for c, v in dd.items()})
PY
rm -rf $D
