#!/bin/bash
# usage: pmc_pass.sh "<counters>" TAG [WORKLOAD]
# Hardware counters of the pass kernels for `bench.py --workload WORKLOAD` (default c4), one rocprofv3 --pmc
# pass (counters only with --kernel-trace: no other trace domain).  The per-kernel averages are written to
# gpurun_out/pmc/TAG.json (copy what is to be judged into profiles/); the raw CSV is removed.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${3:-c4}
D=$R/gpurun_out/pmc/raw_$2
mkdir -p $D
cd /tmp && rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu > $D/log.txt 2>&1
python3 - "$D" "$R/gpurun_out/pmc/$2.json" "$1" "$W" <<'PY'
import csv, glob, collections, json, sys
d, out, counters, workload = sys.argv[1:5]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if any(s in k for s in ('row_pass', 'col_pass', 'fixup', 'rowcol')):
            k = k.split('::')[-1].split('<')[0] if '<' in k else k
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in acc.items()}
for k, dd in res.items():
    dd['_dispatches'] = max(len(v) for v in acc[k].values())
json.dump({'workload': workload, 'counters': counters.split(), 'per_dispatch_mean': res}, open(out, 'w'), indent=1)
for k, dd in res.items():
    print(k, {c: '%.4g' % v for c, v in dd.items()})
PY
rm -rf $D
