#!/bin/bash
# usage: pmc_cmd.sh "<counters>" TAG "<kernel name substrings, |-separated>" script.py [args...]
# One rocprofv3 --pmc pass (counters only with --kernel-trace) over `python3 script.py args`; per-kernel means of the
# matching kernels -> gpurun_out/pmc/TAG.json.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$1; TAG=$2; FILT=$3; shift 3
D=$R/gpurun_out/pmc/raw_$TAG
mkdir -p $D
S=$R/$1; shift
cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $S "$@" > $D/log.txt 2>&1
python3 - "$D" "$R/gpurun_out/pmc/$TAG.json" "$C" "$FILT" <<'PY'
import csv, glob, collections, json, sys
d, out, counters, filt = sys.argv[1:5]
filt = filt.split('|')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if any(s in k for s in filt):
            k = k.replace('void ', '').replace('oriana::', '')
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in acc.items()}
for k, dd in res.items():
    dd['_dispatches'] = max(len(v) for v in acc[k].values())
json.dump({'counters': counters.split(), 'per_dispatch_mean': res}, open(out, 'w'), indent=1)
for k, dd in res.items():
    print(k, {c: '%.4g' % v for c, v in dd.items()})
PY
rm -rf $D
