#!/bin/bash
# usage: pmc_pass.sh "<counters>" tag -- counters of the pass kernels on one rank's share of C4 (bench.py c4_eighth)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$2
cd /tmp && rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$2 -- python3 $R/bench.py --workload c4_eighth --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pmc_$2/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$R/gpurun_out/pmc_$2/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][-28:]
        if 'row_pass' in k or 'col_pass' in k:
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in acc.items():
        print(k, {c: '%.4g' % (sum(v) / len(v)) for c, v in d.items()})
PY
rm -rf $R/gpurun_out/pmc_$2
