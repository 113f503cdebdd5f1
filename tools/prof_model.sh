#!/bin/bash
# usage: prof_model.sh tag <run_models args...>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out/$tag
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o p -- python3 $R/tools/run_models.py "$@" > $R/gpurun_out/$tag/log.txt 2>&1
grep -E "sweep|setup" $R/gpurun_out/$tag/log.txt | tail -5
python3 - <<PY
import csv, glob
for f in glob.glob('$R/gpurun_out/$tag/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:18]:
        print(r['Name'][:80].ljust(80), r['Calls'].rjust(5), '%10.1f us' % (float(r['AverageNs']) / 1e3), '%6.1f%%' % float(r['Percentage']))
PY
