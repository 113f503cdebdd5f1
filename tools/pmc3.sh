#!/bin/bash
# usage: pmc3.sh -- FETCH_SIZE and WRITE_SIZE (separate passes) of the pass kernels, bench.py at full C4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $R/gpurun_out/pmc_c4_$c
  cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c4_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pmc_c4_$c/log.txt 2>&1
  tail -1 $R/gpurun_out/pmc_c4_$c/log.txt | cut -c1-200
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/pmc_c4_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if 'row_pass' in k or 'col_pass' in k or 'fixup' in k:
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
json.dump(out, open('$R/gpurun_out/pmc_c4.json', 'w'), indent=1)
PY
