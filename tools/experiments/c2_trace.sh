cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/c2trace
D=$GRAFT_REPO_ROOT/gpurun_out/c2trace
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 20 --warmup 3 --no-cpu --brief > $D/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('gpurun_out/c2trace/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40 kernels
out=[]
prev=None
for r in rows[-60:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.append('%8.2f us  gap %6.2f  %s' % ((e-s)/1e3, (s-prev)/1e3 if prev else 0, r['Kernel_Name'][:90]))
    prev=e
open('gpurun_out/c2trace/tail.txt','w').write('\n'.join(out))
print('\n'.join(out))
PY
rm -rf gpurun_out/c2trace/*/ 
