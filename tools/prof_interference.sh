#!/bin/bash
# Which main-branch kernels pay for the geometry branch running beside them: rocprofv3 kernel stats of bench.py with and without
# the side branch (PN2_LAB_FREEZE_GEOMETRY=1), per-kernel average durations side by side -> gpurun_out/interference.txt
cd /tmp && export TMPDIR=/tmp
root="$GRAFT_REPO_ROOT"
rm -rf /tmp/pa /tmp/pb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $root/gpurun_out/prof_a.log 2>&1
PN2_LAB_FREEZE_GEOMETRY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $root/gpurun_out/prof_b.log 2>&1
cp $(find /tmp/pa -name "*kernel_stats.csv") $root/gpurun_out/interf_with_branch.csv
cp $(find /tmp/pb -name "*kernel_stats.csv") $root/gpurun_out/interf_frozen.csv
python3 - <<'PY'
import csv,os
root=os.environ['GRAFT_REPO_ROOT']
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['Name']]=(int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3)
    return d
a=load(root+'/gpurun_out/interf_with_branch.csv'); b=load(root+'/gpurun_out/interf_frozen.csv')
rows=[]
for n in a:
    if n in b and a[n][0]==b[n][0]:
        rows.append(((a[n][2]-b[n][2])/25.0, n, a[n][0], a[n][1], b[n][1]))
rows.sort(reverse=True)
with open(root+'/gpurun_out/interference.txt','w') as f:
    f.write("us per step a kernel takes longer with the geometry branch beside it (calls identical in both runs; 25 steps)\n")
    for d,n,c,x,y in rows[:40]:
        f.write("%+7.1f  %-90s calls %4d  with %7.1f  alone %7.1f\n"%(d,n[:90],c,x,y))
    f.write("sum over all common kernels: %+.1f us per step\n"%sum(r[0] for r in rows))
print(open(root+'/gpurun_out/interference.txt').read())
PY
