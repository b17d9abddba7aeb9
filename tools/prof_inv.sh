cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pi
PN2_INVERT_GROUPING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_inv.log 2>&1
cp $(find /tmp/pi -name "*kernel_stats.csv") $GRAFT_REPO_ROOT/gpurun_out/prof_inv_stats.csv
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_inv_stats.csv')))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('gather_sum','index_points_backward','invert_index','elementwise','copyBuffer','add')):
        print("%-80s calls %5s avg %8.1f us"%(n[:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
