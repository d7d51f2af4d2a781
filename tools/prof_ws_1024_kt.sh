cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ws1024kt
mkdir -p $O
for c in watershed watershed_sk; do
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -o kt -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 1 --warmup 1 --no-cpu > /dev/null 2>&1
done
find $O -name "*_kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
for c in watershed watershed_sk; do python - $(find $O/$c -name "kt_kernel_stats.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-50s calls %5s avg_us %8.1f per_flood_ms %8.3f"%(r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:50],r["Calls"],float(r["AverageNs"])/1e3,int(r["TotalDurationNs"])/2e6))
print("total per flood", tot/2e6); print()
PY
done
