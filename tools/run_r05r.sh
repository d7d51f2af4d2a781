cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05r
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in watershed watershed_sk; do
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -o kt -- python $GRAFT_REPO_ROOT/bench.py --config $c --size 512 --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -delete
for c in watershed watershed_sk; do python - $(find $O/$c -name "kt_kernel_stats.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-50s calls %5s avg_us %8.1f per_flood_ms %8.3f"%(r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:50],r["Calls"],float(r["AverageNs"])/1e3,int(r["TotalDurationNs"])/3e6))
print("total per flood", tot/3e6); print()
PY
done
