cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05j
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for l in 1 0; do
IVX_WS_LINKS=$l timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/links$l -o kt -- python $GRAFT_REPO_ROOT/bench.py --config watershed --size 512 --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -delete
for l in 1 0; do python - $(find $O/links$l -name "kt_kernel_stats.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-50s calls %5s avg_us %8.1f per_flood_ms %8.3f"%(r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:50],r["Calls"],float(r["AverageNs"])/1e3,int(r["TotalDurationNs"])/3e6))
print()
PY
done
