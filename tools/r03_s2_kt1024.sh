export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_kt1024
mkdir -p $O
cd $R
for c in watershed watershed_sk; do
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${c}_kt -- python bench.py --config $c --size 1024 --steps 1 --warmup 1 --no-cpu < /dev/null > $O/${c}_kt.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
python - $(find $O -name "${c}_kt_kernel_stats.csv" | head -1) <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    m=re.search(r"(k_\w+(<[^>]*>)?)",r["Name"]); n=m.group(1) if m else r["Name"][:40]
    print("%-36s calls/flood %7.1f avg %9.1f us per flood %7.2f ms"%(n[:36],int(r["Calls"])/2,float(r["AverageNs"])/1e3,int(r["TotalDurationNs"])/2e6))
PY
done
