# kernel stats of the default bench (rocprofv3 --kernel-trace --stats), printed compactly
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_kt_$1
mkdir -p $O
cd $R
shift
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu "$@" < /dev/null > $O/kt.log 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
python - $(find $O -name "kt_kernel_stats.csv" | head -1) <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    m=re.search(r"(k_\w+(<[^>]*>)?)",r["Name"]); n=m.group(1) if m else r["Name"][:40]
    print("%-36s calls %5s avg %8.1f us tot %7.2f ms"%(n[:36],r["Calls"],float(r["AverageNs"])/1e3,int(r["TotalDurationNs"])/1e6))
PY
