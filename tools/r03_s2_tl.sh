# one step's kernel timeline (rocprofv3 --kernel-trace) of the default bench under the given environment
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_tl_$1
mkdir -p $O
cd $R
shift
env "$@" timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu < /dev/null > $O/kt.log 2>&1
python - $(find $O -name "kt_kernel_trace.csv" | head -1) <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def nm(r):
    m=re.search(r"(k_\w+)",r["Kernel_Name"]); return m.group(1) if m else r["Kernel_Name"][:30]
idx=[i for i,r in enumerate(rows) if nm(r)=="k_threshold16_bits"]
i0,i1=idx[8],idx[9]
t0=int(rows[i0]["Start_Timestamp"]); pe=None
for r in rows[i0:i1]:
    s=int(r["Start_Timestamp"]);e=int(r["End_Timestamp"])
    print("%-24s start %8.1f dur %6.1f gap %5.1f grid %s"%(nm(r),(s-t0)/1e3,(e-s)/1e3,(s-pe)/1e3 if pe else 0,int(r["Grid_Size_X"])//256)); pe=e
PY
find $O -name "*_kernel_trace.csv" -size +8M -delete
