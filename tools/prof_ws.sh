export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ws1
IVX_WS_GATE=0 IVX_WS_TRACE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ws1 -o kt -- python tools/bench_wsift.py 256 > gpurun_out/ws1/log.txt 2> gpurun_out/ws1/trace.txt
D=$(dirname $(find gpurun_out/ws1 -name "kt_kernel_stats.csv" | head -1))
head -30 $D/kt_kernel_stats.csv
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ws1/**/kt_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_ws_relax' in r['Kernel_Name']]
print(len(rows))
for r in rows[:400:8]:
    print(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), int(r['End_Timestamp'])-int(r['Start_Timestamp']))
PY
