export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fr
IVX_FLOOD_TRACE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fr -o kt -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/fr/log.txt 2> gpurun_out/fr/trace.txt
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/fr/**/kt_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find last k_flood_clear and print the step after it
idx = [i for i, r in enumerate(rows) if 'k_flood_clear' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
import re
for r in rows[i0:i0 + 40]:
    m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
    print('%8.1f us  +%6.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, m.group(1) if m else r['Kernel_Name'][:30]))
PY
grep "ivx flood" gpurun_out/fr/trace.txt | tail -18
