export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ws2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/ws2 -o pmc -- python tools/bench_wsift.py 256 > gpurun_out/ws2/log.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ws2/**/pmc_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    name = 'relax6' if 'k_ws_relax<6>' in k else 'relax26' if 'k_ws_relax<26>' in k else None
    if name: acc[name][r['Counter_Name']] += float(r['Counter_Value'])
for n, d in acc.items():
    print(n, {k: '%.3g' % v for k, v in d.items()})
    wc = d.get('SQ_WAVE_CYCLES', 1)
    print('   frac wait_any %.2f wait_inst %.2f active %.2f active_lds %.2f | valu/lds insts %.3g %.3g | bank conflict cycles %.3g' % (
        d['SQ_WAIT_ANY']/wc, d['SQ_WAIT_INST_ANY']/wc, d['SQ_ACTIVE_INST_ANY']/wc, d['SQ_ACTIVE_INST_LDS']/wc, d['SQ_INSTS_VALU'], d['SQ_INSTS_LDS'], d['SQ_LDS_BANK_CONFLICT']))
PY
tail -2 gpurun_out/ws2/log.txt
