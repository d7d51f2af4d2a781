export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c2
mkdir -p $O
cd $R
for t in 4 8 12 16; do for c in 2 4 8; do IVX_STAGE_THREADS=$t IVX_STAGE_CHUNK_MB=$c timeout 120 python tools/bench_stage.py < /dev/null >> $O/stage.txt 2>&1; done; done
IVX_STAGE_THREADS=0 timeout 120 python tools/bench_stage.py < /dev/null >> $O/stage.txt 2>&1
for i in 1 2; do
  timeout 200 python bench.py --no-cpu --steps 20 < /dev/null > $O/bench_seq_$i.json 2>> $O/ab.err
  IVX_PREFETCH=1 timeout 200 python bench.py --no-cpu --steps 20 < /dev/null > $O/bench_prefetch_$i.json 2>> $O/ab.err
done
timeout 300 python tools/ccl_size.py < /dev/null > $O/ccl_rounds.txt 2>&1
IVX_FLOOD_MODE=ccl timeout 300 python tools/ccl_size.py < /dev/null > $O/ccl_ccl.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_mip -- python bench.py --config mip --steps 5 --warmup 1 --no-cpu < /dev/null > $O/kt_mip.log 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
cat $O/stage.txt; cat $O/ccl_rounds.txt $O/ccl_ccl.txt | grep mode
python - $O <<'PY'
import json,sys,os,glob
O=sys.argv[1]
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), j["ms_per_step"], j["stage_ms"])
    except Exception as e: print(f, "FAILED", e)
PY
f=$(find $O -name "kt_mip_kernel_stats.csv" | head -1); head -12 $f
