# round 3, A/B call 2: emit from the active-word list with count planes, closed tiles, adaptive round grids
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ab_$1
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_flood.py tests/test_gpu_mc.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_slab.py tests/test_gpu_cranium.py tests/test_gpu_holes.py tests/test_gpu_mesh.py -m gpu -x -q < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error" $O/tests.txt | tail -3
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 5 200 python bench.py --no-cpu < /dev/null > $O/bench_$n.json 2> $O/bench_$n.err
  python - "$O/bench_$n.json" $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], j["ms_per_step"], j["region_grow_rounds"], j["stage_ms"], j.get("region_grow_ms_min_med_max"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default IVX_X=0
run default2 IVX_X=0
run mclist IVX_X=0
run batch6 IVX_FLOOD_BATCH=6 
run batch2 IVX_FLOOD_BATCH=2 
timeout -k 5 300 python bench.py < /dev/null > $O/bench_full.json 2> $O/bench_full.err
python - "$O/bench_full.json" full <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], j["ms_per_step"], j["region_grow_rounds"], j["stage_ms"], j.get("region_grow_ms_min_med_max"), j["parity"]["ok"])
PY
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu < /dev/null > $O/kt.log 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
python - $(find $O -name "kt_kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1000,1), r["Percentage"])
PY
