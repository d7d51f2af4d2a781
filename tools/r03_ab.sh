# round 3, first A/B call: focused GPU tests of the touched paths, then the default bench with each new path switched off in turn
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ab_$1
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_flood.py tests/test_gpu_mc.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_slab.py tests/test_gpu_cranium.py tests/test_gpu_holes.py tests/test_gpu_mesh.py -m gpu -x -q < /dev/null > $O/tests.txt 2>&1
tail -5 $O/tests.txt
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 5 200 python bench.py --no-cpu < /dev/null > $O/bench_$n.json 2> $O/bench_$n.err
  python - "$O/bench_$n.json" $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], j["ms_per_step"], j["region_grow_rounds"], j["stage_ms"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default IVX_X=0
run block64 IVX_FLOOD_BLOCK=64
run unfused IVX_FLOOD_FUSED=0
run mclist IVX_MC_LIST=1
run old IVX_FLOOD_BLOCK=64 IVX_FLOOD_FUSED=0 IVX_MC_LIST=1
timeout -k 5 300 python bench.py < /dev/null > $O/bench_full.json 2> $O/bench_full.err
tail -c 1500 $O/bench_full.json
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu < /dev/null > $O/kt.log 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "kt_kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | head -25'
