# round 3: marching cubes from the mask's known byte levels (ivx_dev_mc_emit_levels): parity + A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_mc_$1
mkdir -p $O
cd $R
timeout -k 5 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_mc.py tests/test_gpu_slab.py tests/test_gpu_cranium.py tests/test_gpu_headless.py -m gpu -q -W ignore < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error|Error|assert" $O/tests.txt | tail -8
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "bench_step or eight_slabs" < /dev/null > $O/tests_full.txt 2>&1
grep -E "passed|failed|error|Error" $O/tests_full.txt | tail -5
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
run levels IVX_X=0
run gathers IVX_MC_LEVELS=0
run levels2 IVX_X=0
timeout -k 5 300 python bench.py < /dev/null > $O/bench_full.json 2> $O/bench_full.err
python - "$O/bench_full.json" full <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], j["ms_per_step"], j["region_grow_rounds"], j["stage_ms"], j["parity"]["ok"], j["roofline"]["per_stage_frac"])
PY
