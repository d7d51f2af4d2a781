cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_mc.py tests/test_gpu_fused.py tests/test_gpu_cranium.py tests/test_gpu_slab.py tests/test_gpu_headless.py -m gpu -q -x -W ignore 2>&1 | tail -15 > gpurun_out/r05b/tests_mc.txt
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "strong or bench_step or eight" 2>&1 | tail -15 > gpurun_out/r05b/tests_full.txt
timeout -k 5 300 python bench.py --no-others > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err
IVX_MC_ONE_LAUNCH=0 timeout -k 5 300 python bench.py --no-others --no-cpu > gpurun_out/r05b/bench_old.json 2> gpurun_out/r05b/bench_old.err
timeout -k 5 300 python bench.py --no-others --no-cpu --size 1024 --hbm-synth --steps 5 > gpurun_out/r05b/bench_1024.json 2> gpurun_out/r05b/bench_1024.err
IVX_MC_ONE_LAUNCH=0 timeout -k 5 300 python bench.py --no-others --no-cpu --size 1024 --hbm-synth --steps 5 > gpurun_out/r05b/bench_1024_old.json 2> gpurun_out/r05b/bench_1024_old.err
cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05b/prof -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-others > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r05b/prof -name "*_kernel_trace.csv" -delete
cat gpurun_out/r05b/tests_mc.txt gpurun_out/r05b/tests_full.txt
for f in bench bench_old bench_1024 bench_1024_old; do python - gpurun_out/r05b/$f.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["stage_ms"], j["roofline"]["per_stage_frac"], (j.get("parity") or {}).get("ok"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
tail -3 gpurun_out/r05b/bench.err
