cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05c}
mkdir -p $O
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_mc.py tests/test_gpu_fused.py -m gpu -q -x -W ignore 2>&1 | tail -5 > $O/tests_mc.txt
timeout -k 5 300 python bench.py --no-others > $O/bench.json 2> $O/bench.err
timeout -k 5 300 python bench.py --no-others --no-cpu --size 1024 --hbm-synth --steps 5 > $O/bench_1024.json 2> $O/bench_1024.err
cat $O/tests_mc.txt
for f in bench bench_1024; do python - $O/$f.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["stage_ms"], j["roofline"]["per_stage_frac"], (j.get("parity") or {}).get("ok"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
tail -3 $O/bench.err
