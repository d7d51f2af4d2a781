cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_rays.py tests/test_gpu_mip.py tests/test_gpu_slab.py -m gpu -q -x -W ignore 2>&1 | tail -8 > $O/tests_rays.txt
timeout -k 5 300 python bench.py --config mip > $O/bench_mip.json 2> $O/bench_mip.err
cat $O/tests_rays.txt
python - $O/bench_mip.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(j["ms_per_step"], j.get("stage_ms"), j["roofline"]["frac"], (j.get("parity") or {}).get("ok"))
PY
