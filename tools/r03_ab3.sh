# round 3, call 3: device stitch (2/3-slab loop-back + 8-slab configs[3] geometry), sharded2048 bench on one GPU, dry-comm, default bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ab_$1
mkdir -p $O
cd $R
timeout -k 5 1200 python -m pytest tests/test_gpu_slab.py tests/test_gpu_fullsize.py tests/test_gpu_mc.py tests/test_gpu_mesh.py -m gpu -x -q < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error|Error" $O/tests.txt | tail -5
timeout -k 5 120 python bench.py --dry-comm < /dev/null > $O/dry_comm.json 2> $O/dry_comm.err; cat $O/dry_comm.json; tail -3 $O/dry_comm.err
timeout -k 5 600 python bench.py --config sharded2048 < /dev/null > $O/bench_sharded2048.json 2> $O/bench_sharded2048.err
python - "$O/bench_sharded2048.json" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("sharded2048", j["ms_per_step"], j["stage_ms"], j["stitch"], j.get("parity"), j["cpu_baseline"])
except Exception as e:
    print("sharded FAILED", e)
PY
tail -3 $O/bench_sharded2048.err
timeout -k 5 300 python bench.py < /dev/null > $O/bench_full.json 2> $O/bench_full.err
python - "$O/bench_full.json" full <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], j["ms_per_step"], j["region_grow_rounds"], j["stage_ms"], j.get("region_grow_ms_min_med_max"), j["parity"]["ok"])
PY
