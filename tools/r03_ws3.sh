# round 3, watershed call 3: the cost map's level floods (ivx_dev_ws_cost_levels): parity tests, then the 512^3 flood by share
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ws_$1
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_wsift.py -m gpu -x -q < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error|Error|assert" $O/tests.txt | tail -8
run() { # config name env...
  c=$1; n=$2; shift; shift
  env "$@" timeout -k 5 300 python bench.py --config $c --size 512 --no-cpu < /dev/null > $O/bench_${c}_$n.json 2> $O/bench_${c}_$n.err
  python - "$O/bench_${c}_$n.json" $c $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f=j["flood"]
    print(sys.argv[2], sys.argv[3], "flood_ms", j["stage_ms"]["flood"], {k:f[k] for k in f if k.startswith("us_") or k.startswith("cost_") or k in ("rounds","tile_visits")}, "object", j["object_voxels"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  tail -2 $O/bench_${c}_$n.err
}
run watershed off IVX_WS_LEVELS=0
run watershed f60 IVX_WS_LEVELS_FRAC=0.6
run watershed f80 IVX_WS_LEVELS_FRAC=0.8
run watershed f90 IVX_WS_LEVELS_FRAC=0.9
run watershed f95 IVX_WS_LEVELS_FRAC=0.95
run watershed f98 IVX_WS_LEVELS_FRAC=0.98
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "watershed_ift" < /dev/null > $O/tests_full.txt 2>&1
grep -E "passed|failed|error|Error|differs" $O/tests_full.txt | tail -5
