# round 3, watershed call 1: cost-relaxation gate policies (IVX_WS_GATE) on the 512^3 floods; the full-volume parity tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ws_$1
mkdir -p $O
cd $R
run() { # config name env...
  c=$1; n=$2; shift; shift
  env "$@" timeout -k 5 300 python bench.py --config $c --size 512 --no-cpu < /dev/null > $O/bench_${c}_$n.json 2> $O/bench_${c}_$n.err
  python - "$O/bench_${c}_$n.json" $c $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f=j["flood"]
    print(sys.argv[2], sys.argv[3], "flood_ms", j["stage_ms"]["flood"], "rounds", f["rounds"], "visits", f["tile_visits"], "us_costs", f["us_costs"], "object", j["object_voxels"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
}
for g in 0 2 12 14 16 20 28; do run watershed g$g IVX_WS_GATE=$g; done
for g in 0 2 1 8 32; do run watershed_sk g$g IVX_WS_GATE=$g; done
timeout -k 5 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error|Error|differs_from_reference" $O/tests.txt | tail -8
