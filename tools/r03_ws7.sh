# round 3, watershed call 7: scikit-image branch's cost levels: parity tests + timing (windowed and raw), IFT default
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ws_$1
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_wssk.py -m gpu -q -W ignore < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error|Error|assert" $O/tests.txt | tail -8
run() { # config name extra env...
  c=$1; n=$2; x=$3; shift; shift; shift
  env "$@" timeout -k 5 300 python bench.py --config $c --size 512 --no-cpu $x < /dev/null > $O/bench_${c}_$n.json 2> $O/bench_${c}_$n.err
  python - "$O/bench_${c}_$n.json" $c $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f=j["flood"]
    print(sys.argv[2], sys.argv[3], "flood_ms", j["stage_ms"]["flood"], {k:f[k] for k in f if k.startswith("us_") or k.startswith("cost_") or k in ("rounds","tile_visits")}, j["object_voxels"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  tail -2 $O/bench_${c}_$n.err
}
run watershed_sk off "" IVX_SK_LEVELS=0
run watershed_sk on "" IVX_X=0
run watershed_sk on3f99 "" IVX_SK_LEVELS_FRAC=0.99 IVX_SK_LEVELS=3
run watershed_sk rawoff "--ws-raw" IVX_SK_LEVELS=0
run watershed_sk rawon "--ws-raw" IVX_X=0
run watershed dflt "" IVX_X=0
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "watershed" < /dev/null > $O/tests_full.txt 2>&1
grep -E "passed|failed|error|Error|differs" $O/tests_full.txt | tail -5
