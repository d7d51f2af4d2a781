# round 3, watershed call 2: k_sk_round grid / batch sizes (IVX_SK_MIN_GRID, IVX_SK_BATCH0) on the 512^3 GUI-default flood; configs[4]
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
    print(sys.argv[2], sys.argv[3], "flood_ms", j["stage_ms"]["flood"], {k:f[k] for k in f if k.startswith("us_") or k in ("frontier_launches","generation_steps","basin_rounds")}, "object", j["object_voxels"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
}
run watershed_sk old IVX_SK_MIN_GRID=1024 IVX_SK_BATCH0=16
run watershed_sk g128b8 IVX_X=0
run watershed_sk g64b8 IVX_SK_MIN_GRID=64
run watershed_sk g256b8 IVX_SK_MIN_GRID=256
run watershed_sk g128b4 IVX_SK_BATCH0=4
run watershed_sk g128b16 IVX_SK_BATCH0=16
run watershed_sk g128b8p512 IVX_SK_PER_WG=512
timeout -k 5 300 python bench.py --config mip < /dev/null > $O/bench_mip.json 2> $O/bench_mip.err
tail -c 2500 $O/bench_mip.json; tail -3 $O/bench_mip.err
timeout -k 5 600 python -m pytest tests/test_gpu_wssk.py tests/test_golden_vectors.py -m gpu -x -q < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error" $O/tests.txt | tail -3
