# round 3, watershed call 6: neighbour wake-up filter in the level floods
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ws_$1
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_wsift.py -m gpu -q < /dev/null > $O/tests.txt 2>&1
grep -E "passed|failed|error|Error|assert" $O/tests.txt | tail -8
run() { # config name env...
  c=$1; n=$2; shift; shift
  env "$@" timeout -k 5 300 python bench.py --config $c --size 512 --no-cpu < /dev/null > $O/bench_${c}_$n.json 2> $O/bench_${c}_$n.err
  python - "$O/bench_${c}_$n.json" $c $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f=j["flood"]
    print(sys.argv[2], sys.argv[3], "flood_ms", j["stage_ms"]["flood"], {k:f[k] for k in f if k.startswith("us_") or k.startswith("cost_") or k in ("rounds","tile_visits")})
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  tail -2 $O/bench_${c}_$n.err
}
run watershed f60 IVX_WS_LEVELS_FRAC=0.60
run watershed f70 IVX_WS_LEVELS_FRAC=0.70
run watershed f80 IVX_WS_LEVELS_FRAC=0.80
run watershed f90 IVX_WS_LEVELS_FRAC=0.90
run watershed f70b6 IVX_WS_LEVELS_FRAC=0.70 IVX_FLOOD_BATCH=6
run watershed f70it32 IVX_WS_LEVELS_FRAC=0.70 IVX_FLOOD_ITCAP=32
IVX_WS_LEVELS_FRAC=0.7 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --config watershed --size 512 --steps 2 --warmup 1 --no-cpu < /dev/null > $O/kt.log 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
python - $(find $O -name "kt_kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1000,1), round(float(r["TotalDurationNs"])/3e6,2), "ms/flood")
PY
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "watershed_ift" < /dev/null > $O/tests_full.txt 2>&1
grep -E "passed|failed|error|Error|differs" $O/tests_full.txt | tail -5
