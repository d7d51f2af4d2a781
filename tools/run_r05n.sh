cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05n}
mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_wsift.py tests/test_gpu_wssk.py -m gpu -q -x -W ignore 2>&1 | tail -3 > $O/tests_ws.txt
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -W ignore -k "v512_watershed" 2>&1 | tail -3 > $O/tests_full.txt
timeout -k 5 300 python bench.py --config watershed --size 512 --no-cpu --steps 3 > $O/ws512.json 2> $O/ws512.err
timeout -k 5 300 python bench.py --config watershed_sk --size 512 --no-cpu --steps 3 > $O/sk512.json 2> $O/sk512.err
cat $O/tests_ws.txt $O/tests_full.txt
for f in ws512 sk512; do python - $O/$f.json $f <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fl=j["flood"]; print(sys.argv[2], j["stage_ms"], {k:fl[k] for k in fl if k.startswith("us_")})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
