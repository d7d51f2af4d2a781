cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05h}
mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_wsift.py tests/test_gpu_misc.py -m gpu -q -x -W ignore 2>&1 | tail -5 > $O/tests_ws.txt
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -W ignore -k "watershed_ift" 2>&1 | tail -5 > $O/tests_full.txt
for l in 1 0; do
IVX_WS_LINKS=$l timeout -k 5 300 python bench.py --config watershed --size 512 --no-cpu --steps 3 > $O/ws512_links$l.json 2> $O/ws512_links$l.err
IVX_WS_LINKS=$l timeout -k 5 300 python bench.py --config watershed --no-cpu --steps 2 > $O/ws1024_links$l.json 2> $O/ws1024_links$l.err
done
cat $O/tests_ws.txt $O/tests_full.txt
for f in ws512_links1 ws512_links0 ws1024_links1 ws1024_links0; do python - $O/$f.json $f <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fl=j["flood"]; print(sys.argv[2], j["stage_ms"], {k:fl[k] for k in fl if k.startswith("us_")})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
