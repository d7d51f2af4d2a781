export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wsift.py tests/test_gpu_wssk.py tests/test_golden_vectors.py -m gpu -q -x 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "watershed" 2>&1 | tail -5
for v in 1 0; do
  IVX_WS_UNION_LOCAL=$v timeout 300 python bench.py --config watershed --size 512 --no-cpu --steps 3 > $O/ift512_u$v.json 2>> $O/err.txt
  IVX_WS_UNION_LOCAL=$v timeout 300 python bench.py --config watershed_sk --size 512 --no-cpu --steps 3 > $O/sk512_u$v.json 2>> $O/err.txt
  IVX_WS_UNION_LOCAL=$v timeout 300 python bench.py --config watershed --no-cpu --steps 2 > $O/ift1024_u$v.json 2>> $O/err.txt
  IVX_WS_UNION_LOCAL=$v timeout 300 python bench.py --config watershed_sk --no-cpu --steps 2 > $O/sk1024_u$v.json 2>> $O/err.txt
done
python - $O <<'PY'
import json,sys,os,glob
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); fl=j["flood"]; print(os.path.basename(f), j["ms_per_step"], {k:v for k,v in fl.items() if k.startswith("us_")})
    except Exception as e: print(f, "FAILED", e)
PY
tail -5 $O/err.txt
