# round 3, session 2, call 1: where the scikit-image branch's level chain spends its time after k_sk_level (kernel stats + per-level trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_diag
mkdir -p $O
cd $R
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o sk_kt -- python bench.py --config watershed_sk --size 512 --steps 2 --warmup 1 --no-cpu < /dev/null > $O/sk_kt.log 2>&1
IVX_WS_TRACE=1 timeout -k 5 200 python bench.py --config watershed_sk --size 512 --steps 1 --warmup 1 --no-cpu < /dev/null > $O/sk_trace.json 2> $O/sk_trace.err
timeout -k 5 200 python bench.py --config watershed_sk --size 512 --steps 3 --warmup 1 --no-cpu < /dev/null > $O/sk_512.json 2> $O/sk_512.err
IVX_WS_TRACE=1 timeout -k 5 200 python bench.py --config watershed --size 512 --steps 1 --warmup 1 --no-cpu < /dev/null > $O/ift_trace.json 2> $O/ift_trace.err
find $O -name "*_kernel_trace.csv" -size +8M -delete
python - $O/sk_512.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], j["stage_ms"], j["flood"])
PY
