# round 3, session 2: marching-cubes changes (emit: host-divided interpolation factors, 16-byte stores, dword table staging; list: LDS-staged descriptors)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_mc_$1
mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_mc.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_mesh.py tests/test_gpu_cranium.py -m gpu -x -q -W ignore < /dev/null > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for i in 1 2; do
timeout -k 5 300 python bench.py --no-cpu < /dev/null > $O/bench_$i.json 2> $O/bench_$i.err
python - $O/bench_$i.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], j["stage_ms"], j["parity"]["ok"] if j.get("parity") else None)
PY
done
