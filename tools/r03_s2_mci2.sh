export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_mc.py tests/test_gpu_mesh.py tests/test_gpu_slab.py -m gpu -x -q -W ignore < /dev/null 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -W ignore -k "2048" < /dev/null 2>&1 | grep -E "passed|failed|rror" | tail -2
timeout -k 5 600 python bench.py --config sharded2048 --no-cpu < /dev/null 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sharded2048', j['ms_per_step'], j['stage_ms'])"
