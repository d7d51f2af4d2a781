export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_wsift.py tests/test_golden_vectors.py -m gpu -x -q -W ignore -k "ift or watershed" < /dev/null 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
for sz in 512 1024; do timeout -k 5 300 python bench.py --config watershed --size $sz --steps 3 --warmup 1 --no-cpu < /dev/null 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['flood']; print('watershed $sz', j['ms_per_step'], {k:v for k,v in f.items() if k.startswith('us_')}, j['object_voxels'])"; done
