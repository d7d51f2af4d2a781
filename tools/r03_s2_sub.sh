# round 3, session 2: round lists sharded into NSUB sublists (one counter per sublist): flood suites, then the bench and the watershed floods
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_flood.py tests/test_gpu_fused.py tests/test_gpu_slab.py tests/test_gpu_wsift.py tests/test_gpu_holes.py -m gpu -x -q -W ignore < /dev/null 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -W ignore -k "bench_step or 2048" < /dev/null 2>&1 | grep -E "passed|failed|rror" | tail -2
bash tools/r03_s2_env.sh -
for c in watershed watershed_sk; do timeout -k 5 300 python bench.py --config $c --size 512 --steps 3 --warmup 1 --no-cpu < /dev/null 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', j['ms_per_step'], j['flood'].get('us_costs'))"; done
