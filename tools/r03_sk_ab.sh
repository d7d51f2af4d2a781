timeout 600 python -m pytest tests/test_gpu_wssk.py -m gpu -x -q -W ignore 2>&1 | tail -2
for cfg in "1024 1" "512 1" "256 1" "512 2" "256 2" "256 4"; do set -- $cfg
IVX_SK_PER_WG=$1 IVX_SK_RES_PER_CU=$2 timeout 200 python bench.py --config watershed_sk --size 512 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per_wg $1 res $2',j['ms_per_step'])"
done
