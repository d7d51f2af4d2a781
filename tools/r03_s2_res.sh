# round 3, session 2: resident rounds (k_flood_resident): flood tests, then A/B on the default bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_res_$1
mkdir -p $O
cd $R
if [ "$2" != "notest" ]; then
timeout -k 5 600 python -m pytest tests/test_gpu_flood.py tests/test_gpu_fused.py -m gpu -x -q -W ignore < /dev/null > $O/tests.txt 2>&1
tail -4 $O/tests.txt
fi
bash tools/r03_s2_env.sh IVX_FLOOD_APPLY_RIDE=0 - IVX_FLOOD_BATCH=1 IVX_FLOOD_BATCH=2
IVX_FLOOD_TRACE=1 timeout -k 5 120 python bench.py --no-cpu --steps 2 --warmup 1 < /dev/null 2>&1 | grep "resident launch" | tail -2
