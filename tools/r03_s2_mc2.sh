export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_mc.py tests/test_gpu_mesh.py -m gpu -x -q -W ignore < /dev/null 2>&1 | tail -2
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -W ignore -k "bench_step and fused" < /dev/null 2>&1 | tail -2
bash tools/r03_s2_kt.sh $1
