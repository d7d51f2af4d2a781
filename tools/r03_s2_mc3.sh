export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_mc.py tests/test_gpu_fused.py tests/test_gpu_mesh.py tests/test_gpu_cranium.py tests/test_gpu_slab.py tests/test_golden_vectors.py -m gpu -x -q -W ignore < /dev/null 2>&1 | grep -E "passed|failed|rror|assert" | tail -6
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -W ignore -k "bench_step" < /dev/null 2>&1 | grep -E "passed|failed|rror" | tail -2
bash tools/r03_s2_env.sh IVX_MC_SCAN_FUSED=0 -
