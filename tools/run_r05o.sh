cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o
mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_mesh_tail.py tests/test_gpu_mesh.py tests/test_gpu_cranium.py tests/test_gpu_headless.py -m gpu -q -x -W ignore 2>&1 | tail -8 > $O/tests_mesh.txt
cat $O/tests_mesh.txt
