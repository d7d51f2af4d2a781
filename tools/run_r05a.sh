cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_wssk.py tests/test_gpu_misc.py tests/test_golden_vectors.py tests/test_gpu_wsift.py -m gpu -q -x -W ignore 2>&1 | tail -5 > gpurun_out/r05a/tests_ws.txt
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "strong or bench_step" 2>&1 | tail -5 > gpurun_out/r05a/tests_strong.txt
timeout -k 5 900 python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err
timeout -k 5 300 python bench.py --scaling strong --no-others --no-cpu > gpurun_out/r05a/bench_strong1.json 2> gpurun_out/r05a/bench_strong1.err
timeout -k 5 600 python tools/bench_host.py > gpurun_out/r05a/bench_host.json 2> gpurun_out/r05a/bench_host.err
cat gpurun_out/r05a/tests_ws.txt gpurun_out/r05a/tests_strong.txt
tail -c 1500 gpurun_out/r05a/bench.err
