cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05s
timeout -k 5 600 python tools/bench_host.py > gpurun_out/r05s/bench_host_512.json 2> gpurun_out/r05s/bench_host.err
tail -40 gpurun_out/r05s/bench_host_512.json
