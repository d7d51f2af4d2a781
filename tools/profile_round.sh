export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_v12
cd $R
timeout -k 5 400 python -m pytest tests -m gpu -q < /dev/null 2>&1 | tail -4 > gpurun_out/prof_v12/gpu_tests.txt
timeout -k 5 200 python bench.py < /dev/null > gpurun_out/prof_v12/bench.json 2> gpurun_out/prof_v12/bench.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v12 -o kt -- python bench.py --steps 5 --warmup 1 --cpu-slices 0 < /dev/null > gpurun_out/prof_v12/kt.log 2>&1
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_v12 -o fetch -- python bench.py --steps 3 --warmup 1 --cpu-slices 0 < /dev/null > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_v12 -o write -- python bench.py --steps 3 --warmup 1 --cpu-slices 0 < /dev/null > /dev/null 2>&1
find gpurun_out/prof_v12 -name "*.csv" | head -20
D=$(dirname $(find gpurun_out/prof_v12 -name "kt_kernel_stats.csv" | head -1))
echo "D=$D"
for f in fetch_counter_collection.csv write_counter_collection.csv; do s=$(find gpurun_out/prof_v12 -name $f | head -1); [ -n "$s" ] && [ "$(dirname $s)" != "$D" ] && cp $s $D/; done
python tools/summarize_pmc.py $D gpurun_out/prof_v12/kernels_pmc.md gpurun_out/prof_v12/pmc_traffic.json auto < /dev/null | head -30
timeout -k 5 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v12 -o mesh -- python tools/bench_mesh.py 512 < /dev/null > gpurun_out/prof_v12/mesh.log 2>&1
tail -1 gpurun_out/prof_v12/mesh.log | cut -c1-900
cat gpurun_out/prof_v12/gpu_tests.txt
cat gpurun_out/prof_v12/bench.json
