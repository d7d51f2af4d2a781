# One family of evidence files per call: GPU suite summary, the three bench configs, rocprofv3 kernel stats and the two PMC
# passes of the default bench (same command line), joined by tools/summarize_pmc.py.  Usage (on the GPU box, via gpurun):
#   bash tools/profile_round.sh r02_v2 [--tests]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
if [ "$2" = "--tests" ]; then
  timeout -k 5 1500 python -m pytest tests -m gpu -q < /dev/null > $O/gpu_tests_full.txt 2>&1
  grep -E "passed|failed|error" $O/gpu_tests_full.txt | tail -3 > $O/gpu_tests.txt
fi
timeout -k 5 300 python bench.py < /dev/null > $O/bench.json 2> $O/bench.err
timeout -k 5 300 python bench.py --config mip < /dev/null > $O/bench_mip.json 2> $O/bench_mip.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu < /dev/null > $O/kt.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- python bench.py --steps 3 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
D=$(dirname $(find $O -name "kt_kernel_stats.csv" | head -1))
for f in fetch_counter_collection.csv write_counter_collection.csv; do s=$(find $O -name $f | head -1); [ -n "$s" ] && [ "$(dirname $s)" != "$D" ] && cp $s $D/; done
python tools/summarize_pmc.py $D $O/kernels_pmc.md $O/pmc_traffic.json auto < /dev/null | head -40
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o wskt -- python bench.py --config watershed --size 512 --steps 2 --warmup 1 --no-cpu < /dev/null > $O/wskt.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o skkt -- python bench.py --config watershed_sk --size 512 --steps 2 --warmup 1 --no-cpu < /dev/null > $O/skkt.log 2>&1
timeout -k 5 400 python bench.py --config watershed < /dev/null > $O/bench_watershed_1024.json 2> $O/bench_watershed.err
timeout -k 5 400 python bench.py --config watershed_sk < /dev/null > $O/bench_watershed_sk_1024.json 2> $O/bench_watershed_sk.err
timeout -k 5 400 python bench.py --config watershed_sk --ws-raw < /dev/null > $O/bench_watershed_sk_raw_1024.json 2> $O/bench_watershed_sk_raw.err
timeout -k 5 300 python tools/bench_wssk.py 512 < /dev/null > $O/wssk_512_modes.jsonl 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
cat $O/gpu_tests.txt 2>/dev/null
cat $O/bench.json
