# configs[1]'s step past the Infinity Cache: rocprofv3 kernel stats + the two PMC passes of `bench.py --size 1024 --hbm-synth`
# (the volume is made in HBM by k_synth).  Usage (GPU box): bash tools/prof_step_1024.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof1024_${1:-r05}
mkdir -p $O
cd /tmp
CMD="python $R/bench.py --size 1024 --hbm-synth --steps 5 --warmup 1 --no-cpu --no-others"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- $CMD < /dev/null > $O/kt.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- $CMD < /dev/null > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- $CMD < /dev/null > /dev/null 2>&1
cd $R
D=$(dirname $(find $O -name "kt_kernel_stats.csv" | head -1))
for f in fetch_counter_collection.csv write_counter_collection.csv; do s=$(find $O -name $f | head -1); [ -n "$s" ] && [ "$(dirname $s)" != "$D" ] && cp $s $D/; done
python tools/summarize_pmc.py $D $O/kernels_pmc.md $O/pmc_traffic_1024.json auto < /dev/null | head -30
cp $D/kt_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_counter_collection.csv" -size +8M -delete
cat $O/pmc_traffic_1024.json
