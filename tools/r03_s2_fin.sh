export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_fin
mkdir -p $O
cd $R
timeout -k 5 300 python -m pytest tests/test_gpu_wssk.py tests/test_gpu_wsift.py -m gpu -x -q -W ignore < /dev/null 2>&1 | grep -E "passed|failed|rror|assert" | tail -2
for c in watershed watershed_sk; do timeout -k 5 200 python bench.py --config $c --size 1024 --steps 2 --warmup 1 --no-cpu < /dev/null > $O/$c.json 2>/dev/null; python -c "
import sys,json
j=json.loads(open('$O/$c.json').read().strip().splitlines()[-1]); f=j['flood']; print('$c', j['ms_per_step'], {k:v for k,v in f.items() if k.startswith('us_')}, j['object_voxels'])"; done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu < /dev/null > $O/kt.log 2>&1
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- python bench.py --steps 3 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
D=$(dirname $(find $O -name "kt_kernel_stats.csv" | head -1))
for f in fetch_counter_collection.csv write_counter_collection.csv; do s=$(find $O -name $f | head -1); [ -n "$s" ] && [ "$(dirname $s)" != "$D" ] && cp $s $D/; done
python tools/summarize_pmc.py $D $O/kernels_pmc.md $O/pmc_traffic.json auto < /dev/null | tail -3
find $O -name "*_kernel_trace.csv" -size +4M -delete
find $O -name "*_counter_collection.csv" -size +4M -delete
