# One family of evidence files per call: GPU suite summary, every bench config, rocprofv3 kernel stats and the two PMC passes of
# the default bench and of both 512^3 watershed floods (same command lines), joined by tools/summarize_pmc.py / summarize_ws_pmc.py.
# Usage (on the GPU box, via gpurun):   bash tools/profile_round.sh r03_v1 [--tests]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
if [ "$2" = "--tests" ]; then
  timeout -k 5 2400 python -m pytest tests -m gpu -q -W ignore < /dev/null > $O/gpu_tests_full.txt 2>&1
  grep -E "passed|failed|error" $O/gpu_tests_full.txt | tail -3 > $O/gpu_tests.txt
fi
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu < /dev/null > $O/kt.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- python bench.py --steps 3 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
D=$(dirname $(find $O -name "kt_kernel_stats.csv" | head -1))
for f in fetch_counter_collection.csv write_counter_collection.csv; do s=$(find $O -name $f | head -1); [ -n "$s" ] && [ "$(dirname $s)" != "$D" ] && cp $s $D/; done
python tools/summarize_pmc.py $D $O/kernels_pmc.md $O/pmc_traffic.json auto < /dev/null | head -40
cp $O/pmc_traffic.json profiles/ 2>/dev/null  # (so that the line below quotes it: same sources, same box)
timeout -k 5 900 python bench.py < /dev/null > $O/bench.json 2> $O/bench.err
timeout -k 5 300 python bench.py --scaling strong --no-others --no-cpu < /dev/null > $O/bench_strong_1gpu.json 2> $O/bench_strong_1gpu.err
timeout -k 5 600 python tools/bench_host.py < /dev/null > $O/bench_host_512.json 2> $O/bench_host_512.err
# the opt-in paths kept for A/B: marching cubes in one launch, the IFT level chain without link records
IVX_MC_ONE_LAUNCH=1 timeout -k 5 300 python bench.py --no-others --no-cpu < /dev/null > $O/bench_mc_one_launch.json 2> /dev/null
IVX_WS_LINKS=0 timeout -k 5 300 python bench.py --config watershed --size 512 --no-cpu < /dev/null > $O/bench_watershed_512_nolinks.json 2> /dev/null
timeout -k 5 300 python bench.py --config mip < /dev/null > $O/bench_mip.json 2> $O/bench_mip.err
timeout -k 5 120 python bench.py --dry-comm < /dev/null > $O/dry_comm.json 2> $O/dry_comm.err
# the stages behind marching cubes on the bench surface (bench line + kernel stats), the remaining edit / resample kernels
timeout -k 5 300 python bench.py --config surface_tail < /dev/null > $O/bench_surface_tail.json 2> $O/bench_surface_tail.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o tail_kt -- python bench.py --config surface_tail --steps 3 < /dev/null > $O/tail_kt.log 2>&1
timeout -k 5 300 python tools/bench_mesh.py < /dev/null > $O/bench_mesh_512.json 2> $O/bench_mesh_512.err
timeout -k 5 300 python tools/bench_edit.py < /dev/null > $O/bench_edit_512.json 2> $O/bench_edit_512.err
# the sharded path's own overhead against the resident single volume: same step through SlabVolume at world 1
IVX_FORCE_SLAB=1 timeout -k 5 300 python bench.py --no-cpu < /dev/null > $O/force_slab.json 2> $O/force_slab.err
IVX_FORCE_SLAB=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o slab_kt -- python bench.py --steps 5 --warmup 1 --no-cpu < /dev/null > $O/slab_kt.log 2>&1
# configs[4]'s kernels, and the device cost of the cross-slab stitch where it has work: 8 loop-back slabs of configs[3]'s geometry
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o mip_kt -- python bench.py --config mip --steps 5 --warmup 1 --no-cpu < /dev/null > $O/mip_kt.log 2>&1
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stitch_kt -- python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k eight_slabs < /dev/null > $O/stitch_kt.log 2>&1
for c in watershed watershed_sk; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${c}_kt -- python bench.py --config $c --size 512 --steps 2 --warmup 1 --no-cpu < /dev/null > $O/${c}_kt.log 2>&1
  timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o ${c}_fetch -- python bench.py --config $c --size 512 --steps 2 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
  timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o ${c}_write -- python bench.py --config $c --size 512 --steps 2 --warmup 1 --no-cpu < /dev/null > /dev/null 2>&1
  python tools/summarize_ws_pmc.py $c $(find $O -name "${c}_fetch_counter_collection.csv" | head -1) $(find $O -name "${c}_write_counter_collection.csv" | head -1) $O/pmc_traffic_$c.json $O/${c}_kernels_pmc.md < /dev/null | tail -12
done
# ... and the same two counters at configs[2]'s stated size (one flood per pass)
for c in watershed watershed_sk; do
  timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o ${c}_1024_fetch -- python bench.py --config $c --steps 1 --warmup 0 --no-cpu < /dev/null > /dev/null 2>&1
  timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o ${c}_1024_write -- python bench.py --config $c --steps 1 --warmup 0 --no-cpu < /dev/null > /dev/null 2>&1
  python tools/summarize_ws_pmc.py $c $(find $O -name "${c}_1024_fetch_counter_collection.csv" | head -1) $(find $O -name "${c}_1024_write_counter_collection.csv" | head -1) $O/pmc_traffic_${c}_1024.json $O/${c}_1024_kernels_pmc.md 1024 < /dev/null | tail -4
done
cp $O/pmc_traffic_watershed.json $O/pmc_traffic_watershed_sk.json $O/pmc_traffic_watershed_1024.json $O/pmc_traffic_watershed_sk_1024.json profiles/ 2>/dev/null  # (so that the lines below quote them)
timeout -k 5 400 python bench.py --config watershed --size 512 < /dev/null > $O/bench_watershed_512.json 2> $O/bench_watershed_512.err
timeout -k 5 400 python bench.py --config watershed_sk --size 512 < /dev/null > $O/bench_watershed_sk_512.json 2> $O/bench_watershed_sk_512.err
timeout -k 5 400 python bench.py --config watershed < /dev/null > $O/bench_watershed_1024.json 2> $O/bench_watershed.err
timeout -k 5 400 python bench.py --config watershed_sk < /dev/null > $O/bench_watershed_sk_1024.json 2> $O/bench_watershed_sk.err
timeout -k 5 600 python bench.py --config sharded2048 < /dev/null > $O/bench_sharded2048_1gpu.json 2> $O/bench_sharded2048.err
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*_counter_collection.csv" -size +8M -delete
cat $O/gpu_tests.txt 2>/dev/null
for f in bench bench_strong_1gpu bench_mc_one_launch bench_mip bench_surface_tail bench_watershed_512 bench_watershed_512_nolinks bench_watershed_sk_512 bench_watershed_1024 bench_watershed_sk_1024 bench_sharded2048_1gpu; do
python - $O/$f.json $f <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", j["ms_per_step"], "stage_ms", j.get("stage_ms"), "frac", j["roofline"]["frac"], "traffic", j["roofline"].get("traffic"), "parity", (j.get("parity") or {}).get("ok"), "e2e", j.get("end_to_end_ms"), j.get("end_to_end_pinned_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
