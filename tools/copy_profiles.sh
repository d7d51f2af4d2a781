#!/bin/bash
# copies one family produced by tools/profile_round.sh (gpurun_out/prof_<tag>) into profiles/ under the round's names
# usage: bash tools/copy_profiles.sh <tag> <round prefix, e.g. r04>
set -e
O=gpurun_out/prof_$1; P=profiles; R=$2
[ -d "$O" ] || { echo "no $O"; exit 1; }
cp $O/bench.json $P/${R}_bench.json; cp $O/bench_mip.json $P/${R}_bench_mip.json; cp $O/dry_comm.json $P/${R}_dry_comm_1gpu.json
cp $O/kt_kernel_stats.csv $P/${R}_bench_kernel_stats.csv; cp $O/kernels_pmc.md $P/${R}_bench_kernels_pmc.md
f=$(find $O -name "fetch_counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $P/${R}_bench_pmc_fetch.csv
f=$(find $O -name "write_counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $P/${R}_bench_pmc_write.csv
cp $O/pmc_traffic.json $O/pmc_traffic_watershed.json $O/pmc_traffic_watershed_sk.json $P/
cp $O/pmc_traffic_watershed_1024.json $O/pmc_traffic_watershed_sk_1024.json $P/ 2>/dev/null || true
cp $O/watershed_1024_kernels_pmc.md $P/${R}_wsift_1024_kernels_pmc.md 2>/dev/null || true; cp $O/watershed_sk_1024_kernels_pmc.md $P/${R}_wssk_1024_kernels_pmc.md 2>/dev/null || true
cp $O/watershed_kt_kernel_stats.csv $P/${R}_wsift_512_kernel_stats.csv; cp $O/watershed_kernels_pmc.md $P/${R}_wsift_512_kernels_pmc.md
cp $O/watershed_sk_kt_kernel_stats.csv $P/${R}_wssk_512_kernel_stats.csv; cp $O/watershed_sk_kernels_pmc.md $P/${R}_wssk_512_kernels_pmc.md
for f in bench_watershed_512 bench_watershed_sk_512 bench_watershed_1024 bench_watershed_sk_1024 bench_sharded2048_1gpu; do cp $O/$f.json $P/${R}_$f.json; done
for f in bench_strong_1gpu bench_host_512 bench_mc_one_launch bench_watershed_512_nolinks; do [ -f $O/$f.json ] && cp $O/$f.json $P/${R}_$f.json; done
[ -f $O/gpu_tests.txt ] && cp $O/gpu_tests.txt $P/${R}_gpu_tests.txt
[ -f $O/force_slab.json ] && cp $O/force_slab.json $P/${R}_force_slab.json
for f in bench_surface_tail bench_mesh_512 bench_edit_512; do [ -f $O/$f.json ] && cp $O/$f.json $P/${R}_$f.json; done
#  a host-call file made from other sources than the tree's is not a family member: refuse it
if [ -f $O/bench_host_512.json ]; then
  python3 - $O/bench_host_512.json <<'PY' || { echo "bench_host_512.json was not made from these sources: not copied"; rm -f $P/${R}_bench_host_512.json; }
import hashlib, json, os, sys
h = hashlib.sha256()
for d in ("invesalius3_amd/csrc", "invesalius3_amd"):
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".py")):
            h.update(open(os.path.join(d, f), "rb").read())
sys.exit(0 if json.load(open(sys.argv[1])).get("sources_sha16") == h.hexdigest()[:16] else 1)
PY
fi
for k in slab_kt mip_kt stitch_kt tail_kt; do f=$(find $O -name "${k}_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${R}_${k}_kernel_stats.csv; done
echo "copied $O -> $P/${R}_*  (re-run 'python bench.py' afterwards for a line that quotes the new pmc_traffic.json)"
