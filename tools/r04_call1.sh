# round 4, GPU call 1: full GPU suite + default bench (with other_configs) + the per-config lines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c1
mkdir -p $O
cd $R
nproc > $O/host.txt; lscpu | grep -i "model name" >> $O/host.txt; free -g | sed -n 2p >> $O/host.txt
timeout -k 5 1500 python -m pytest tests -m gpu -q -W ignore -x --durations=15 < /dev/null > $O/gpu_tests_full.txt 2>&1
tail -25 $O/gpu_tests_full.txt > $O/gpu_tests.txt
( time timeout -k 5 600 python bench.py < /dev/null > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
timeout -k 5 300 python bench.py --config mip < /dev/null > $O/bench_mip.json 2> $O/bench_mip.err
timeout -k 5 400 python bench.py --config watershed --size 512 < /dev/null > $O/bench_watershed_512.json 2> $O/bench_watershed_512.err
timeout -k 5 400 python bench.py --config watershed_sk --size 512 < /dev/null > $O/bench_watershed_sk_512.json 2> $O/bench_watershed_sk_512.err
tail -5 $O/gpu_tests.txt; cat $O/bench.time | tail -4
python - $O <<'PY'
import json,sys,os
O=sys.argv[1]
for f in ("bench","bench_mip","bench_watershed_512","bench_watershed_sk_512"):
    try:
        j=json.loads(open(os.path.join(O,f+".json")).read().strip().splitlines()[-1])
        print(f, "ms", j["ms_per_step"], "stage", j.get("stage_ms"), "frac", j["roofline"]["frac"], "parity", (j.get("parity") or {}).get("ok"), "dfr", j.get("differs_from_reference"), "e2e", j.get("end_to_end_ms"), j.get("end_to_end_pinned_ms"))
        if "other_configs" in j:
            for k,v in j["other_configs"].items(): print("   ", k, {a:v.get(a) for a in ("ms","frac","parity_ok","differs_from_reference","wall_s","error")})
    except Exception as e:
        print(f, "FAILED", e); print(open(os.path.join(O,f+".err")).read()[-1500:])
PY
