cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e
mkdir -p $O
for d in 0 1 2 3 4 5; do
IVX_MC_FUSED_DBG=$d timeout -k 5 300 python bench.py --no-others --no-cpu --steps 10 > $O/bench_dbg$d.json 2> $O/bench_dbg$d.err
python - $O/bench_dbg$d.json $d <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("dbg", sys.argv[2], j["ms_per_step"], j["stage_ms"].get("mc_emit"), j["triangles"])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
