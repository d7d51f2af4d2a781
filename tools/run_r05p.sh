cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p
mkdir -p $O
for c in 8 16 32 64; do
IVX_FLOOD_ITCAP=$c timeout -k 5 300 python bench.py --config watershed --size 512 --no-cpu --steps 3 > $O/ws512_itcap$c.json 2> $O/ws512_itcap$c.err
python - $O/ws512_itcap$c.json itcap$c <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); fl=j["flood"]; print(sys.argv[2], j["stage_ms"]["flood"], {k:fl[k] for k in fl if k.startswith("us_") or k.startswith("cost_level")})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
