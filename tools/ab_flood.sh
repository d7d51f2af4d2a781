# A/B of the region-growing engine's knobs on the bench step (each run its own process: the knobs are read once)
run() { timeout 300 python bench.py --no-others --no-cpu $X < /dev/null 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j.get('stage_ms'), j.get('region_grow_rounds'))"; }
X=""; echo default; run
X="--config watershed --size 512"; echo ift512; run
X="--config watershed_sk --size 512"; echo sk512; run
