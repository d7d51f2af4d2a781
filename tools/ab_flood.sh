# the bench step and the same step past the cache (each run its own process)
run() { timeout 300 python bench.py --no-others --no-cpu $X < /dev/null 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j.get('stage_ms'), j['roofline'].get('per_stage_frac'))"; }
X=""; echo 512; run; run
X="--size 1024 --hbm-synth"; echo 1024; run
