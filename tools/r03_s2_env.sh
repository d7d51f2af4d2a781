# A/B of environment switches on the default bench: each argument is one "VAR=value[,VAR=value]" set ("-" = defaults)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for cfg in "$@"; do
  envs=$(echo $cfg | tr ',' ' ')
  [ "$cfg" = "-" ] && envs=""
  for i in 1 2; do
  env $envs timeout -k 5 300 python bench.py --no-cpu < /dev/null 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', j['ms_per_step'], j['region_grow_rounds'], j['stage_ms'], j.get('region_grow_ms_min_med_max'))"
  done
done
