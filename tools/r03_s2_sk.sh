# round 3, session 2: k_sk_level with the solo stretch: tests, then A/B on the 512^3 flood
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2_sk_$1
mkdir -p $O
cd $R
if [ "$2" != "notest" ]; then
timeout -k 5 900 python -m pytest tests/test_gpu_wssk.py -m gpu -x -q -W ignore < /dev/null > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -W ignore -k "gui_default" < /dev/null > $O/tests_full.txt 2>&1
tail -3 $O/tests_full.txt
fi
shift; shift
for cfg in "$@"; do
  envs=$(echo $cfg | tr ',' ' ')
  [ "$cfg" = "-" ] && envs=""
  env $envs timeout -k 5 200 python bench.py --config watershed_sk --size 512 --steps 3 --warmup 1 --no-cpu < /dev/null 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=j['flood']
print('$cfg', j['ms_per_step'], 'levels_us', f['us_levels'], 'gens', f['generations'], 'steps', f['generation_steps'], 'brounds', f['basin_rounds'], 'obj', j['object_voxels'])"
done
