# environment sweeps at 1024^3 (configs[2]'s size): the IFT cost levels' share, the scikit-image branch's resident grid
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() { c=$1; shift; env "$@" timeout -k 5 200 python bench.py --config $c --size 1024 --steps 2 --warmup 1 --no-cpu < /dev/null 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['flood']; print('$c', '$*', j['ms_per_step'], {k:v for k,v in f.items() if k.startswith('us_') or k in ('cost_levels','cost_level_rounds')})"; }
run watershed IVX_WS_LEVELS_FRAC=0.5
run watershed IVX_WS_LEVELS_FRAC=0.8
run watershed IVX_WS_LEVELS_FRAC=0.9
run watershed_sk IVX_SK_RES_PER_CU=2
run watershed_sk IVX_SK_RES_PER_CU=2 IVX_SK_PER_WG=512
run watershed_sk IVX_SK_PER_WG=2048
