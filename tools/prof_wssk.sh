# rocprofv3 kernel stats of the scikit-image watershed flood: bash tools/prof_wssk.sh <size> <mode> <conn> [tag]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${1:-512}; MODE=${2:-lut}; CONN=${3:-1}; TAG=${4:-sk}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python tools/bench_wssk.py $N mode=$MODE conn=$CONN > $OUT/log.txt 2> $OUT/err.txt
tail -2 $OUT/log.txt
S=$(find $OUT -name "kt_kernel_stats.csv" | head -1)
cp $S $OUT/kernel_stats_${N}_${MODE}_${CONN}.csv
python - "$S" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    n=r['Name']
    for a in ('void ','(anonymous namespace)::','rocprim::detail::','rocprim::ROCPRIM_400000_NS::detail::'): n=n.replace(a,'')
    print('%-60s calls %6s avg %10.1f us tot %9.2f ms %6s%%'%(n[:60],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6,r['Percentage'][:6]))
PY
find $OUT -name "*.csv" -size +2M -delete
