# rocprofv3 kernel stats of one command, summary copied next to the log.  Usage (on the GPU box): bash tools/prof_kt.sh OUTDIR NAME cmd...
# (everything under a timeout, stdin closed: a gpurun call must never wait for input)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
N=$2
shift 2
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 5 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$N -o $N -- "$@" < /dev/null > $O/$N.log 2>&1
f=$(find $O/kt_$N -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then
  cp "$f" $O/${N}_kernel_stats.csv
  python tools/kt_summary.py $O/${N}_kernel_stats.csv ${KT_DIV:-1} < /dev/null | head -${KT_LINES:-25}
else
  echo "no kernel stats"; tail -5 $O/$N.log
fi
find $O/kt_$N -name "*_kernel_trace.csv" -size +8M -delete 2>/dev/null
