# per-launch durations of k_sk_round: bash tools/prof_wssk_trace.sh <size> <mode> <conn>
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${1:-512}; MODE=${2:-lut}; CONN=${3:-1}
OUT=/tmp/prof_tr
rm -rf $OUT; mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python tools/bench_wssk.py $N mode=$MODE conn=$CONN > $OUT/log.txt 2> $OUT/err.txt
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_tr/**/kt_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rr=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']), int(r.get('Grid_Size_X',r.get('Grid_Size',0))), i) for i,r in enumerate(rows) if 'k_sk_round' in r['Kernel_Name']]
print('launches',len(rr),'total ms',sum(d for d,_,_ in rr)/1e6)
import collections
bins=collections.Counter(); tot=collections.Counter()
for d,g,i in rr:
    b=0
    while (1<<b)*1000 < d: b+=1
    bins[b]+=1; tot[b]+=d
for b in sorted(bins): print('<= %6d us: %6d launches %9.2f ms'%(1<<b,bins[b],tot[b]/1e6))
print('top 15:',sorted(rr,reverse=True)[:15])
# time line: cumulative by index
half=len(rr)//10
for k in range(10):
    seg=rr[k*half:(k+1)*half]
    print('launch decile %d: %.2f ms, max %d us, mean grid %d'%(k,sum(d for d,_,_ in seg)/1e6,max(d for d,_,_ in seg)//1000,sum(g for _,g,_ in seg)//max(1,len(seg))))
PY
