# per-kernel rocprofv3 statistics of the default bench's graph-replayed steps only (every extra leg off), top rows + rows matching $1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks2
GENESIS_BENCH_LONG_STEPS=0 rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o ks --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --cpu-seconds 0 --profile-steps 0 --host-input-steps 0 --extra-leg-steps 0 --fp32-pipe-steps 0 > /dev/null 2>&1
python - "$1" <<'PY'
import csv,glob,sys,re
pat=sys.argv[1] if len(sys.argv)>1 and sys.argv[1] else None
f=glob.glob('/tmp/ks2/**/ks_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=48.0   # 40 timed + 5 warm-up + 3 eager warm-up iterations of the capture
print('total kernel time per step ~ %.1f us' % (tot/steps/1e3))
for i,r in enumerate(rows):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    if i<28 or (pat and re.search(pat,n)):
        print('%-60s %7.2f x %8.1f us  %5.2f%%'%(n[:60], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
