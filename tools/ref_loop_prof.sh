cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GENESIS_AUTOSTEP=0 python $R/tools/ref_loop_time.py 40
GENESIS_AUTOSTEP=1 python $R/tools/ref_loop_time.py 40
rm -rf /tmp/rl; rocprofv3 --kernel-trace --stats -d /tmp/rl -o p --output-format csv -- python $R/tools/ref_loop_time.py 40 2>&1 | grep "reference loop"
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/rl/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('kernel time %.3f ms / iteration, %.0f launches / iteration (45 iterations traced)'%(tot/45/1e6, calls/45))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print('   %-70s %7.1f us/it  %5.1f calls'%(r['Name'].replace('(anonymous namespace)::','')[:70], float(r['TotalDurationNs'])/45/1e3, int(r['Calls'])/45))
PY
