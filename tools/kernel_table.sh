cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --profile-steps 0 --host-input-steps 0 --extra-leg-steps 0 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/**/ks_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:32]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    print('%-52s %5s x %8.1f us  %5.2f%%'%(n[:52], int(r['Calls'])//25, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
