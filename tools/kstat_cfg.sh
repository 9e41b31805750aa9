# per-kernel rocprofv3 statistics of one bench configuration's graph-replayed steps (every extra leg off):
#   bash tools/kstat_cfg.sh "<bench.py arguments>" [rows]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks3
GENESIS_BENCH_LONG_STEPS=0 rocprofv3 --kernel-trace --stats -d /tmp/ks3 -o ks --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $1 --steps 20 --warmup 3 --cpu-seconds 0 --profile-steps 0 --host-input-steps 0 --extra-leg-steps 0 --fp32-pipe-steps 0 > /dev/null 2>&1
python - "${2:-30}" <<'PY'
import csv,glob,sys
rows_n=int(sys.argv[1])
f=glob.glob('/tmp/ks3/**/ks_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=26.0   # 20 timed + 3 warm-up + 3 eager warm-up iterations of the capture
print('total kernel time per step ~ %.1f us, %d launches per step' % (tot/steps/1e3, sum(int(r['Calls']) for r in rows)/steps))
for i,r in enumerate(rows[:rows_n]):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    print('%-64s %7.2f x %8.1f us  %5.2f%%'%(n[:64], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
