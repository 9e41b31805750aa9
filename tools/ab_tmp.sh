python -m pytest tests/test_kernels_gpu.py -x -q -k "conv3x3 or deconv" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py tests/test_error_budget_gpu.py tests/test_monet_gpu.py tests/test_genesis_gpu.py -x -q 2>&1 | tail -2
python bench.py --steps 60 --warmup 10 --cpu-seconds 0 --profile-steps 0 --host-input-steps 0 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --profile-steps 0 --host-input-steps 0 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/**/ks_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    if n.startswith('tapconv'): print(n[:50], r['Calls'], float(r['AverageNs'])/1e3)
PY
