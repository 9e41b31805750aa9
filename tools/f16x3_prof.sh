#!/bin/bash
# kernel times of tools/f16x3_probe.py under rocprofv3 (the fp16 x 3 form's amax launches show up as their own rows)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/f16x3_prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/f16x3_prof -o f16 --output-format csv -- python tools/f16x3_probe.py > gpurun_out/f16x3_prof/log.txt 2>&1
f=$(find gpurun_out/f16x3_prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:90]
    print('%6s calls  avg %8.1f us  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, n))
PY
