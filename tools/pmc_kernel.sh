#!/bin/bash
# PMC passes (SQ instruction mix, stall and I-cache counters) over a micro-benchmark command; prints per-kernel sums.
# usage: bash tools/pmc_kernel.sh "<python command>" <kernel-name-substring>
cd /tmp && export TMPDIR=/tmp
CMD="$1"; PAT="$2"
i=0
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC -d /tmp/pmc$i -o p --output-format csv -- $CMD > /tmp/pmc$i.log 2>&1
done
python - "$PAT" <<'PY'
import csv, sys, glob, collections
pat = sys.argv[1]
for d in sorted(glob.glob('/tmp/pmc[0-9]')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
        for k in acc:
            print('%-28s %16.0f  (per dispatch %14.0f, %d dispatches)' % (k, acc[k], acc[k] / n[k], n[k]))
PY
