#!/bin/bash
# PMC passes over tools/kq_probe.py: matrix-pipe busy fraction, effective clock, stall split, instruction mix.
# usage: bash tools/pmc_kq.sh "<probe args>" <kernel-name-substring>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="$1"; PAT="$2"
CMD="python $R/tools/kq_probe.py $ARGS"
i=0
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pk$i
  rocprofv3 --pmc $PMC --kernel-trace -d /tmp/pk$i -o p --output-format csv -- $CMD > /tmp/pk$i.log 2>&1
done
python - "$PAT" <<'PY'
import csv, sys, glob, collections
pat = sys.argv[1]
tot = {}
for d in sorted(glob.glob('/tmp/pk[0-9]')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter(); dur = {}
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
                if 'Start_Timestamp' in r:
                    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
        for k in acc:
            tot[k] = acc[k] / n[k]
            print('%-28s per dispatch %16.0f  (%d dispatches)' % (k, acc[k] / n[k], n[k]))
        if dur:
            tot['us_pass_' + d[-1]] = sum(dur.values()) / len(dur)
            print('  avg duration under this pass: %.1f us' % (sum(dur.values()) / len(dur)))
if 'GRBM_GUI_ACTIVE' in tot:
    us = tot.get('us_pass_3')
    print('effective clock %.3f GHz' % (tot['GRBM_GUI_ACTIVE'] / us / 1e3))
if 'SQ_VALU_MFMA_BUSY_CYCLES' in tot:
    us = tot.get('us_pass_2')
    print('matrix-pipe busy: %.3f of 1024 SIMD-cycles at 2.4 GHz' % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (us * 2400.0 * 1024)))
PY
