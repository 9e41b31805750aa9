#!/bin/bash
# MFMA-pipe utilisation and HBM traffic per kernel symbol over two eager bench steps (PMC passes, kernel-trace only).
# Writes gpurun_out/profiles_new/pmc_mfma_util.json:  SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration * 2.4 GHz * 1024 SIMDs).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GENESIS_BENCH_LONG_STEPS=0
CMD="python $R/bench.py $PMC_BENCH_ARGS --steps 2 --warmup 1 --no-graph --profile-steps 0 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0"      # PMC_BENCH_ARGS: another configuration (e.g. "--model monet")
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d /tmp/pm1 -o m --output-format csv -- $CMD > /tmp/pm1.log 2>&1
python - <<'PY'
import csv, glob, json, collections, os
f = glob.glob('/tmp/pm1/**/m_counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
cols = rows[0].keys()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    did = r.get('Dispatch_Id')
    if did not in seen:
        seen.add(did); n[k] += 1
        if 'Start_Timestamp' in cols and 'End_Timestamp' in cols:
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
out = {}
for k, c in acc.items():
    if k.startswith('at::') or 'rocclr' in k or not c.get('SQ_INSTS_MFMA'):
        continue
    d = dur.get(k, 0.0)
    out[k] = {'dispatches': n[k], 'avg_us_under_pmc': 1e6 * d / max(n[k], 1) if d else None,
              'mfma_busy_cycles_per_dispatch': c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(n[k], 1),
              # fraction of the 1024 SIMD matrix pipes' cycles at the 2.4 GHz peak clock (the chip sustains ~1.95 GHz
              # under this load, so 0.81 here = pipes always busy)
              'mfma_busy_frac_at_2p4GHz': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (d * 2.4e9 * 1024.0) if d else None,
              # the same against the cycles the chip actually ran: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 = busy
              # shader-clock cycles of the dispatch; their ratio to the duration is the average clock under this kernel
              'gui_active_cycles_per_dispatch_per_xcd': c['GRBM_GUI_ACTIVE'] / 8.0 / max(n[k], 1),
              'avg_clock_GHz': (c['GRBM_GUI_ACTIVE'] / 8.0) / (d * 1e9) if d else None,
              'mfma_busy_frac_of_active_cycles': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
              if c.get('GRBM_GUI_ACTIVE') else None,
              'insts_per_mfma': {'valu_non_mfma': (c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA'],
                                 'salu': c['SQ_INSTS_SALU'] / c['SQ_INSTS_MFMA'], 'lds': c['SQ_INSTS_LDS'] / c['SQ_INSTS_MFMA']}}
out = dict(sorted(out.items(), key=lambda kv: -(kv[1]['avg_us_under_pmc'] or 0) * kv[1]['dispatches']))
json.dump(out, open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/profiles_new/pmc_mfma_util.json', 'w'), indent=1)
print(list(cols))
for k, v in list(out.items())[:14]:
    print('%-34s x%-3d %8.1f us  mfma busy(2.4GHz) %s  busy/active %s at %s GHz  valu/mfma %.2f salu/mfma %.2f lds/mfma %.2f' % (k[:34], v['dispatches'], v['avg_us_under_pmc'] or 0, v['mfma_busy_frac_at_2p4GHz'] and round(v['mfma_busy_frac_at_2p4GHz'], 3), v['mfma_busy_frac_of_active_cycles'] and round(v['mfma_busy_frac_of_active_cycles'], 3), v['avg_clock_GHz'] and round(v['avg_clock_GHz'], 2), v['insts_per_mfma']['valu_non_mfma'], v['insts_per_mfma']['salu'], v['insts_per_mfma']['lds']))
PY
