#!/bin/bash
# Stall attribution per kernel symbol over two eager bench steps (three PMC passes, kernel-trace only; 8 SQ slots per pass).
# Writes gpurun_out/profiles_new/pmc_stalls.json.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
# waves; WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GENESIS_BENCH_LONG_STEPS=0
CMD="python $R/bench.py --steps 2 --warmup 1 --no-graph --profile-steps 0 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 $STALL_BENCH_ARGS"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/ps1 -o s --output-format csv -- $CMD > /tmp/ps1.log 2>&1 || tail -5 /tmp/ps1.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d /tmp/ps2 -o s --output-format csv -- $CMD > /tmp/ps2.log 2>&1 || tail -5 /tmp/ps2.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace -d /tmp/ps3 -o s --output-format csv -- $CMD > /tmp/ps3.log 2>&1 || tail -5 /tmp/ps3.log
python - <<'PY'
import csv, glob, json, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(float); ndur = collections.Counter()
for pi, d in enumerate(('/tmp/ps1', '/tmp/ps2', '/tmp/ps3')):
    fs = glob.glob(d + '/**/s_counter_collection.csv', recursive=True)
    if not fs:
        print('no counters from', d); continue
    rows = list(csv.DictReader(open(fs[0])))
    seen = set()
    for r in rows:
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        did = r.get('Dispatch_Id')
        if did not in seen:
            seen.add(did); n[k][pi] += 1
            if pi == 0 and 'Start_Timestamp' in r:
                dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9; ndur[k] += 1
out = {}
for k, c in acc.items():
    if k.startswith('at::') or 'rocclr' in k:
        continue
    nd = max(n[k].values()) or 1
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    e = {'dispatches': nd, 'avg_us_under_pmc': 1e6 * dur[k] / ndur[k] if ndur[k] else None}
    if wc:
        e['share_of_wave_cycles'] = {'parked_waitcnt_barrier': c.get('SQ_WAIT_ANY', 0) / wc, 'issue_stall': c.get('SQ_WAIT_INST_ANY', 0) / wc,
                                     'issue_stall_lds': c.get('SQ_WAIT_INST_LDS', 0) / wc, 'issuing': c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                                     'issuing_valu_incl_mfma': c.get('SQ_ACTIVE_INST_VALU', 0) / wc, 'issuing_lds': c.get('SQ_ACTIVE_INST_LDS', 0) / wc,
                                     'issuing_vmem': c.get('SQ_ACTIVE_INST_VMEM', 0) / wc, 'issuing_scalar': c.get('SQ_ACTIVE_INST_SCA', 0) / wc}
        if c.get('SQ_BUSY_CYCLES'):
            e['avg_waves_in_flight_per_busy_sq_cycle'] = wc / c['SQ_BUSY_CYCLES']
    if c.get('SQ_LDS_IDX_ACTIVE'):
        e['lds'] = {'bank_conflict_cycles_over_active': c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'],
                    'addr_conflict_cycles_over_active': c.get('SQ_LDS_ADDR_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'],
                    'unaligned_stall_over_active': c.get('SQ_LDS_UNALIGNED_STALL', 0) / c['SQ_LDS_IDX_ACTIVE'],
                    'active_cycles_per_lds_inst': c['SQ_LDS_IDX_ACTIVE'] / max(c.get('SQ_INSTS_LDS', 0), 1.0)}
    if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        e['mfma_busy_frac_of_active_cycles'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
        e['lds_array_active_frac_of_active_cycles'] = c.get('SQ_LDS_IDX_ACTIVE', 0) / (c['GRBM_GUI_ACTIVE'] / 8.0 * 256.0)
    if c.get('SQ_WAVES'):
        e['waves_per_dispatch'] = c['SQ_WAVES'] / nd
    if c.get('SQ_INSTS_MFMA'):
        e['insts_per_mfma'] = {'valu_non_mfma': (c.get('SQ_INSTS_VALU', 0) - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA'] if c.get('SQ_INSTS_VALU') else None,
                               'salu': c.get('SQ_INSTS_SALU', 0) / c['SQ_INSTS_MFMA'], 'lds': c.get('SQ_INSTS_LDS', 0) / c['SQ_INSTS_MFMA'],
                               'vmem': c.get('SQ_INSTS_VMEM', 0) / c['SQ_INSTS_MFMA']}
    e['raw_per_dispatch'] = {kk: vv / nd for kk, vv in sorted(c.items())}
    out[k] = e
out = dict(sorted(out.items(), key=lambda kv: -(kv[1]['avg_us_under_pmc'] or 0) * kv[1]['dispatches']))
json.dump(out, open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/profiles_new/pmc_stalls.json', 'w'), indent=1)
for k, v in list(out.items())[:12]:
    s = v.get('share_of_wave_cycles', {})
    print('%-30s x%-3d %7.1f us  parked %.2f stall %.2f (lds %.2f) issuing %.2f | waves/sq %.1f | lds conflict %.2f busy %.2f | mfma %.2f' % (
        k[:30], v['dispatches'], v['avg_us_under_pmc'] or 0, s.get('parked_waitcnt_barrier', 0), s.get('issue_stall', 0), s.get('issue_stall_lds', 0),
        s.get('issuing', 0), v.get('avg_waves_in_flight_per_busy_sq_cycle', 0), v.get('lds', {}).get('bank_conflict_cycles_over_active', 0),
        v.get('lds_array_active_frac_of_active_cycles', 0), v.get('mfma_busy_frac_of_active_cycles', 0)))
PY
