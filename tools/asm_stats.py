"""Per-function register / scratch / instruction-mix statistics of a hipcc -save-temps .s file.
    python tools/asm_stats.py genesis_amd/csrc/gx_wgq-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re
import subprocess
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cur, stats, counts = None, {}, {}
for line in open(path):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1)
        counts[cur] = {}
        continue
    if cur is None:
        continue
    m = re.match(r'^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|codeLenInByte|Occupancy): (\d+)', line)
    if m:
        stats.setdefault(cur, {})[m.group(1)] = int(m.group(2))
        continue
    m = re.match(r'^\s+([a-z_0-9]+)\s', line)
    if m:
        op = m.group(1)
        key = ('mfma' if 'mfma' in op else 'scratch' if op.startswith('scratch_') else 'ds_read' if op.startswith('ds_read') or op.startswith('ds_load')
               else 'ds_write' if op.startswith('ds_write') or op.startswith('ds_store') else 'global' if op.startswith('global_') or op.startswith('buffer_')
               else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') and not op.startswith('s_waitcnt') and not op.startswith('s_nop')
               else 'waitcnt' if op.startswith('s_waitcnt') else 'other')
        counts[cur][key] = counts[cur].get(key, 0) + 1
for f, st in stats.items():
    if flt and flt not in f:
        continue
    try:
        name = subprocess.check_output(['c++filt', f]).decode().strip()
    except Exception:
        name = f
    print(name[:150])
    print('   ', st, counts.get(f))
