"""Instruction mix per basic block of one function of a hipcc -save-temps .s file.
    python tools/asm_blocks.py file.s <mangled-name substring> [min instructions]"""
import re
import sys

path, flt = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
on = False
blocks, cur = [], None
for line in open(path):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        on = flt in m.group(1)
        if on:
            cur = [m.group(1)[:60], {}]
            blocks.append(cur)
        continue
    if not on:
        continue
    if re.match(r'^; codeLenInByte', line):
        on = False
        continue
    m = re.match(r'^(\.LBB\w+):', line)
    if m:
        cur = [m.group(1), {}]
        blocks.append(cur)
        continue
    m = re.match(r'^\s+([a-z_0-9]+)\s', line)
    if m and cur is not None:
        op = m.group(1)
        key = ('mfma' if 'mfma' in op else 'scratch' if op.startswith('scratch_') else 'ds_rd' if op.startswith('ds_read') or op.startswith('ds_load')
               else 'ds_wr' if op.startswith('ds_write') or op.startswith('ds_store') else 'glob' if op.startswith('global_') or op.startswith('buffer_')
               else 'accmov' if op.startswith('v_accvgpr') else 'valu' if op.startswith('v_') else 'wait' if op.startswith('s_waitcnt') else 'nop' if op.startswith('s_nop')
               else 'branch' if op.startswith('s_cbranch') or op.startswith('s_branch') else 'barrier' if op.startswith('s_barrier') else 'salu' if op.startswith('s_') else 'other')
        cur[1][key] = cur[1].get(key, 0) + 1
for name, c in blocks:
    n = sum(c.values())
    if n >= minn:
        print('%-40s %5d  %s' % (name, n, ' '.join('%s=%d' % kv for kv in sorted(c.items()))))
