"""Instruction sequence of one basic block as a string: M mfma, v valu, r ds_read, w ds_write, g global load/store, W s_waitcnt,
s salu, B barrier, n nop.   python tools/asm_seq.py file.s <function substring> <block label>"""
import re
import sys
path, flt, label = sys.argv[1:4]
on = blk = False
out = []
for line in open(path):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        on = flt in m.group(1); blk = False
        continue
    if not on:
        continue
    m = re.match(r'^(\.LBB\w+):', line)
    if m:
        blk = m.group(1) == label
        continue
    if re.match(r'^; codeLenInByte', line):
        on = False
    if not blk:
        continue
    m = re.match(r'^\s+([a-z_0-9]+)\s*(.*)', line)
    if m:
        op = m.group(1)
        c = ('M' if 'mfma' in op else 'r' if op.startswith('ds_read') or op.startswith('ds_load') else 'w' if op.startswith('ds_write') or op.startswith('ds_store')
             else 'g' if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') else 'S' if op.startswith('scratch_') else 'a' if op.startswith('v_accvgpr')
             else 'v' if op.startswith('v_') else 'W' if op.startswith('s_waitcnt') else 'n' if op.startswith('s_nop') else 'B' if op.startswith('s_barrier') else 's')
        out.append(c)
s = ''.join(out)
for i in range(0, len(s), 150):
    print(s[i:i + 150])
