"""Where a translation unit's scratch traffic sits relative to its MFMA loops: per function of a hipcc .s file the number of scratch_*
instructions, how many of them lie between the function's first and last MFMA (i.e. can execute inside the tile loop), and the MFMA
count.  The stream-K weight-gradient kernel calls its tile bodies as out-of-line functions (DESIGN.md finding 7): their callee-save
spills belong to the call, once per SEGMENT, and this listing is the evidence that none sits in a tile loop.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only genesis_amd/csrc/gx_wgq.hip -o /tmp/wgq.s
    python tools/asm_scratch_vs_mfma.py /tmp/wgq.s"""
import re
import subprocess
import sys

cur, data = None, {}
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1); data[cur] = []
        continue
    if cur is None:
        continue
    if re.match(r'^; codeLenInByte', line):
        cur = None
        continue
    t = line.strip()
    if t.startswith('scratch_'):
        data[cur].append('S')
    elif 'mfma' in t:
        data[cur].append('M')
tot = inside_tot = 0
for f, seq in data.items():
    s = ''.join(seq)
    if 'S' not in s:
        continue
    first, last = s.find('M'), s.rfind('M')
    inside = s[first:last].count('S') if first >= 0 else 0
    tot += s.count('S'); inside_tot += inside
    name = subprocess.check_output(['c++filt', f]).decode().strip()
    name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
    print('%-44s scratch instructions %4d   between first and last MFMA %3d   MFMAs %4d' % (name[:44], s.count('S'), inside, s.count('M')))
print('total: %d scratch instructions, %d of them between MFMAs' % (tot, inside_tot))
