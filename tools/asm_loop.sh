#!/bin/bash
# compiles one csrc file to /tmp/asm/<name>.s (device assembly) and prints the block mix + sequence of the block of function $2 that holds the most MFMAs
# usage: tools/asm_loop.sh gx_wgq.hip wr_segment_callILi0ELi64 [extra hipcc flags]
set -e
src=$1; fn=$2; shift 2
mkdir -p /tmp/asm
out=/tmp/asm/$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed "$@" -S --cuda-device-only -o $out.s $(dirname $0)/../genesis_amd/csrc/$src
python $(dirname $0)/asm_stats.py $out.s $fn
python $(dirname $0)/asm_blocks.py $out.s $fn 100
L=$(python $(dirname $0)/asm_blocks.py $out.s $fn 100 | grep mfma= | sort -t= -k2 | awk '{print $1}' | tail -1)
L=$(python $(dirname $0)/asm_blocks.py $out.s $fn 100 | python -c "
import sys,re
best=None
for l in sys.stdin:
    m=re.search(r'mfma=(\d+)',l)
    if m and (best is None or int(m.group(1))>best[0]): best=(int(m.group(1)),l.split()[0])
print(best[1])")
echo "block $L"
python $(dirname $0)/asm_seq.py $out.s $fn $L
