#!/bin/bash
# output: gpurun_out/f16x3_probe.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/f16x3_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/f16x3_probe.txt
