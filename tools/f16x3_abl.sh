#!/bin/bash
# what bounds the fp16 x 3 transposed-conv kernels: launch times of the default build and of the -DGX_QH_ABL builds (tools/abl_build.sh:
# 1 = no output stores, 2 = no input split / LDS store after a tile's first chunk, 3 = both); measurement builds, wrong results
cd "$(dirname "$0")/.."
for L in genesis_amd/libgenesis_hip.so tools/abl/lib_a1.so tools/abl/lib_a2.so tools/abl/lib_a3.so; do
  echo "== $L"
  GENESIS_HIP_LIB=$PWD/$L python tools/kq_time.py fwd:224:16 fwd:224:32 dgrad:224:16 dgrad:224:32 2>&1 | grep -v amdgpu.ids
done
