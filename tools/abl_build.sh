#!/bin/bash
# Builds a variant of libgenesis_hip.so with one translation unit recompiled under extra flags:
#   tools/abl_build.sh <name> <source.hip> "<extra hipcc flags>"   ->  tools/abl/lib_<name>.so
# (run a probe against it with GENESIS_HIP_LIB=tools/abl/lib_<name>.so)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
name=$1; src=$2; flags=$3
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed $flags -c genesis_amd/csrc/$src -o tools/abl/${base}_$name.o
objs=$(ls genesis_amd/csrc/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/lib_$name.so $objs tools/abl/${base}_$name.o
echo tools/abl/lib_$name.so
