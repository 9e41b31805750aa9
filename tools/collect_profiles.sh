#!/bin/bash
# Collects the round's evidence on a GPU box into gpurun_out/profiles_new/: the default bench line, the rocprofv3
# kernel-trace stats of the same command, and the two HBM-traffic PMC passes (separate runs, kernel-trace only).
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_default.log 2>&1
grep "^{" $OUT/bench_default.log > $OUT/bench_1gpu.json
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 > $OUT/bench_under_rocprof.log 2>&1
cp /tmp/ks/*/ks_kernel_stats.csv $OUT/ 2>/dev/null || cp /tmp/ks/ks_kernel_stats.csv $OUT/
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-graph --profile-steps 0 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-graph --profile-steps 0 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 > /tmp/pw.log 2>&1
F=$(find /tmp/pf -name "f_counter_collection.csv" | head -1); W=$(find /tmp/pw -name "w_counter_collection.csv" | head -1)
python $R/tools/pmc_traffic.py $F $W $OUT/pmc_hbm_traffic.json
ls -la $OUT
