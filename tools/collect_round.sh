#!/bin/bash
# tools/collect_round.sh <tag>: every measurement the round's profiles/ files come from, on one GPU box, into
# gpurun_out/round_<tag>/ (copy what is to be judged into profiles/<tag>_*).
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round_$TAG
mkdir -p $OUT
cd $R
Q="--cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0"
# 1. the default bench line (with the CPU baseline and the fp32-pipe-only leg), kernel stats, HBM traffic (two PMC passes)
bash tools/collect_profiles.sh > $OUT/collect_profiles.log 2>&1
cp gpurun_out/profiles_new/bench_1gpu.json $OUT/bench_1gpu.json
cp gpurun_out/profiles_new/ks_kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv
cp gpurun_out/profiles_new/pmc_hbm_traffic.json $OUT/pmc_hbm_traffic.json
# 2. MFMA-pipe utilisation per kernel
bash tools/pmc_mfma_util.sh > $OUT/pmc_mfma_util.log 2>&1
cp gpurun_out/profiles_new/pmc_mfma_util.json $OUT/pmc_mfma_util.json
# 2b. stall attribution (three PMC passes): parked / issue-stall / issuing shares of the wave cycles, LDS conflicts, occupancy
bash tools/pmc_stalls.sh > $OUT/pmc_stalls.log 2>&1
cp gpurun_out/profiles_new/pmc_stalls.json $OUT/pmc_stalls.json
# 3. one step's kernel sequence
bash tools/trace_step.sh > /dev/null 2>&1
cp gpurun_out/step_seq.txt $OUT/step_kernel_sequence.txt
# 4. the other GENESIS-V2 configurations
python bench.py --K 5 --batch 64 --steps 30 --warmup 5 $Q 2> /dev/null > $OUT/bench_cfg2_K5_B64.json
python bench.py --K 11 --img 128 --steps 20 --warmup 5 $Q 2> /dev/null > $OUT/bench_cfg5_K11_128.json
# 4b. kernel stats, HBM traffic and matrix-pipe utilisation of the 128 x 128 configuration (and kernel stats of config 2)
cd /tmp && export TMPDIR=/tmp
C5="--K 11 --img 128 --steps 10 --warmup 3 --profile-steps 0 $Q"
rocprofv3 --kernel-trace --stats -d /tmp/ks5 -o ks --output-format csv -- python $R/bench.py $C5 > /tmp/ks5.log 2>&1
cp $(find /tmp/ks5 -name "ks_kernel_stats.csv" | head -1) $OUT/cfg5_rocprofv3_kernel_stats.csv
P5="--K 11 --img 128 --steps 2 --warmup 1 --no-graph --profile-steps 0 $Q"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf5 -o f --output-format csv -- python $R/bench.py $P5 > /tmp/pf5.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw5 -o w --output-format csv -- python $R/bench.py $P5 > /tmp/pw5.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pf5 -name "f_counter_collection.csv" | head -1) $(find /tmp/pw5 -name "w_counter_collection.csv" | head -1) $OUT/cfg5_pmc_hbm_traffic.json
STALL_BENCH_ARGS="--K 11 --img 128" bash $R/tools/pmc_stalls.sh > $OUT/cfg5_pmc_stalls.log 2>&1
cp $R/gpurun_out/profiles_new/pmc_stalls.json $OUT/cfg5_pmc_stalls.json
rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o ks --output-format csv -- python $R/bench.py --K 5 --batch 64 --steps 10 --warmup 3 --profile-steps 0 $Q > /tmp/ks2.log 2>&1
cp $(find /tmp/ks2 -name "ks_kernel_stats.csv" | head -1) $OUT/cfg2_rocprofv3_kernel_stats.csv
cd $R
# 5. the other model families: bench line + kernel stats + step sequence
for m in genesis monet vae; do
    python bench.py --model $m --steps 30 --warmup 5 $Q 2> /dev/null > $OUT/bench_$m.json
    bash tools/trace_model.sh $m > /dev/null 2>&1
    cp gpurun_out/trace_${m}_kernel_stats.csv $OUT/${m}_rocprofv3_kernel_stats.csv
    cp gpurun_out/trace_${m}_seq.txt $OUT/${m}_step_kernel_sequence.txt
done
# 6. the data-parallel step on one GPU: bucket all-reduce forced (world 1), captured in the step's graph / split / off
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
GENESIS_FORCE_ALLREDUCE=1 python bench.py --steps 50 --warmup 10 $Q 2> /dev/null > $OUT/bench_1gpu_forced_allreduce.json
GENESIS_FORCE_ALLREDUCE=1 GENESIS_GRAPH_ALLREDUCE=0 python bench.py --steps 50 --warmup 10 $Q 2> /dev/null > $OUT/bench_1gpu_forced_allreduce_split_graphs.json
# 6b. the same collective through the library's own entry points (gx_allreduce_*), and the early decoder flush + early collective
GENESIS_FORCE_ALLREDUCE=1 GENESIS_CABI_ALLREDUCE=1 python bench.py --steps 50 --warmup 10 $Q 2> /dev/null > $OUT/bench_1gpu_forced_allreduce_cabi.json
GENESIS_FORCE_ALLREDUCE=1 GENESIS_WGQ_EARLY_FLUSH=1 python bench.py --steps 50 --warmup 10 $Q 2> /dev/null > $OUT/bench_1gpu_forced_allreduce_early_flush.json
ls -la $OUT
