mkdir -p gpurun_out/s5
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "row_ring or streamk or conv5x5_stride1 or deferred_weight" -s > gpurun_out/s5/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/s5/tests.log
Q="--cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0"
for m in genesis vae monet; do
for st in 0 1 0 1; do
  GENESIS_WGQ_KSPLIT=$st timeout 600 python bench.py --model $m --steps 30 --warmup 5 $Q 2> /dev/null > gpurun_out/s5/bench_${m}_$st.json
  python -c "import json;d=json.load(open('gpurun_out/s5/bench_${m}_$st.json'));print('$m ksplit $st', round(d['value'],1), round(d['ms_per_step'],3), d['final_elbo'])" >> gpurun_out/s5/time.log
done
GENESIS_WGQ_TIMES=1 GENESIS_WGQ_DEBUG=1 timeout 300 python bench.py --model $m --steps 2 --warmup 1 --no-graph --profile-steps 0 $Q > /dev/null 2> gpurun_out/s5/times_$m.log
done
tail -4 gpurun_out/s5/tests.log; cat gpurun_out/s5/time.log
