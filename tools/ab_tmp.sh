cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_genesis_gpu.py tests/test_monet_gpu.py tests/test_fullbatch_gpu.py tests/test_error_budget_gpu.py tests/test_sample_models.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -6
  for m in genesis monet; do
  python tools/glue_trace.py --model $m 2>&1 | grep "aten::" | cut -c1-200
  python bench.py --model $m --steps 30 --warmup 10 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 --profile-steps 0 2>/dev/null | cut -c1-300
  done
) > gpurun_out/t2.log 2>&1; cat gpurun_out/t2.log
