cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_genesis_gpu.py -x -q -m gpu 2>&1 | tail -15
  timeout 900 python -m pytest tests/test_fullbatch_gpu.py tests/test_error_budget_gpu.py tests/test_sample_models.py -x -q -m gpu -k "genesis" 2>&1 | tail -8
  python tools/glue_trace.py --model genesis 2>&1 | grep "aten::" | cut -c1-230
  python bench.py --model genesis --steps 30 --warmup 10 --cpu-seconds 0 --fp32-pipe-steps 0 --host-input-steps 0 --extra-leg-steps 0 --profile-steps 0 2>/dev/null | cut -c1-400
) > gpurun_out/t1.log 2>&1; cat gpurun_out/t1.log
