python -m pytest tests/test_kernels_gpu.py -x -q -k "icsbp or wgrad or conv3x3" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py tests/test_error_budget_gpu.py -x -q 2>&1 | tail -2
python bench.py --steps 60 --warmup 10 --cpu-seconds 0 --profile-steps 0 --host-input-steps 0 2>&1 | tail -1 | cut -c1-200
