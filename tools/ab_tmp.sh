cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_genesis_gpu.py tests/test_monet_gpu.py tests/test_dp_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25
) > gpurun_out/t3.log 2>&1; cat gpurun_out/t3.log
