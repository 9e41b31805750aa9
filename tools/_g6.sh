mkdir -p gpurun_out/s6
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -k "row_ring or conv5x5_stride1" -s 2>&1 | grep -v "^$" | grep "rel\|FAILED\|passed\|failed\|error" > gpurun_out/s6/tests.log
cat gpurun_out/s6/tests.log | tail -60
