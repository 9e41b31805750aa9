python tools/check_wino.py 2>&1 | tail -16 | cut -c1-175
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv3x3 or wino" 2>&1 | tail -2
python bench.py --steps 60 --warmup 10 --cpu-seconds 0 --profile-steps 0 --host-input-steps 0 2>&1 | tail -1 | cut -c1-200
