for e in "X=1" "GENESIS_MASKPOOL_KT16=1"; do
echo "== $e"; env $e python -m pytest tests/test_error_budget_gpu.py -x -q -s -k "k5" 2>&1 | grep -E "decoder_module.1.weight|encoder.down.0.0.weight|worst|passed|failed"
done
