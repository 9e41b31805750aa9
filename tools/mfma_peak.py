"""Practical fp32-MFMA ceiling of the chip: a kernel of nothing but v_mfma_f32_32x32x2_f32 on register operands
(gx_mfma_fp32_probe), 1 and 2 workgroups per CU, short and long runs (the clock settles under sustained load)."""
import ctypes, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import _lib

scratch = torch.zeros(16, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for wgs, iters in ((256, 2000), (512, 2000), (512, 20000), (2048, 5000)):
    fl = ctypes.c_double(0.0)
    _lib.call('gx_mfma_fp32_probe', wgs, iters, ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(fl), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(5):
        _lib.call('gx_mfma_fp32_probe', wgs, iters, ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(fl), st)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print('%5d workgroups x %6d iters: %8.3f ms  %6.1f TFLOP/s (nominal 157.3)' % (wgs, iters, ms, fl.value / ms / 1e9))
