"""Shader clock reported by the kq-loop probe (mode 5) for short and long launches, back to back."""
import ctypes, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import _lib
scratch = torch.zeros(16, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode in (5, 6, 0):
    for wgs, iters, reps in ((512, 100, 40), (512, 400, 20), (512, 4000, 3), (256, 100, 40), (1024, 100, 40)):
        fl = ctypes.c_double(0.0)
        args = (wgs, iters, mode, ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(fl), st)
        for _ in range(3): _lib.call('gx_mfma_fp32_probe', *args)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        for _ in range(reps): _lib.call('gx_mfma_fp32_probe', *args)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        print('mode %d %5d wgs x %4d iters: %8.1f us  %6.1f TF  clock %.2f GHz' % (mode, wgs, iters, ms * 1e3, fl.value / ms / 1e9, scratch[1].item() * 0.1), flush=True)
