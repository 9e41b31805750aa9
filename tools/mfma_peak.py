"""fp32-MFMA ceiling of the chip and what operand delivery costs (gx_mfma_fp32_probe): a kernel of nothing but
v_mfma_f32_32x32x2_f32, 2 workgroups (8 waves) per CU; mode 0 register operands, 1 B from LDS, 2 A and B from LDS,
3 as 2 plus a barrier every 32 MFMAs, 4 A and B from LDS with one 16-byte read per four MFMAs, 5 the gx_kq.hip inner
loop (four 16-byte reads per 16 MFMAs, issued one step ahead, 2 x 2 tiles), 6 the same with 2 x 4 tiles; 5 / 6 also
report the shader clock the kernel ran at."""
import ctypes, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import _lib

scratch = torch.zeros(16, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode in (0, 1, 2, 3, 4, 5, 6):
    for wgs, iters in ((256, 4000), (512, 4000), (2048, 4000)):
        fl = ctypes.c_double(0.0)
        args = (wgs, iters, mode, ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(fl), st)
        _lib.call('gx_mfma_fp32_probe', *args)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        for _ in range(3):
            _lib.call('gx_mfma_fp32_probe', *args)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        clk = ('  shader clock %.2f GHz' % (scratch[1].item() * 0.1)) if mode >= 5 else ''
        print('mode %d  %5d workgroups: %8.3f ms  %6.1f TFLOP/s (nominal 157.3)%s' % (mode, wgs, ms, fl.value / ms / 1e9, clk))
