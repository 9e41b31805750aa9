"""Times the Winograd conv3x3 launches (HIP events) on both matrix pipes: argv N:Cin:Cout:S ..."""
import os, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip, _lib, profiling
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for spec in sys.argv[1:]:
    N, Ci, Co, S = [int(v) for v in spec.split(':')]
    x = torch.randn(N, Ci, S, S, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1) if N * S <= 2048 else None
    for mode in ((1,) if os.environ.get('WINO_ONLY_H') else (0, 1)):
        _lib.call('gx_wino_precision', mode)
        _lib.call('gx_conv3x3_wino_policy', 2)
        timeit(lambda: hip.conv3x3_fwd(x, w), 3)
        profiling.enable(True)
        for _ in range(20): hip.conv3x3_fwd(x, w)
        torch.cuda.synchronize()
        rows = {r['name']: r for r in profiling.collect()}
        profiling.enable(False)
        t = 1e3 * rows['wino_conv_kernel']['ms'] / rows['wino_conv_kernel']['launches']     # the kernel alone (HIP events)
        y = hip.conv3x3_fwd(x, w)
        err = float((y.double() - ref).norm() / ref.norm()) if ref is not None else float('nan')
        fl = 2.0 * N * Ci * Co * 9 * S * S
        print('%-18s pipe %s  %8.1f us  %6.1f TF/s algorithmic   rel err vs fp64 %.2e' % (spec, 'bf16x6' if mode else 'fp32  ', t, fl / t / 1e6, err), flush=True)
_lib.call('gx_wino_precision', 1); _lib.call('gx_conv3x3_wino_policy', 1)
