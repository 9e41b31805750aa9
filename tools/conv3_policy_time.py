"""conv3x3 forward / data gradient per layer shape under both Winograd dispatch policies (1: chip-filling layers only, 2: every
eligible shape): whole-call time by HIP events (pack + kernel(s) + reduce).  argv: N:Cin:Cout:S ..."""
import os, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip, _lib
torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for spec in sys.argv[1:]:
    N, Ci, Co, S = [int(v) for v in spec.split(':')]
    x = torch.randn(N, Ci, S, S, device='cuda'); dy = torch.randn(N, Co, S, S, device='cuda')
    w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05
    out = []
    for pol in (1, 2):
        _lib.call('gx_conv3x3_wino_policy', pol)
        out.append((timeit(lambda: hip.conv3x3_fwd(x, w)), timeit(lambda: hip.conv3x3_dgrad(dy, w))))
    _lib.call('gx_conv3x3_wino_policy', 1)
    print('%-16s fwd %7.1f -> %7.1f us   dgrad %7.1f -> %7.1f us   (policy 1 -> 2)' % (spec, out[0][0], out[1][0], out[0][1], out[1][1]), flush=True)
