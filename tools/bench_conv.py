"""Micro-benchmark of single conv entry points at the metric workload's shapes (for PMC / ISA work)."""
import sys, os.path as osp, time
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip

DEV = 'cuda'
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


which = sys.argv[1] if len(sys.argv) > 1 else 'all'
N, C, S = 32, 64, 64
x = torch.randn(N, C, S, S, device=DEV)
w = torch.randn(C, C, 3, 3, device=DEV) * 0.05
dy = torch.randn(N, C, S, S, device=DEV)
fl = 2.0 * N * C * C * 9 * S * S
if which in ('all', 'c3'):
    t = timeit(lambda: hip.conv3x3_fwd(x, w)); print('conv3x3 fwd 64->64@64 B32: %.1f us  %.1f TF' % (t, fl / t / 1e6))
    t = timeit(lambda: hip.conv3x3_wgrad(x, dy)); print('conv3x3 wgrad           : %.1f us  %.1f TF' % (t, fl / t / 1e6))
if which in ('all', 'dc'):
    Nd = 224
    xd = torch.randn(Nd, 64, 32, 32, device=DEV)
    wd = torch.randn(64, 64, 5, 5, device=DEV) * 0.05
    bd = torch.zeros(64, device=DEV)
    dyd = torch.randn(Nd, 64, 64, 64, device=DEV)
    fd = 2.0 * Nd * 64 * 64 * 25 * 32 * 32
    t = timeit(lambda: hip.deconv5x5s2_fwd(xd, wd, bd), 10); print('deconv fwd 64->64 32->64 N224: %.1f us  %.1f TF' % (t, fd / t / 1e6))
    t = timeit(lambda: hip.deconv5x5s2_dgrad(dyd, wd), 10); print('deconv dgrad               : %.1f us  %.1f TF' % (t, fd / t / 1e6))
    t = timeit(lambda: hip.deconv5x5s2_wgrad(xd, dyd), 10); print('deconv wgrad               : %.1f us  %.1f TF' % (t, fd / t / 1e6))
