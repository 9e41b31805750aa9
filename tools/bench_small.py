"""Timing of the HBM-bound non-MFMA kernels at the metric workload's shapes (B=32, K=7, 64x64)."""
import sys, os.path as osp
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
    return a.elapsed_time(b) / n * 1e3


N, C, S = 224, 64, 64
x = torch.randn(N, C, S, S, device=DEV)
w = torch.randn(4, C, device=DEV) * 0.1
b = torch.zeros(4, device=DEV)
dy = torch.randn(N, 4, S, S, device=DEV)
mb = x.numel() * 4 / 1e6
t = timeit(lambda: hip.conv1x1_fwd(x, w, b)); print('conv1x1 fwd  64->4 N224 @64: %.1f us  %.2f TB/s' % (t, mb / t))
t = timeit(lambda: hip.conv1x1_bwd(x, dy, w, b)); print('conv1x1 bwd (dgrad+wgrad)  : %.1f us  %.2f TB/s (x read + dx write)' % (t, 2 * mb / t))
x2 = torch.randn(32, 64, 64, 64, device=DEV)
w8 = torch.randn(8, 64, device=DEV) * 0.1
b8 = torch.zeros(8, device=DEV)
t = timeit(lambda: hip.conv1x1_fwd(x2, w8, b8)); print('conv1x1 fwd  64->8 N32  @64: %.1f us  %.2f TB/s' % (t, x2.numel() * 4 / 1e6 / t))
f = torch.randn(32, 64, 64, 64, device=DEV)
log_m = torch.log_softmax(torch.randn(7, 32, 1, 64, 64, device=DEV), 0)
t = timeit(lambda: hip.maskpool_fwd(f, log_m)); print('maskpool fwd: %.1f us' % t)
gS = torch.randn(32, 7, 64, device=DEV); gm = torch.randn(32, 7, device=DEV)
t = timeit(lambda: hip.maskpool_bwd(f, log_m, gS, gm)); print('maskpool bwd: %.1f us' % t)
xx = torch.rand(32, 3, 64, 64, device=DEV); dec = torch.randn(224, 4, 64, 64, device=DEV)
t = timeit(lambda: hip.mixture_fwd(xx, dec, 7, 0.7, True)); print('mixture fwd: %.1f us' % t)
ge = torch.full((32,), 1.0 / 32, device=DEV)
t = timeit(lambda: hip.mixture_bwd(xx, dec, ge, 7, 0.7, True)); print('mixture bwd: %.1f us' % t)
