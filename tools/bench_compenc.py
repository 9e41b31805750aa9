"""Per-layer timing of the ComponentVAE encoder's four stride-2 3x3 convs (modules/encoders.py:31-37; MONet / GENESIS,
N = K * B = 224) on the generic implicit-GEMM kernels: forward (+bias, ReLU), data gradient, weight gradient."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip

DEV = 'cuda'
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


N = 224
print('conv k3 s2 p1 (N=%d)       fwd us   GB/s | dgrad us | wgrad us' % N)
for ci, co, s in [(4, 32, 64), (32, 32, 32), (32, 64, 16), (64, 64, 8)]:
    x = torch.randn(N, ci, s, s, device=DEV)
    w = torch.randn(co, ci, 3, 3, device=DEV) * 0.05
    b = torch.randn(co, device=DEV)
    so = s // 2
    dy = torch.randn(N, co, so, so, device=DEV)
    mb = (x.numel() + dy.numel()) * 4 / 1e6
    tf = timeit(lambda: hip.conv2d_direct_fwd(x, w, b, 'relu', 2, 1))
    td = timeit(lambda: hip.conv2d_direct_dgrad(dy, w, s, s, 2, 1))
    tw = timeit(lambda: hip.conv2d_direct_wgrad(x, dy, 3, 2, 1))
    print('%3d->%3d @%2d  %6.1f MB  %7.1f %6.0f | %7.1f | %7.1f' % (ci, co, s, mb, tf, mb / tf * 1e3, td, tw))
