"""Per-layer timing of the generic implicit-GEMM conv (gx_conv2d_direct_*) at the sylvester VAE's shapes (N = 224)."""
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
print('conv k5 (N=%d)            fwd us   TF | dgrad us   TF | wgrad us   TF' % N)
for ci, co, s, st in [(4, 64, 64, 1), (32, 64, 64, 2), (32, 128, 32, 1), (64, 128, 32, 2), (64, 128, 16, 1)]:
    x = torch.randn(N, ci, s, s, device=DEV)
    w = torch.randn(co, ci, 5, 5, device=DEV) * 0.05
    so = (s + 4 - 5) // st + 1
    dy = torch.randn(N, co, so, so, device=DEV)
    fl = 2.0 * N * ci * co * 25 * so * so
    tf = timeit(lambda: hip.conv2d_direct_fwd(x, w, None, None, st, 2))
    td = timeit(lambda: hip.conv2d_direct_dgrad(dy, w, s, s, st, 2))
    tw = timeit(lambda: hip.conv2d_direct_wgrad(x, dy, 5, st, 2))
    print('%3d->%3d @%2d s%d %7.2f GF  %7.1f %5.1f | %7.1f %5.1f | %7.1f %5.1f' %
          (ci, co, s, st, fl / 1e9, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6))
