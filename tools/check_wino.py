"""Winograd conv3x3 vs the direct tap-conv kernels: accuracy (against an fp64 torch reference) and time."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
import torch.nn.functional as F
from genesis_amd import hip_ops as hip

def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

torch.manual_seed(0)
for (N, Cin, Cout, S) in [(2, 16, 16, 16), (3, 24, 40, 32), (32, 64, 64, 64), (32, 128, 64, 64), (32, 64, 128, 64), (32, 64, 64, 32), (32, 128, 64, 32), (32, 256, 64, 16)]:
    x = torch.randn(N, Cin, S, S, device='cuda')
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05
    dy = torch.randn(N, Cout, S, S, device='cuda')
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    refd = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
    yw, yd = hip.conv3x3_wino(x, w, 0), hip.conv3x3_fwd(x, w)
    dw_, dd = hip.conv3x3_wino(dy, w, 1), hip.conv3x3_dgrad(dy, w)
    e = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    fl = 2.0 * N * Cin * Cout * 9 * S * S
    tw, td = t_us(lambda: hip.conv3x3_wino(x, w, 0)), t_us(lambda: hip.conv3x3_fwd(x, w))
    tw2, td2 = t_us(lambda: hip.conv3x3_wino(dy, w, 1)), t_us(lambda: hip.conv3x3_dgrad(dy, w))
    print('%3d %3d->%3d @%2d  fwd err wino %.1e direct %.1e | dgrad err wino %.1e direct %.1e | fwd us wino %6.1f direct %6.1f | dgrad us wino %6.1f direct %6.1f'
          % (N, Cin, Cout, S, e(yw, ref), e(yd, ref), e(dw_, refd), e(dd, refd), tw, td, tw2, td2))

# kernel-only times (HIP events around the launches, genesis_amd.profiling)
from genesis_amd import profiling
print('kernel-only us (forward):')
for (N, Cin, Cout, S) in [(32, 64, 64, 64), (32, 128, 64, 64), (32, 64, 128, 64), (32, 64, 64, 32), (32, 128, 64, 32), (32, 128, 128, 32)]:
    x = torch.randn(N, Cin, S, S, device='cuda'); w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05
    for _ in range(3): hip.conv3x3_wino(x, w, 0)
    torch.cuda.synchronize(); profiling.enable(True)
    for _ in range(10): hip.conv3x3_wino(x, w, 0)
    torch.cuda.synchronize(); rows = {r['name']: r for r in profiling.collect()}; profiling.enable(False)
    r = rows['wino_conv_kernel']; us = 1e3 * r['ms'] / r['launches']
    fl = 2.0 * N * Cin * Cout * 9 * S * S
    print('  %3d %3d->%3d @%2d  %6.1f us  %5.1f TF algorithmic  (%4.1f TF on the MFMA pipe)' % (N, Cin, Cout, S, us, fl / us / 1e6, fl / 2.25 / us / 1e6))

