"""fp32 products from THREE fp16 piece products (gx_kq_precision(2)) against six bf16 ones (gx_kq_precision(1)) in the transposed-conv
kernels (DESIGN.md finding 40): relative L2 error of forward and data gradient against fp64 (next to the CPU's fp32 op) -- on uniform
data, on data whose channels span 12 binary orders of magnitude with exact zeros (what ONE power-of-two scale per tensor has to
survive), on tensors scaled as a whole by 2^-30 / 2^+30 (the scale follows the tensor), and launch times at the workload's shapes
(the fp16 form's time includes its two amax launches).     python tools/f16x3_probe.py"""
import os, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from genesis_amd import hip_ops as hip, _lib

DEV = 'cuda'
NAMES = {1: 'bf16 x 6', 2: 'fp16 x 3'}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def rel(a, ref):
    return float((a.detach().cpu().double() - ref).norm() / (ref.norm() + 1e-300))


def case(N, Cin, Cout, H, kind):
    x, w, b = rnd(N, Cin, H, H, seed=4), rnd(Cin, Cout, 5, 5, seed=5, scale=1 / np.sqrt(Cin * 6.25)), rnd(Cout, seed=6)
    dy = rnd(N, Cout, 2 * H, 2 * H, seed=7)
    if kind == 'wide':      # channels spread over 12 binary orders of magnitude, exact zeros (ReLU outputs)
        x = torch.relu(x * torch.pow(2.0, -(torch.arange(Cin).float() % 13)).view(1, -1, 1, 1))
        dy = dy * torch.pow(2.0, -(torch.arange(Cout).float() % 13)).view(1, -1, 1, 1)
        w = w * torch.pow(2.0, -(torch.arange(Cout).float() % 7)).view(1, -1, 1, 1)
    elif kind == 'tiny':
        x, dy, w, b = x * 2.0 ** -30, dy * 2.0 ** -30, w * 2.0 ** -20, b * 2.0 ** -50
    elif kind == 'huge':
        x, dy, w, b = x * 2.0 ** 30, dy * 2.0 ** 30, w * 2.0 ** 20, b * 2.0 ** 50
    outs = {}
    for dt in (torch.float64, torch.float32):
        xr = x.to(dt).requires_grad_()
        y = F.conv_transpose2d(xr, w.to(dt), b.to(dt), 2, 2, 1)
        y.backward(dy.to(dt))
        outs[dt] = (y.detach(), xr.grad)
    line = 'deconv %3d,%d,%d,%d %-8s cpu32 fwd %.2e dgrad %.2e' % (N, Cin, Cout, H, kind, rel(outs[torch.float32][0], outs[torch.float64][0]),
                                                                 rel(outs[torch.float32][1], outs[torch.float64][1]))
    for mode in (1, 2):
        _lib.call('gx_kq_precision', mode)
        f = hip.deconv5x5s2_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
        d = hip.deconv5x5s2_dgrad(dy.to(DEV), w.to(DEV))
        line += ' | %s fwd %.2e dgrad %.2e' % (NAMES[mode], rel(f, outs[torch.float64][0]), rel(d, outs[torch.float64][1]))
    print(line, flush=True)


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


_lib.call('gx_kq_policy', 2)      # every eligible shape on the kq kernels (the small error cases too)
for kind in ('uniform', 'wide', 'tiny', 'huge'):
    case(8, 64, 64, 16, kind)
    case(8, 64, 64, 32, kind)
_lib.call('gx_kq_policy', 1)
torch.manual_seed(0)
for N, s in ((224, 16), (224, 32)):
    x = torch.randn(N, 64, s, s, device=DEV)
    w = torch.randn(64, 64, 5, 5, device=DEV) * 0.05
    b = torch.randn(64, device=DEV)
    dy = torch.randn(N, 64, 2 * s, 2 * s, device=DEV)
    fl = 2.0 * N * 64 * 64 * 25 * s * s
    for name, fn in (('fwd', lambda: hip.deconv5x5s2_fwd(x, w, b)), ('dgrad', lambda: hip.deconv5x5s2_dgrad(dy, w))):
        line = '%-6s N%d 64->64 @%d ' % (name, N, s)
        for mode in (1, 2):
            _lib.call('gx_kq_precision', mode)
            t = timeit(fn)
            line += ' | %s %7.1f us %6.1f TF/s' % (NAMES[mode], t, fl / t / 1e6)
        print(line, flush=True)
_lib.call('gx_kq_precision', -1)
