"""Times the projected-gradient GroupNorm backward (+ fused 1x1 weight gradient) of the decoder head: argv N:C:Cout:S"""
import os, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip
for spec in sys.argv[1:]:
    N, C, Co, S = [int(v) for v in spec.split(':')]
    y = torch.randn(N, C, S, S, device='cuda'); gam = torch.rand(C, device='cuda') + 0.5; bet = torch.randn(C, device='cuda') * 0.1
    w = torch.randn(Co, C, device='cuda') * 0.1; b = torch.zeros(Co, device='cuda'); g = torch.randn(N, Co, S, S, device='cuda')
    mean, rstd = hip.gn_relu_fwd(y, gam, bet, 8, 1e-5, None)
    f = lambda: hip.conv1x1_gn_bwd_fused(y, gam, bet, mean, rstd, 8, g, w, b, None, True)
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    t = a.elapsed_time(e) / 5 * 1e3
    print('%-18s %8.1f us   %.2f TB/s of (2 |y| + |dy|)' % (spec, t, 3 * y.numel() * 4 / t / 1e6), flush=True)
