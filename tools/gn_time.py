"""Times the GroupNorm kernels of the decoder / heads at the metric configuration's shapes (HIP events) and checks the
decoder-head backward (gx_gn_relu_bwd_proj with the fused 1x1 weight gradient) against fp64 autograd."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
import torch.nn.functional as F
from genesis_amd import hip_ops as hip


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


torch.manual_seed(0)
for (N, C, H, Co) in [(224, 64, 64, 4), (32, 64, 64, 8), (8, 64, 64, 4)]:
    y = torch.randn(N, C, H, H, device='cuda')
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1
    g = torch.randn(N, Co, H, H, device='cuda')
    w2 = torch.randn(Co, C, device='cuda') * 0.1
    mean, rstd = hip.gn_relu_fwd(y, gamma, beta, 8, 1e-5, None)
    wpart = torch.empty(N, Co, C, device='cuda'); bpart = torch.empty(N, Co, device='cuda')
    fuse = Co == 4
    fn = lambda: hip.gn_relu_bwd_proj(y, gamma, beta, mean, rstd, 8, g, w2, parts=(wpart, bpart) if fuse else (None, None))
    t = timeit(fn)
    mb = (2 * y.numel() + g.numel()) * 4 / 1e6
    print('gn bwd proj N=%d C=%d %dx%d Cout=%d: %7.1f us  %5.2f TB/s (%.0f MB)' % (N, C, H, H, Co, t, mb / t, mb), flush=True)
    if N <= 32:
        dy, dgamma, dbeta, _ = fn()
        yr = y.double().cpu().requires_grad_(); gr, br = gamma.double().cpu().requires_grad_(), beta.double().cpu().requires_grad_()
        wr = w2.double().cpu().requires_grad_()
        a = F.relu(F.group_norm(yr, 8, gr, br, 1e-5))
        out = F.conv2d(a, wr.view(Co, C, 1, 1))
        out.backward(g.double().cpu())
        rel = lambda u, v: float((u.double().cpu() - v).norm() / v.norm())
        print('   dy %.2e dgamma %.2e dbeta %.2e' % (rel(dy, yr.grad), rel(dgamma, gr.grad), rel(dbeta, br.grad)), end='')
        if fuse:
            print('  dW %.2e db %.2e' % (rel(wpart.sum(0), wr.grad), rel(bpart.sum(0), g.double().cpu().sum((0, 2, 3)))))
        else:
            print()
for (N, C, H) in [(224, 64, 32), (224, 64, 16), (32, 64, 64), (32, 128, 16)]:
    y = torch.randn(N, C, H, H, device='cuda')
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1
    out = torch.empty_like(y)
    g = torch.randn_like(y)
    mean, rstd = hip.gn_relu_fwd(y, gamma, beta, 8, 1e-5, (out, 0, 0))
    tf = timeit(lambda: hip.gn_relu_fwd(y, gamma, beta, 8, 1e-5, (out, 0, 0)))
    tb = timeit(lambda: hip.gn_relu_bwd(y, gamma, beta, mean, rstd, 8, (g, 0, 0)))
    mb = y.numel() * 4 / 1e6
    print('gn N=%d C=%d %dx%d: fwd %6.1f us %5.2f TB/s | bwd %6.1f us %5.2f TB/s' % (N, C, H, H, tf, 2 * mb / tf, tb, 3 * mb / tb), flush=True)
