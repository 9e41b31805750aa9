"""Diagnostic (GPU): backward error of the decoder chain, layer by layer, vs fp64 CPU."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from genesis_amd import hip_ops as hip
from oracle import v2_oracle as O

DEV = 'cuda'
torch.manual_seed(0)
N, D, S, K, B = 20, 32, 64, 5, 4
d = S // 16


def rel(a, ref):
    return float((a.detach().cpu().double() - ref.double()).norm() / (ref.double().norm() + 1e-300))


# parameters (torch default-like init)
convs = [torch.nn.ConvTranspose2d(D + 2 if l == 0 else D, D, 5, 2, 2, 1) for l in range(4)]
gns = [torch.nn.GroupNorm(8, D) for _ in range(4)]
outc = torch.nn.Conv2d(D, 4, 1)
z = torch.randn(N, D)
x = torch.rand(B, 3, S, S)
coords = O.pixel_coords(d)


def cpu_chain(dt):
    ps = [(convs[l].weight.detach().to(dt).requires_grad_(), convs[l].bias.detach().to(dt).requires_grad_(),
           gns[l].weight.detach().to(dt).requires_grad_(), gns[l].bias.detach().to(dt).requires_grad_()) for l in range(4)]
    PARAMS[dt] = ps
    h = torch.cat((z.to(dt).view(N, D, 1, 1).expand(-1, -1, d, d), coords.to(dt).expand(N, -1, -1, -1)), 1)
    inter = []
    for l in range(4):
        y = F.conv_transpose2d(h, ps[l][0], ps[l][1], 2, 2, 1)
        y.retain_grad()
        a = F.relu(F.group_norm(y, 8, ps[l][2], ps[l][3], 1e-5))
        a.retain_grad()
        inter.append((h, y, a))
        h = a
    dec = F.conv2d(h, outc.weight.detach().to(dt), outc.bias.detach().to(dt))
    dec.retain_grad()
    ch = dec.chunk(K, 0)
    xr = [torch.sigmoid(c[:, :3]) for c in ch]
    lm = torch.log_softmax(torch.stack([c[:, 3:] for c in ch], 4), 4)
    err = O.x_loss(x.to(dt), [lm[..., k] for k in range(K)], xr, 0.7)
    return inter, dec, err


res = {}
PARAMS = {}
for dt in (torch.float64, torch.float32):
    zz = z
    inter, dec, err = cpu_chain(dt)
    # make first h require grad
    inter[0][0].requires_grad_(True) if inter[0][0].is_leaf else None
    err.mean().backward()
    res[dt] = (inter, dec, err)

# HIP chain
h = torch.cat((z.view(N, D, 1, 1).expand(-1, -1, d, d), coords.expand(N, -1, -1, -1)), 1).contiguous().to(DEV)
saved = []
for l in range(4):
    w, b = convs[l].weight.detach().to(DEV), convs[l].bias.detach().to(DEV)
    y = hip.deconv5x5s2_fwd(h, w, b)
    a = torch.empty_like(y)
    mean, rstd = hip.gn_relu_fwd(y, gns[l].weight.detach().to(DEV), gns[l].bias.detach().to(DEV), 8, 1e-5, (a, 0, 0))
    saved.append((h, y, a, mean, rstd))
    h = a
ow, ob = outc.weight.detach().view(4, D).to(DEV), outc.bias.detach().to(DEV)
dec = hip.conv1x1_fwd(h, ow, ob)
err, _, _, _ = hip.mixture_fwd(x.to(DEV), dec, K, 0.7, True)
i64, d64, e64 = res[torch.float64]
i32, d32, e32 = res[torch.float32]
print('fwd dec   hip %.3e cpu32 %.3e' % (rel(dec, d64), rel(d32, d64)))
print('fwd err   hip %.3e cpu32 %.3e' % (rel(err, e64), rel(e32, e64)))
g_err = torch.full((B,), 1.0 / B, device=DEV)
ddec = hip.mixture_bwd(x.to(DEV), dec, g_err, K, 0.7, True)
print('ddec      hip %.3e cpu32 %.3e' % (rel(ddec, d64.grad), rel(d32.grad, d64.grad)))
da, dow, dob, _ = hip.conv1x1_bwd(h, ddec, ow, ob)
for l in reversed(range(4)):
    hh, y, a, mean, rstd = saved[l]
    print('L%d da     hip %.3e cpu32 %.3e' % (l, rel(da, i64[l][2].grad), rel(i32[l][2].grad, i64[l][2].grad)))
    dy, dgamma, dbeta, dbias = hip.gn_relu_bwd(y, gns[l].weight.detach().to(DEV), gns[l].bias.detach().to(DEV), mean, rstd, 8, (da, 0, 0), None, True)
    print('L%d dy     hip %.3e cpu32 %.3e' % (l, rel(dy, i64[l][1].grad), rel(i32[l][1].grad, i64[l][1].grad)))
    # same GN bwd but fed the exact fp64 upstream gradient rounded to fp32 (isolates this kernel)
    dy2, _, _, _ = hip.gn_relu_bwd(y, gns[l].weight.detach().to(DEV), gns[l].bias.detach().to(DEV), mean, rstd, 8, (i64[l][2].grad.float().to(DEV), 0, 0), None, True)
    print('L%d dy|exact-in  hip %.3e' % (l, rel(dy2, i64[l][1].grad)))
    dw = hip.deconv5x5s2_wgrad(hh, dy)
    p64, p32 = PARAMS[torch.float64][l], PARAMS[torch.float32][l]
    for nm, got, i in (('dw', dw, 0), ('dbias', dbias, 1), ('dgamma', dgamma, 2), ('dbeta', dbeta, 3)):
        print('L%d %-6s hip %.3e cpu32 %.3e' % (l, nm, rel(got, p64[i].grad), rel(p32[i].grad, p64[i].grad)))
    da = hip.deconv5x5s2_dgrad(dy, convs[l].weight.detach().to(DEV))
    da2 = hip.deconv5x5s2_dgrad(i64[l][1].grad.float().to(DEV), convs[l].weight.detach().to(DEV))
    if l > 0:
        print('L%d dgrad|exact-in hip %.3e' % (l, rel(da2, i64[l - 1][2].grad)))
