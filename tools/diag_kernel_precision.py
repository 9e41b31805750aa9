"""Diagnostic (GPU): relative L2 error of every HIP kernel against an fp64 CPU reference, next to the
error of the fp32 CPU op.  Not a test; prints a table."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from genesis_amd import hip_ops as hip

DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def rel(a, ref):
    return float((a.detach().cpu().double() - ref).norm() / (ref.norm() + 1e-300))


def row(name, got, r32, r64):
    print('%-36s hip %10.3e   cpu32 %10.3e' % (name, rel(got, r64), rel(r32, r64)))


def conv_case(N, Cin, Cout, H):
    x, w, dy = rnd(N, Cin, H, H, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=1 / np.sqrt(Cin * 9)), rnd(N, Cout, H, H, seed=3)
    outs = {}
    for dt in (torch.float64, torch.float32):
        xr, wr = x.to(dt).requires_grad_(), w.to(dt).requires_grad_()
        y = F.conv2d(xr, wr, None, 1, 1); y.backward(dy.to(dt))
        outs[dt] = (y.detach(), xr.grad, wr.grad)
    tag = 'conv3x3 %d,%d,%d,%d ' % (N, Cin, Cout, H)
    row(tag + 'fwd', hip.conv3x3_fwd(x.to(DEV), w.to(DEV)), outs[torch.float32][0], outs[torch.float64][0])
    row(tag + 'dgrad', hip.conv3x3_dgrad(dy.to(DEV), w.to(DEV)), outs[torch.float32][1], outs[torch.float64][1])
    row(tag + 'wgrad', hip.conv3x3_wgrad(x.to(DEV), dy.to(DEV)), outs[torch.float32][2], outs[torch.float64][2])


def deconv_case(N, Cin, Cout, H):
    x, w, b, dy = rnd(N, Cin, H, H, seed=4), rnd(Cin, Cout, 5, 5, seed=5, scale=1 / np.sqrt(Cin * 6.25)), rnd(Cout, seed=6), rnd(N, Cout, 2 * H, 2 * H, seed=7)
    outs = {}
    for dt in (torch.float64, torch.float32):
        xr, wr = x.to(dt).requires_grad_(), w.to(dt).requires_grad_()
        y = F.conv_transpose2d(xr, wr, b.to(dt), 2, 2, 1); y.backward(dy.to(dt))
        outs[dt] = (y.detach(), xr.grad, wr.grad)
    tag = 'deconv %d,%d,%d,%d ' % (N, Cin, Cout, H)
    row(tag + 'fwd', hip.deconv5x5s2_fwd(x.to(DEV), w.to(DEV), b.to(DEV)), outs[torch.float32][0], outs[torch.float64][0])
    row(tag + 'dgrad', hip.deconv5x5s2_dgrad(dy.to(DEV), w.to(DEV)), outs[torch.float32][1], outs[torch.float64][1])
    row(tag + 'wgrad', hip.deconv5x5s2_wgrad(x.to(DEV), dy.to(DEV)), outs[torch.float32][2], outs[torch.float64][2])


def gn_case(N, C, H):
    y, gamma, beta, g = rnd(N, C, H, H, seed=8, scale=2.0) + 0.3, 1 + 0.3 * rnd(C, seed=9), 0.2 * rnd(C, seed=10), rnd(N, C, H, H, seed=11)
    outs = {}
    for dt in (torch.float64, torch.float32):
        yr, gr, br = y.to(dt).requires_grad_(), gamma.to(dt).requires_grad_(), beta.to(dt).requires_grad_()
        o = F.relu(F.group_norm(yr, 8, gr, br, 1e-5)); o.backward(g.to(dt))
        outs[dt] = (o.detach(), yr.grad, gr.grad, br.grad)
    out = torch.empty(N, C, H, H, device=DEV)
    mean, rstd = hip.gn_relu_fwd(y.to(DEV), gamma.to(DEV), beta.to(DEV), 8, 1e-5, (out, 0, 0))
    dy, dgamma, dbeta, _ = hip.gn_relu_bwd(y.to(DEV), gamma.to(DEV), beta.to(DEV), mean, rstd, 8, (g.to(DEV), 0, 0))
    tag = 'gn %d,%d,%d ' % (N, C, H)
    for nm, got, i in (('fwd', out, 0), ('dy', dy, 1), ('dgamma', dgamma, 2), ('dbeta', dbeta, 3)):
        row(tag + nm, got, outs[torch.float32][i], outs[torch.float64][i])


def mixture_case(B, S, K):
    from oracle import v2_oracle as O
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(26))
    dec = rnd(K * B, 4, S, S, seed=27, scale=2.0)
    g = rnd(B, seed=28) + 1.5
    outs = {}
    for dt in (torch.float64, torch.float32):
        dr = dec.to(dt).requires_grad_()
        ch = dr.chunk(K, 0)
        xr = [torch.sigmoid(c[:, :3]) for c in ch]
        lm = torch.log_softmax(torch.stack([c[:, 3:] for c in ch], 4), 4)
        err = O.x_loss(x.to(dt), [lm[..., k] for k in range(K)], xr, 0.7)
        (err * g.to(dt)).sum().backward()
        outs[dt] = (err.detach(), dr.grad)
    err, _, _, _ = hip.mixture_fwd(x.to(DEV), dec.to(DEV), K, 0.7, True)
    ddec = hip.mixture_bwd(x.to(DEV), dec.to(DEV), g.to(DEV), K, 0.7, True)
    row('mixture err', err, outs[torch.float32][0], outs[torch.float64][0])
    row('mixture ddec', ddec, outs[torch.float32][1], outs[torch.float64][1])


def conv1x1_case(N, Cin, Cout, S):
    x, w, b, dy = rnd(N, Cin, S, S, seed=29), rnd(Cout, Cin, 1, 1, seed=30, scale=0.2), rnd(Cout, seed=31), rnd(N, Cout, S, S, seed=33)
    outs = {}
    for dt in (torch.float64, torch.float32):
        xr, wr, br = x.to(dt).requires_grad_(), w.to(dt).requires_grad_(), b.to(dt).requires_grad_()
        y = F.conv2d(xr, wr, br); y.backward(dy.to(dt))
        outs[dt] = (y.detach(), xr.grad, wr.grad, br.grad)
    y = hip.conv1x1_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
    dx, dw, db, _ = hip.conv1x1_bwd(x.to(DEV), dy.to(DEV), w.to(DEV), b.to(DEV))
    for nm, got, i in (('fwd', y, 0), ('dx', dx, 1), ('dw', dw, 2), ('db', db, 3)):
        row('conv1x1 %d,%d,%d,%d %s' % (N, Cin, Cout, S, nm), got, outs[torch.float32][i], outs[torch.float64][i])


if __name__ == '__main__':
    conv_case(4, 64, 64, 64); conv_case(4, 256, 128, 8); conv_case(4, 3, 64, 64)
    deconv_case(8, 64, 64, 32); deconv_case(8, 66, 64, 4)
    gn_case(4, 64, 64); gn_case(8, 64, 8)
    mixture_case(4, 64, 5)
    conv1x1_case(8, 64, 4, 64)
