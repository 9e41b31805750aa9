"""Per-kernel parity: every HIP entry point (called through the C ABI) against the plain PyTorch fp32
CPU op it replaces, on seeded inputs, fp32 tolerances stated per test."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

hip = pytest.importorskip('genesis_amd.hip_ops')
DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(a, b, rtol=1e-4, atol=1e-4, msg=''):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, '%s max err %.3e (ref max %.3e)' % (msg, err, ref)


CONV_CASES = [  # N, Cin, Cout, H, W
    (2, 3, 64, 64, 64), (2, 64, 64, 32, 32), (3, 128, 128, 8, 8), (2, 256, 128, 4, 4),
    (2, 128, 64, 64, 64), (1, 64, 64, 128, 128), (5, 16, 32, 16, 16), (2, 8, 8, 2, 2), (33, 32, 16, 4, 4),
    (2, 4, 32, 64, 64), (3, 4, 20, 32, 48),      # four input channels: the forward runs on the vector-ALU input-layer kernel
]


@pytest.mark.parametrize('N,Cin,Cout,H,W', CONV_CASES)
def test_conv3x3(N, Cin, Cout, H, W):
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(Cin * 9))
    dy = rnd(N, Cout, H, W, seed=3)
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    y_ref = F.conv2d(xr, wr, None, 1, 1)
    y_ref.backward(dy)
    y = hip.conv3x3_fwd(x.to(DEV), w.to(DEV))
    close(y, y_ref, 2e-5, 2e-5, 'fwd')
    dx = hip.conv3x3_dgrad(dy.to(DEV), w.to(DEV))
    close(dx, xr.grad, 2e-5, 2e-5, 'dgrad')
    dw = hip.conv3x3_wgrad(x.to(DEV), dy.to(DEV))
    close(dw, wr.grad, 1e-4, 1e-4, 'wgrad')


@pytest.mark.parametrize('act', [None, 'relu', 'elu'])
def test_conv3x3_input_layer_kernel_with_bias_and_activation(act):
    """conv3x3_smallcin_fwd_kernel (Cin = 4: MONet's [x | log-scope] input layer): bias and activation epilogue, a ragged
    channel block (Cout = 20), against torch."""
    N, Cin, Cout, H, W = 3, 4, 20, 32, 48
    x, w, b = rnd(N, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / 6), rnd(Cout, seed=3, scale=0.5)
    ref = F.conv2d(x, w, b, 1, 1)
    ref = F.relu(ref) if act == 'relu' else (F.elu(ref) if act == 'elu' else ref)
    y = hip.conv3x3_bias_act_fwd(x.to(DEV), w.to(DEV), b.to(DEV), act)
    close(y, ref, 2e-5, 2e-5, 'fwd ' + str(act))


DECONV_CASES = [  # N, Cin, Cout, Hin
    (2, 66, 64, 4), (3, 64, 64, 8), (2, 64, 64, 32), (1, 64, 64, 64), (6, 18, 16, 2), (5, 16, 16, 16), (14, 66, 64, 4),
]


@pytest.mark.parametrize('N,Cin,Cout,Hin', DECONV_CASES)
def test_deconv5x5s2(N, Cin, Cout, Hin):
    x = rnd(N, Cin, Hin, Hin, seed=4)
    w = rnd(Cin, Cout, 5, 5, seed=5, scale=1.0 / np.sqrt(Cin * 6.25))
    b = rnd(Cout, seed=6)
    dy = rnd(N, Cout, 2 * Hin, 2 * Hin, seed=7)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = F.conv_transpose2d(xr, wr, br, 2, 2, 1)
    y_ref.backward(dy)
    y = hip.deconv5x5s2_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
    close(y, y_ref, 2e-5, 2e-5, 'fwd')
    dx = hip.deconv5x5s2_dgrad(dy.to(DEV), w.to(DEV))
    close(dx, xr.grad, 2e-5, 2e-5, 'dgrad')
    if Cin > 2:
        dx2 = hip.deconv5x5s2_dgrad(dy.to(DEV), w.to(DEV), Cin - 2)
        close(dx2, xr.grad[:, :Cin - 2], 2e-5, 2e-5, 'dgrad(cin_out)')
    dw = hip.deconv5x5s2_wgrad(x.to(DEV), dy.to(DEV))
    close(dw, wr.grad, 1e-4, 1e-4, 'wgrad')


@pytest.mark.parametrize('N,C,H,W,groups', [(2, 64, 64, 64, 8), (3, 128, 8, 8, 8), (2, 16, 4, 4, 8), (2, 8, 32, 32, 8),
                                            (1, 64, 128, 128, 8), (4, 32, 2, 2, 8), (5, 64, 8, 8, 8), (2, 128, 4, 4, 8),
                                            (2, 16, 8, 8, 16), (3, 24, 4, 8, 8)])
def test_gn_relu_plain(N, C, H, W, groups):
    y = rnd(N, C, H, W, seed=8, scale=2.0) + 0.3
    gamma = 1 + 0.3 * rnd(C, seed=9)
    beta = 0.2 * rnd(C, seed=10)
    g = rnd(N, C, H, W, seed=11)
    yr, gr, br = y.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = F.relu(F.group_norm(yr, groups, gr, br, 1e-5))
    ref.backward(g)
    yd = y.to(DEV)
    out = torch.empty(N, C, H, W, device=DEV)
    mean, rstd = hip.gn_relu_fwd(yd, gamma.to(DEV), beta.to(DEV), groups, 1e-5, (out, 0, 0))
    close(out, ref, 1e-5, 1e-5, 'fwd')
    dy, dgamma, dbeta, dbias = hip.gn_relu_bwd(yd, gamma.to(DEV), beta.to(DEV), mean, rstd, groups,
                                               (g.to(DEV), 0, 0), None, True)
    close(dy, yr.grad, 1e-4, 1e-5, 'dy')
    close(dgamma, gr.grad, 1e-4, 1e-4, 'dgamma')
    close(dbeta, br.grad, 1e-4, 1e-4, 'dbeta')
    close(dbias, yr.grad.sum((0, 2, 3)), 1e-4, 1e-4, 'dbias')


@pytest.mark.parametrize('N,C,H,W', [(2, 16, 8, 8), (2, 64, 128, 128)])      # 128 x 128: slabs of 512 KB (the two-pass kernels)
def test_gn_relu_views(N, C, H, W):
    """Destinations: skip slice of a concat buffer + 2x down-sampled copy; up-sampled slice.
    Gradients gathered from the same views (modules/unet.py:78,86,89)."""
    y = rnd(N, C, H, W, seed=12, scale=2.0)
    gamma, beta = 1 + 0.3 * rnd(C, seed=13), 0.2 * rnd(C, seed=14)
    yr = y.clone().requires_grad_()
    a = F.relu(F.group_norm(yr, 8, gamma, beta, 1e-5))
    cat = torch.cat([torch.zeros(N, 5, H, W), a], 1)
    down = F.interpolate(a, scale_factor=0.5, mode='nearest')
    up = F.interpolate(a, scale_factor=2.0, mode='nearest')
    g_cat, g_down, g_up = rnd(N, 5 + C, H, W, seed=15), rnd(N, C, H // 2, W // 2, seed=16), rnd(N, C + 3, 2 * H, 2 * W, seed=17)
    up_cat = torch.cat([up, torch.zeros(N, 3, 2 * H, 2 * W)], 1)
    ((cat * g_cat).sum() + (down * g_down).sum()).backward(retain_graph=True)
    grad_cd = yr.grad.clone()
    yr.grad = None
    (up_cat * g_up).sum().backward()
    grad_up = yr.grad.clone()

    yd, gd, bd = y.to(DEV), gamma.to(DEV), beta.to(DEV)
    cat_d = torch.zeros(N, 5 + C, H, W, device=DEV)
    down_d = torch.zeros(N, C, H // 2, W // 2, device=DEV)
    mean, rstd = hip.gn_relu_fwd(yd, gd, bd, 8, 1e-5, (cat_d, 5, 0), (down_d, 0, 2))
    close(cat_d, cat, 1e-5, 1e-5, 'cat')
    close(down_d, down, 1e-5, 1e-5, 'down')
    up_d = torch.zeros(N, C + 3, 2 * H, 2 * W, device=DEV)
    hip.gn_relu_fwd(yd, gd, bd, 8, 1e-5, (up_d, 0, 1))
    close(up_d, up_cat, 1e-5, 1e-5, 'up')
    dy, _, _, _ = hip.gn_relu_bwd(yd, gd, bd, mean, rstd, 8, (g_cat.to(DEV), 5, 0), (g_down.to(DEV), 0, 2))
    close(dy, grad_cd, 1e-4, 1e-5, 'dy(cat+down)')
    dy, _, _, _ = hip.gn_relu_bwd(yd, gd, bd, mean, rstd, 8, (g_up.to(DEV), 0, 1))
    close(dy, grad_up, 1e-4, 1e-5, 'dy(up)')


def _icsbp_ref(colour, log_sigma, rand_pixel, K, kernel, seed_idx=None):
    from oracle import v2_oracle as O
    return O.ic_sbp(colour, log_sigma, K - 1, rand_pixel, kernel, seed_idx)


@pytest.mark.parametrize('kernel', ['gaussian', 'laplacian', 'epanechnikov'])
@pytest.mark.parametrize('B,S,K', [(3, 32, 4), (2, 64, 7), (1, 128, 11), (2, 16, 3)])
def test_icsbp(kernel, B, S, K):
    colour = rnd(B, 8, S, S, seed=18, scale=0.7)
    colour[:, -2:] += torch.stack(torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing='ij'))
    rand_pixel = torch.rand(B, 1, S, S, generator=torch.Generator().manual_seed(19))
    log_sigma = torch.tensor(1.0 / (K * np.log(2)), dtype=torch.float64).log()
    cr = colour.clone().requires_grad_()
    lsr = log_sigma.clone().requires_grad_()
    log_m_k, log_s_k, seeds, idxs = _icsbp_ref(cr, lsr, rand_pixel, K, kernel)
    g = rnd(K, B, 1, S, S, seed=20)
    (torch.stack(log_m_k) * g).sum().backward()

    cd, ld, rd = colour.to(DEV), log_sigma.to(DEV), rand_pixel.to(DEV)
    log_m, log_s, seeds_d, idx_d = hip.icsbp_fwd(cd, ld, rd, K, kernel)
    ref_idx = torch.stack(idxs)
    if not torch.equal(idx_d.cpu(), ref_idx):
        # near-tie in the discontinuous argmax: replay with the oracle's seeds
        log_m, log_s, seeds_d, idx_d = hip.icsbp_fwd(cd, ld, rd, K, kernel, ref_idx.to(DEV))
    close(log_m, torch.stack(log_m_k), 1e-5, 2e-5, 'log_m')
    close(log_s, torch.stack(log_s_k), 1e-5, 2e-5, 'log_s')
    close(seeds_d, torch.stack(seeds), 0, 0, 'seeds')
    s = log_m.exp().sum(0)
    assert float((s - 1).abs().max()) < 1e-3  # the reference's own invariant (utils/misc.py:258-270)
    dcol, dls = hip.icsbp_bwd(cd, ld, seeds_d, idx_d, g.to(DEV), kernel)
    close(dcol, cr.grad, 2e-4, 2e-4, 'dcolour')
    close(dls, lsr.grad, 2e-4, 1e-5, 'dlog_sigma')


def test_icsbp_forced_seed_and_first_max():
    B, S, K = 2, 16, 3
    colour = rnd(B, 8, S, S, seed=21)
    rand_pixel = torch.full((B, 1, S, S), 0.5)  # all ties -> first pixel must win (torch.argmax semantics)
    ls = torch.tensor(0.2, dtype=torch.float64).log()
    _, _, _, idx = hip.icsbp_fwd(colour.to(DEV), ls.to(DEV), rand_pixel.to(DEV), K)
    assert idx[0].tolist() == [0, 0]
    forced = torch.tensor([[5, 7], [100, 3]], dtype=torch.int64)
    _, _, seeds, idx = hip.icsbp_fwd(colour.to(DEV), ls.to(DEV), rand_pixel.to(DEV), K, 'gaussian', forced.to(DEV))
    assert torch.equal(idx.cpu(), forced)
    close(seeds[1, 0], colour.flatten(2)[0, :, 100], 0, 0)


@pytest.mark.parametrize('B,C,S,K', [(2, 64, 64, 7), (3, 16, 32, 4), (1, 64, 128, 11), (2, 8, 32, 3)])
def test_maskpool(B, C, S, K):
    f = rnd(B, C, S, S, seed=22).relu()
    log_m = torch.log_softmax(rnd(K, B, 1, S, S, seed=23, scale=3.0), 0)
    fr, lr = f.clone().requires_grad_(), log_m.clone().requires_grad_()
    m = lr.exp()
    S_ref = torch.stack([(m[k] * fr).sum((2, 3)) for k in range(K)], 1)
    ms_ref = torch.stack([m[k].sum((1, 2, 3)) for k in range(K)], 1)
    gS, gms = rnd(B, K, C, seed=24), rnd(B, K, seed=25)
    ((S_ref * gS).sum() + (ms_ref * gms).sum()).backward()
    Sd, msd = hip.maskpool_fwd(f.to(DEV), log_m.to(DEV))
    close(Sd, S_ref, 1e-5, 1e-5, 'S')
    close(msd, ms_ref, 1e-5, 1e-5, 'msum')
    df, dlm = hip.maskpool_bwd(f.to(DEV), log_m.to(DEV), gS.to(DEV), gms.to(DEV))
    close(df, fr.grad, 1e-5, 1e-5, 'df')
    close(dlm, lr.grad, 1e-5, 1e-5, 'dlog_m')


@pytest.mark.parametrize('pixel_bound', [True, False])
@pytest.mark.parametrize('B,S,K', [(2, 64, 7), (3, 32, 4), (1, 128, 11), (2, 8, 1)])
def test_mixture(pixel_bound, B, S, K):
    from oracle import v2_oracle as O
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(26))
    dec = rnd(K * B, 4, S, S, seed=27, scale=2.0)
    dr = dec.clone().requires_grad_()
    chunks = dr.chunk(K, 0)
    x_r_k = [c[:, :3] for c in chunks]
    if pixel_bound:
        x_r_k = [torch.sigmoid(t) for t in x_r_k]
    lm = torch.log_softmax(torch.stack([c[:, 3:] for c in chunks], 4), 4)
    lm_k = [lm[..., k] for k in range(K)]
    err_ref = O.x_loss(x, lm_k, x_r_k, 0.7)
    recon_ref = (torch.stack(lm_k, 4).exp() * torch.stack(x_r_k, 4)).sum(4)
    g = rnd(B, seed=28) + 1.5
    (err_ref * g).sum().backward()
    err, recon, x_r, log_m_r = hip.mixture_fwd(x.to(DEV), dec.to(DEV), K, 0.7, pixel_bound)
    close(err, err_ref, 2e-6, 1e-3, 'err')
    close(recon, recon_ref, 1e-5, 1e-5, 'recon')
    close(x_r, torch.stack(x_r_k), 1e-5, 1e-5, 'x_r')
    close(log_m_r, torch.stack(lm_k), 1e-5, 1e-5, 'log_m_r')
    ddec = hip.mixture_bwd(x.to(DEV), dec.to(DEV), g.to(DEV), K, 0.7, pixel_bound)
    close(ddec, dr.grad, 1e-4, 1e-5, 'ddec')


@pytest.mark.parametrize('N,Cin,Cout,S,gated', [(2, 64, 8, 64, True), (3, 64, 4, 32, False), (2, 16, 8, 32, True),
                                                (7, 64, 4, 64, False), (1, 64, 8, 128, True), (2, 8, 8, 8, False)])
def test_conv1x1(N, Cin, Cout, S, gated):
    x = rnd(N, Cin, S, S, seed=29)
    w = rnd(Cout, Cin, 1, 1, seed=30, scale=0.2)
    b = rnd(Cout, seed=31)
    gate = torch.tensor(0.35) if gated else None
    addend = rnd(Cout, S, S, seed=32) if gated else None
    dy = rnd(N, Cout, S, S, seed=33)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    gr = gate.clone().requires_grad_() if gated else None
    y_ref = F.conv2d(xr, wr, br)
    if gated:
        y_ref = gr * y_ref + addend
    y_ref.backward(dy)
    to = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    y = hip.conv1x1_fwd(to(x), to(w), to(b), to(gate), to(addend))
    close(y, y_ref, 1e-5, 1e-5, 'fwd')
    dx, dw, db, dgate = hip.conv1x1_bwd(to(x), to(dy), to(w), to(b), to(gate))
    close(dx, xr.grad, 1e-5, 1e-5, 'dx')
    close(dw, wr.grad, 1e-4, 1e-4, 'dw')
    close(db, br.grad, 1e-4, 1e-4, 'db')
    if gated:
        close(dgate, gr.grad, 1e-4, 1e-3, 'dgate')


def test_no_cpu_fallback():
    from genesis_amd._lib import GenesisHipError
    with pytest.raises(GenesisHipError):
        hip.conv3x3_fwd(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))


# ------------------------------------------------------------------ ComponentVAE / MONet kernels
@pytest.mark.parametrize('N,Cin,Cout,H,W,act', [(3, 18, 32, 72, 72, 'relu'), (2, 32, 32, 72, 72, 'elu'),
                                                  (2, 16, 8, 40, 40, 'relu'), (5, 8, 16, 12, 20, None),
                                                  (2, 64, 64, 64, 64, 'relu')])
def test_conv3x3_any_grid_bias_act(N, Cin, Cout, H, W, act):
    """conv3x3 + bias + activation on non-power-of-two grids (the 72x72 BroadcastDecoder canvas)."""
    x = rnd(N, Cin, H, W, seed=41)
    w = rnd(Cout, Cin, 3, 3, seed=42, scale=1.0 / np.sqrt(Cin * 9))
    b = rnd(Cout, seed=43, scale=0.3)
    g = rnd(N, Cout, H, W, seed=44)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    pre = F.conv2d(xr, wr, br, 1, 1)
    ref = {'relu': F.relu, 'elu': F.elu, None: lambda t: t}[act](pre)
    ref.backward(g)
    y = hip.conv3x3_bias_act_fwd(x.to(DEV), w.to(DEV), b.to(DEV), act)
    close(y, ref, 2e-5, 2e-5, 'fwd')
    dy, db = hip.bias_act_bwd(y, g.to(DEV), act)
    close(db, br.grad, 1e-4, 1e-4, 'dbias')
    dx = hip.conv3x3_dgrad(dy, w.to(DEV))
    close(dx, xr.grad, 1e-4, 2e-5, 'dgrad')
    dw = hip.conv3x3_wgrad(x.to(DEV), dy)
    close(dw, wr.grad, 1e-4, 1e-4, 'wgrad')


def test_valid_conv_chain_equals_same_conv_on_canvas():
    """modules/decoders.py:25-32: L valid 3x3 convs on the (S+2L)^2 broadcast == 'same' convs on the canvas + crop."""
    L, S, N = 4, 16, 2
    x = rnd(N, 6, S + 2 * L, S + 2 * L, seed=45)
    ws = [rnd(8 if i else 8, 6 if i == 0 else 8, 3, 3, seed=46 + i, scale=0.2) for i in range(L)]
    bs = [rnd(8, seed=56 + i, scale=0.2) for i in range(L)]
    ref = x
    for w, b in zip(ws, bs):
        ref = F.relu(F.conv2d(ref, w, b))           # valid
    h = x.to(DEV)
    for w, b in zip(ws, bs):
        h = hip.conv3x3_bias_act_fwd(h, w.to(DEV), b.to(DEV), 'relu')
    close(h[:, :, L:-L, L:-L], ref, 2e-5, 2e-5, 'valid chain')


@pytest.mark.parametrize('N,Cin,Cout,S,k,stride,pad,act', [(3, 4, 32, 64, 3, 2, 1, 'relu'), (2, 32, 64, 16, 3, 2, 1, 'elu'),
                                                            (2, 64, 64, 8, 3, 2, 1, 'relu'), (2, 5, 7, 10, 5, 1, 2, None),
                                                            (2, 3, 6, 9, 3, 1, 0, 'relu')])
def test_conv2d_direct(N, Cin, Cout, S, k, stride, pad, act):
    x = rnd(N, Cin, S, S, seed=61)
    w = rnd(Cout, Cin, k, k, seed=62, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(Cout, seed=63, scale=0.3)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    pre = F.conv2d(xr, wr, br, stride, pad)
    ref = {'relu': F.relu, 'elu': F.elu, None: lambda t: t}[act](pre)
    g = rnd(*ref.shape, seed=64)
    ref.backward(g)
    y = hip.conv2d_direct_fwd(x.to(DEV), w.to(DEV), b.to(DEV), act, stride, pad)
    close(y, ref, 2e-5, 2e-5, 'fwd')
    dy, db = hip.bias_act_bwd(y, g.to(DEV), act)
    close(db, br.grad, 1e-4, 1e-4, 'dbias')
    dx = hip.conv2d_direct_dgrad(dy, w.to(DEV), S, S, stride, pad)
    close(dx, xr.grad, 1e-4, 2e-5, 'dgrad')
    dw = hip.conv2d_direct_wgrad(x.to(DEV), dy, k, stride, pad)
    close(dw, wr.grad, 1e-4, 1e-4, 'wgrad')


@pytest.mark.parametrize('B,S,K,std1', [(2, 64, 7, 0.7), (3, 32, 4, 0.5)])
def test_mixture_external_weights(B, S, K, std1):
    """MONet: mixing weights are the attention masks; first slot may use its own std (monet_config.py:69-72)."""
    from oracle import v2_oracle as O
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(71))
    dec = rnd(K * B, 4, S, S, seed=72, scale=2.0)
    log_w = torch.log_softmax(rnd(K, B, 1, S, S, seed=73, scale=3.0), 0)
    std = 0.7 * torch.ones(1, 1, 1, 1, K)
    std[..., 0] = std1
    dr, lr = dec.clone().requires_grad_(), log_w.clone().requires_grad_()
    x_r_k = [torch.sigmoid(c[:, :3]) for c in dr.chunk(K, 0)]
    err_ref = O.x_loss(x, list(lr.unbind(0)), x_r_k, std)
    recon_ref = (torch.stack(list(lr.unbind(0)), 4).exp() * torch.stack(x_r_k, 4)).sum(4)
    g = rnd(B, seed=74) + 1.5
    (err_ref * g).sum().backward()
    err, recon, x_r = hip.mixture_w_fwd(x.to(DEV), dec.to(DEV), log_w.to(DEV), K, std1, 0.7, True)
    close(err, err_ref, 2e-6, 1e-3, 'err')
    close(recon, recon_ref, 1e-5, 1e-5, 'recon')
    ddec, dlw = hip.mixture_w_bwd(x.to(DEV), dec.to(DEV), log_w.to(DEV), g.to(DEV), K, std1, 0.7, True)
    close(ddec, dr.grad, 1e-4, 1e-5, 'ddec')
    close(dlw, lr.grad, 1e-4, 1e-5, 'dlog_w')


@pytest.mark.parametrize('N,C,S', [(3, 32, 16), (2, 64, 64), (2, 128, 4)])
def test_instance_norm_is_groupnorm_with_c_groups(N, C, S):
    """ConvINReLU (modules/blocks.py:151-157): InstanceNorm2d(affine) == GroupNorm with groups = C."""
    y = rnd(N, C, S, S, seed=81, scale=2.0) + 0.2
    gamma, beta, g = 1 + 0.3 * rnd(C, seed=82), 0.2 * rnd(C, seed=83), rnd(N, C, S, S, seed=84)
    yr, gr, br = y.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = F.relu(F.instance_norm(yr, weight=gr, bias=br, eps=1e-5))
    ref.backward(g)
    out = torch.empty(N, C, S, S, device=DEV)
    mean, rstd = hip.gn_relu_fwd(y.to(DEV), gamma.to(DEV), beta.to(DEV), C, 1e-5, (out, 0, 0))
    close(out, ref, 1e-5, 1e-5, 'fwd')
    dy, dgamma, dbeta, _ = hip.gn_relu_bwd(y.to(DEV), gamma.to(DEV), beta.to(DEV), mean, rstd, C, (g.to(DEV), 0, 0))
    close(dy, yr.grad, 1e-4, 1e-5, 'dy')
    close(dgamma, gr.grad, 1e-4, 1e-4, 'dgamma')
    close(dbeta, br.grad, 1e-4, 1e-4, 'dbeta')


@pytest.mark.parametrize('norm', ['bn', 'in', None])
@pytest.mark.parametrize('N,C,S', [(4, 32, 16), (3, 64, 8), (6, 32, 64)])
def test_gated_norm(norm, N, C, S):
    """sylvester gated unit (layers.py:40-54): norm_h(h) * sigmoid(norm_g(g)), BatchNorm in training mode."""
    y = rnd(N, 2 * C, S, S, seed=91, scale=2.0)
    bias = rnd(2 * C, seed=92, scale=0.5)
    prm = [1 + 0.3 * rnd(C, seed=93), 0.2 * rnd(C, seed=94), 1 + 0.3 * rnd(C, seed=95), 0.2 * rnd(C, seed=96)]
    g = rnd(N, C, S, S, seed=97)
    yr, br = y.clone().requires_grad_(), bias.clone().requires_grad_()
    pr = [t.clone().requires_grad_() for t in prm]
    h, gt = (yr + br.view(1, -1, 1, 1)).chunk(2, 1)
    if norm == 'bn':
        h = F.batch_norm(h, None, None, pr[0], pr[1], True, 0.1, 1e-5)
        gt = F.batch_norm(gt, None, None, pr[2], pr[3], True, 0.1, 1e-5)
    elif norm == 'in':
        h = F.instance_norm(h, weight=pr[0], bias=pr[1], eps=1e-5)
        gt = F.instance_norm(gt, weight=pr[2], bias=pr[3], eps=1e-5)
    ref = h * torch.sigmoid(gt)
    ref.backward(g)
    to = lambda t: t.to(DEV)  # noqa: E731
    args = [to(t) for t in prm] if norm else [None] * 4
    out, stats = hip.gated_norm_fwd(to(y), to(bias), norm, *args)
    close(out, ref, 1e-5, 1e-5, 'fwd')
    dy, dgh, dbh, dgg, dbg, dbias = hip.gated_norm_bwd(to(y), to(bias), norm, *args, stats, to(g))
    close(dy, yr.grad, 1e-4, 1e-5, 'dy')
    if norm:
        for got, r, nm in ((dgh, pr[0], 'dgamma_h'), (dbh, pr[1], 'dbeta_h'), (dgg, pr[2], 'dgamma_g'), (dbg, pr[3], 'dbeta_g')):
            close(got, r.grad, 1e-4, 1e-4, nm)
    # under a norm the conv-bias gradient is analytically zero (the reference carries ~1e-4 of cancellation noise)
    close(dbias, br.grad, 1e-4, 1e-3 if norm else 1e-4, 'dbias')


def test_mixture_external_weights_rgb_only():
    """GENESIS: decoder emits RGB only (3 channels per slot), masks come from the attention process."""
    from oracle import v2_oracle as O
    B, S, K = 2, 32, 3
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(75))
    dec = rnd(K * B, 3, S, S, seed=76, scale=2.0)
    log_w = torch.log_softmax(rnd(K, B, 1, S, S, seed=77, scale=3.0), 0)
    std = 0.7 * torch.ones(1, 1, 1, 1, K)
    dr, lr = dec.clone().requires_grad_(), log_w.clone().requires_grad_()
    x_r_k = [torch.sigmoid(c) for c in dr.chunk(K, 0)]
    err_ref = O.x_loss(x, list(lr.unbind(0)), x_r_k, std)
    (err_ref * 1.3).sum().backward()
    err, recon, x_r = hip.mixture_w_fwd(x.to(DEV), dec.to(DEV), log_w.to(DEV), K, 0.7, 0.7, True)
    close(err, err_ref, 2e-6, 1e-3, 'err')
    ddec, dlw = hip.mixture_w_bwd(x.to(DEV), dec.to(DEV), log_w.to(DEV), torch.full((B,), 1.3, device=DEV), K, 0.7, 0.7, True)
    close(ddec, dr.grad, 1e-4, 1e-5, 'ddec')
    close(dlw, lr.grad, 1e-4, 1e-5, 'dlog_w')


@pytest.mark.parametrize('B,K,D,prior', [(32, 7, 64, True), (3, 5, 16, True), (4, 1, 64, False), (2, 3, 80, True)])
def test_latent_posterior_and_prior(B, K, D, prior):
    """Posterior sample / log_q / log_p launches vs the torch ops of models/genesisv2_config.py:154-160 and
    models/genesis_config.py:288-343 (Normal.log_prob, to_sigma, to_prior_sigma); fp32, rtol 1e-5 on values,
    1e-4 on gradients."""
    from torch.distributions import Normal
    from genesis_amd import functions as fn
    zh = rnd(B, K, 2 * D, seed=1, scale=2.0)
    eps = torch.randn(K, B, D, generator=torch.Generator().manual_seed(2))
    lin = rnd(K - 1, B, 2 * D, seed=3, scale=2.0) if (prior and K > 1) else None
    w = [rnd(K, B, D, seed=4), rnd(K, B, D, seed=5), rnd(K, B, D, seed=6), rnd(K, B, seed=7), rnd(K, B, seed=8)]

    def ref(zh_, lin_):
        mu, sp = zh_.chunk(2, dim=-1)
        sigma = F.softplus(sp + 0.5) + 1e-8
        mu, sigma = mu.transpose(0, 1), sigma.transpose(0, 1)
        z = mu + sigma * eps.to(zh_.dtype)
        log_q = Normal(mu, sigma).log_prob(z).sum(2)
        if lin_ is not None:
            mr, sr = lin_.chunk(2, dim=2)
            lp = Normal(torch.tanh(mr), torch.sigmoid(sr + 4.0) + 1e-4).log_prob(z[1:]).sum(2)
            log_p = torch.cat((Normal(0., 1.).log_prob(z[:1]).sum(2), lp), 0)
        else:
            log_p = Normal(0., 1.).log_prob(z).sum(2)
        return z, mu, sigma, log_q, log_p

    zr = zh.double().requires_grad_()
    lr = lin.double().requires_grad_() if lin is not None else None
    outs_ref = ref(zr, lr)
    loss_ref = sum((o * wi.double()).sum() for o, wi in zip(outs_ref, w))
    loss_ref.backward()

    zg = zh.to(DEV).requires_grad_()
    lg = lin.to(DEV).requires_grad_() if lin is not None else None
    z, mu, sigma, log_q = fn.PosteriorFn.apply(zg, eps.to(DEV))
    log_p = fn.PriorLogPFn.apply(z, lg)
    kl = fn.PriorLogPFn.apply(z.detach(), None if lg is None else lg.detach(), log_q.detach())
    close(kl, outs_ref[3] - outs_ref[4], rtol=1e-5, atol=1e-4, msg='kl = log_q - log_p')
    outs = (z, mu, sigma, log_q, log_p)
    for o, r, name in zip(outs, outs_ref, ('z', 'mu', 'sigma', 'log_q', 'log_p')):
        close(o, r, rtol=1e-5, atol=1e-5, msg=name)
    loss = sum((o * wi.to(DEV)).sum() for o, wi in zip(outs, w))
    loss.backward()
    close(zg.grad, zr.grad, rtol=1e-4, atol=1e-5, msg='dzh')
    if lin is not None:
        close(lg.grad, lr.grad, rtol=1e-4, atol=1e-5, msg='dlin')

    # only z and log_q used (the training step): the unused outputs' gradients arrive as None
    zg2 = zh.to(DEV).requires_grad_()
    z2, _, _, lq2 = fn.PosteriorFn.apply(zg2, eps.to(DEV))
    ((z2 * w[0].to(DEV)).sum() + (lq2 * w[3].to(DEV)).sum()).backward()
    zr2 = zh.double().requires_grad_()
    o2 = ref(zr2, None)
    ((o2[0] * w[0].double()).sum() + (o2[3] * w[3].double()).sum()).backward()
    close(zg2.grad, zr2.grad, rtol=1e-4, atol=1e-5, msg='dzh (z, log_q only)')


@pytest.mark.parametrize('M,N,K,act,bias', [
    (32, 128, 2048, 'relu', True), (32, 2048, 128, 'relu', True), (224, 128, 128, 'relu', True),
    (224, 128, 64, None, False), (192, 128, 256, None, True), (7, 5, 3, None, True), (33, 17, 70, 'relu', True),
    (16, 16, 16, None, False), (224, 256, 1024, 'elu', True), (33, 17, 70, 'elu', True), (224, 128, 64, 'elu', False)])
def test_linear(M, N, K, act, bias):
    """Dense-layer kernels vs F.linear (+ReLU) and its autograd in fp64; fp32 MFMA, rtol 1e-5 fwd, 1e-4 grads."""
    from genesis_amd import functions as fn
    x = rnd(M, K, seed=1)
    w = rnd(N, K, seed=2, scale=1.0 / np.sqrt(K))
    b = rnd(N, seed=3) if bias else None
    g = rnd(M, N, seed=4)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    br = b.double().requires_grad_() if bias else None
    yr = F.linear(xr, wr, br)
    if act == 'relu':
        yr = F.relu(yr)
    if act == 'elu':
        yr = F.elu(yr)
    (yr * g.double()).sum().backward()
    xg, wg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    bg = b.to(DEV).requires_grad_() if bias else None
    y = fn.linear(xg, wg, bg, act)
    close(y, yr, rtol=1e-5, atol=1e-5, msg='y')
    (y * g.to(DEV)).sum().backward()
    close(xg.grad, xr.grad, rtol=1e-4, atol=1e-5, msg='dx')
    close(wg.grad, wr.grad, rtol=1e-4, atol=1e-5, msg='dw')
    if bias:
        close(bg.grad, br.grad, rtol=1e-4, atol=1e-5, msg='db')


@pytest.mark.parametrize('kind,N,Cin,Cout,S', [('conv3x3', 32, 64, 64, 64), ('conv3x3', 8, 128, 64, 32), ('conv3x3', 4, 40, 72, 16),
                                                ('deconv', 56, 64, 64, 32), ('deconv', 16, 64, 64, 16), ('deconv', 9, 24, 40, 8)])
def test_weight_gradients_on_the_bf16_pipe_keep_fp32_accuracy(kind, N, Cin, Cout, S):
    """gx_wgq_precision: the LDS-DMA weight-gradient kernels form every fp32 product from six bf16 piece products on the
    bf16 matrix pipe (default) or multiply on the fp32 pipe.  Both against autograd in fp64: the bf16-pipe error must
    stay within 1.5 x the fp32-pipe error + 1e-7 (measured: equal or smaller), and within the suite's 1e-4 bar."""
    from genesis_amd import hip_ops as hip, _lib
    if kind == 'conv3x3':
        x, dy = rnd(N, Cin, S, S, seed=1), rnd(N, Cout, S, S, seed=2)
        w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
        (F.conv2d(x.double(), w, None, 1, 1) * dy.double()).sum().backward()
        run = lambda: hip.conv3x3_wgrad(x.to(DEV), dy.to(DEV))  # noqa: E731
    else:
        x, dy = rnd(N, Cin, S, S, seed=1), rnd(N, Cout, 2 * S, 2 * S, seed=2)
        w = torch.zeros(Cin, Cout, 5, 5, dtype=torch.float64, requires_grad=True)
        (F.conv_transpose2d(x.double(), w, None, 2, 2, 1) * dy.double()).sum().backward()
        run = lambda: hip.deconv5x5s2_wgrad(x.to(DEV), dy.to(DEV))  # noqa: E731
    ref = w.grad
    err = {}
    try:
        for mode in (0, 1):
            _lib.call('gx_wgq_precision', mode)
            got = run().double().cpu()
            err[mode] = float((got - ref).norm() / ref.norm())
    finally:
        _lib.call('gx_wgq_precision', -1)
    print('%s N=%d %d->%d @%d: relative L2 error fp32 pipe %.3e, bf16 pipe (6 terms) %.3e' % (kind, N, Cin, Cout, S, err[0], err[1]))
    assert err[1] <= 1.5 * err[0] + 1e-7 and err[1] < 1e-4, err


@pytest.mark.parametrize('N,Cin,Co1,Co2,S', [(32, 64, 64, 64, 64), (2, 16, 64, 24, 16), (3, 24, 128, 64, 32)])
def test_conv3x3_pair_one_launch_for_two_layers_on_one_input(N, Cin, Co1, Co2, S):
    """gx_conv3x3_pair_fwd / _dgrad (seg_head + feat_head[0] on the encoder features as one Winograd layer): the forward
    outputs are bit-equal to the single-layer Winograd kernel's (same chunk order per output channel); the data gradient
    equals dgrad(dy1, w1) + dgrad(dy2, w2) against fp64 conv_transpose2d at the single kernel's accuracy."""
    from genesis_amd import hip_ops as hip
    x = rnd(N, Cin, S, S, seed=1).to(DEV)
    w1 = rnd(Co1, Cin, 3, 3, seed=2, scale=0.1).to(DEV)
    w2 = rnd(Co2, Cin, 3, 3, seed=3, scale=0.1).to(DEV)
    assert hip.conv3x3_pair_supported(x, w1, w2)
    y1, y2, ws = hip.conv3x3_pair_fwd(x, w1, w2)
    assert torch.equal(y1, hip.conv3x3_wino(x, w1, 0)) and torch.equal(y2, hip.conv3x3_wino(x, w2, 0))
    d1, d2 = rnd(N, Co1, S, S, seed=4).to(DEV), rnd(N, Co2, S, S, seed=5).to(DEV)
    ref = F.conv_transpose2d(d1.double(), w1.double(), None, 1, 1) + F.conv_transpose2d(d2.double(), w2.double(), None, 1, 1)
    for packed in (ws, None):
        dx = hip.conv3x3_pair_dgrad(d1, d2, w1, w2, packed)
        close(dx, ref, rtol=2e-5, atol=2e-5, msg='dx (packed weights %s)' % ('reused' if packed is not None else 'fresh'))
    assert not hip.conv3x3_pair_supported(x, w1[:Co1 - 8], w2)      # the first layer must fill whole 64-channel tiles


@pytest.mark.parametrize('N,D,Cout,d', [(224, 64, 64, 4), (5, 16, 24, 2), (3, 8, 16, 8), (2, 4, 8, 1)])
def test_broadcast_deconv_as_matrix_product(N, D, Cout, d):
    """The decoder's first layer on the broadcast latent (models/genesisv2_config.py:89-90) computed as
    z @ (tap-summed weights) + (bias + coordinate channels), against ConvTranspose2d(k5,s2,p2,op1) on the materialised
    canvas in fp64 -- forward, dz, dw (all D + 2 input channels) and db; rtol 1e-5 fwd, 1e-4 grads."""
    from genesis_amd import hip_ops as hip
    z = rnd(N, D, seed=1)
    w = rnd(D + 2, Cout, 5, 5, seed=2, scale=0.2)
    b = rnd(Cout, seed=3)
    g = rnd(N, Cout, 2 * d, 2 * d, seed=4)
    lin = torch.linspace(-1, 1, d) if d > 1 else torch.zeros(1)
    coords = torch.stack((lin.view(d, 1).expand(d, d), lin.view(1, d).expand(d, d)), 0).unsqueeze(0).contiguous()
    zr, wr, br = z.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    canvas = torch.cat((zr.view(N, D, 1, 1).expand(N, D, d, d), coords.double().expand(N, 2, d, d)), 1)
    yr = F.conv_transpose2d(canvas, wr, br, 2, 2, 1)
    (yr * g.double()).sum().backward()
    zg, wg, cg = z.to(DEV), w.to(DEV), coords.to(DEV)
    wz, bz = hip.bcast_deconv_pack(wg, b.to(DEV), cg)
    y = hip.linear_fwd(zg, wz, bz).view(N, Cout, 2 * d, 2 * d)
    close(y, yr, rtol=1e-5, atol=1e-5, msg='y')
    dz, dwz, dbz = hip.linear_bwd(zg, wz, None, g.to(DEV).view(N, -1), None)
    dw, db = hip.bcast_deconv_unpack(dwz, dbz, cg, Cout, want_db=True)
    close(dz, zr.grad, rtol=1e-4, atol=1e-5, msg='dz')
    close(dw, wr.grad, rtol=1e-4, atol=1e-5, msg='dw')
    close(db, br.grad, rtol=1e-4, atol=1e-5, msg='db')


@pytest.mark.parametrize('M,N,K,act', [(32, 2048, 128, 'relu'), (192, 1024, 64, None), (7, 20, 12, 'relu'), (40, 36, 20, 'elu')])
def test_linear_strided_and_accumulating(M, N, K, act):
    """gx_linear_fwd_ld / gx_linear_bwd_ex: operands that are column ranges of wider buffers (the UNet MLP's output in
    the concat buffer, z[:-1] inside z), dx added in place, db written twice -- against the contiguous entry points,
    which must give the same bits (same tiles, same reduction order)."""
    from genesis_amd import hip_ops as hip
    x = rnd(M, K, seed=1).to(DEV)
    w = rnd(N, K, seed=2, scale=1.0 / np.sqrt(K)).to(DEV)
    b = rnd(N, seed=3).to(DEV)
    g = rnd(M, N, seed=4).to(DEV)
    y0 = hip.linear_fwd(x, w, b, act)
    wide_x = torch.full((M, K + 12), 7.0, device=DEV); wide_x[:, 4:4 + K] = x
    wide_y = torch.full((M, N + 8), -3.0, device=DEV)
    y1 = hip.linear_fwd(wide_x[:, 4:4 + K], w, b, act, out=wide_y[:, 8:])
    assert torch.equal(y1, y0) and float(wide_y[:, :8].min()) == -3.0 == float(wide_y[:, :8].max())
    dx0, dw0, db0 = hip.linear_bwd(x, w, y0 if act else None, g, act)
    wide_g = torch.zeros(M, N + 8, device=DEV); wide_g[:, 8:] = g
    base = rnd(M, K + 12, seed=9).to(DEV)
    acc = base.clone()
    db2 = torch.empty(N, device=DEV)
    dx1, dw1, db1 = hip.linear_bwd(wide_x[:, 4:4 + K], w, wide_y[:, 8:] if act else None, wide_g[:, 8:], act,
                                   out_db2=db2, accumulate_dx=acc[:, 4:4 + K])
    assert torch.equal(dw1, dw0) and torch.equal(db1, db0) and torch.equal(db2, db0)
    assert torch.equal(acc[:, 4:4 + K], base[:, 4:4 + K] + dx0)
    assert torch.equal(acc[:, :4], base[:, :4]) and torch.equal(acc[:, 4 + K:], base[:, 4 + K:])
    # dw / db / db2 added to what the destinations hold (a parameter used again in one iteration)
    dwa, dba, db2a = torch.full((N, K), 0.5, device=DEV), torch.full((N,), -1.5, device=DEV), torch.full((N,), 2.0, device=DEV)
    hip.linear_bwd(x, w, y0 if act else None, g, act, out_dw=dwa, out_db=dba, out_db2=db2a, accumulate_dw=True)
    assert torch.equal(dwa, dw0 + 0.5) and torch.equal(dba, db0 - 1.5) and torch.equal(db2a, db0 + 2.0)
    # (the launch without dx has its own workgroup size, i.e. its own summation order)
    _, dw2, db2_ = hip.linear_bwd(x, w, y0 if act else None, g, act, need_dx=False)
    hip.linear_bwd(x, w, y0 if act else None, g, act, need_dx=False, out_dw=dwa, out_db=dba, accumulate_dw=True)
    assert torch.equal(dwa, (dw0 + 0.5) + dw2) and torch.equal(dba, (db0 - 1.5) + db2_)
    with pytest.raises(Exception):
        hip.linear_fwd(wide_x.t()[4:4 + K].t()[:, ::2], w[:, ::2].contiguous(), b, act)      # rows not contiguous


@pytest.mark.parametrize('K,B,D,H', [(7, 32, 64, 256), (3, 5, 16, 32), (2, 4, 8, 16)])
def test_ar_prior_kl_node_equals_the_chained_functions(K, B, D, H):
    """ARPriorKLFn (one autograd node for LSTM -> Linear -> log-density KL) against LSTMFn -> linear -> PriorLogPFn:
    the same kernels in the same order, so values and parameter gradients are bit-equal; dz differs only by the order
    in which its two contributions are added (round-off)."""
    from genesis_amd import functions as fn
    g0 = torch.Generator().manual_seed(3)
    mk = lambda *s: ((torch.rand(*s, generator=g0) * 2 - 1) * 0.3).to(DEV)  # noqa: E731
    params = [mk(4 * H, D), mk(4 * H, H), mk(4 * H), mk(4 * H), mk(2 * D, H), mk(2 * D)]
    z, log_q, gk = rnd(K, B, D, seed=1).to(DEV), rnd(K, B, seed=2).to(DEV), rnd(K, B, seed=3).to(DEV)

    def run(fused):
        ps = [p.clone().requires_grad_() for p in params]
        zz, lq = z.clone().requires_grad_(), log_q.clone().requires_grad_()
        if fused:
            kl = fn.ARPriorKLFn.apply(zz, lq, *ps)
        else:
            h = fn.LSTMFn.apply(zz[:-1], *ps[:4])
            kl = fn.PriorLogPFn.apply(zz, fn.linear(h, ps[4], ps[5]), lq)
        (kl * gk).sum().backward()
        return [kl.detach(), zz.grad, lq.grad] + [p.grad for p in ps]
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    for u, v in zip(a[3:], b[3:]):
        assert torch.equal(u, v)
    close(a[1], b[1], rtol=1e-6, atol=1e-7, msg='dz')


@pytest.mark.parametrize('T,B,D,H', [(6, 32, 64, 256), (1, 3, 16, 16), (4, 17, 8, 32)])
def test_lstm(T, B, D, H):
    """Fused LSTM (dense input projection + one launch per step) vs nn.LSTM in fp64; rtol 1e-5 fwd, 1e-4 grads."""
    from genesis_amd import functions as fn
    ref = torch.nn.LSTM(D, H).double()
    g0 = torch.Generator().manual_seed(5)
    for p in ref.parameters():
        p.data.copy_((torch.rand(p.shape, generator=g0) * 2 - 1) * 0.3)
    x = rnd(T, B, D, seed=1)
    g = rnd(T, B, H, seed=2)
    xr = x.double().requires_grad_()
    out_ref, _ = ref(xr)
    (out_ref * g.double()).sum().backward()
    ps = [p.detach().float().to(DEV).requires_grad_() for p in
          (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    xg = x.to(DEV).requires_grad_()
    out = fn.LSTMFn.apply(xg, *ps)
    close(out, out_ref, rtol=1e-5, atol=1e-5, msg='h')
    (out * g.to(DEV)).sum().backward()
    close(xg.grad, xr.grad, rtol=1e-4, atol=1e-5, msg='dx')
    for p, r, name in zip(ps, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0),
                          ('dw_ih', 'dw_hh', 'db_ih', 'db_hh')):
        close(p.grad, r.grad, rtol=1e-4, atol=1e-5, msg=name)


@pytest.mark.parametrize('T,B,D,H', [(6, 32, 64, 256), (10, 64, 64, 256), (2, 3, 16, 16), (4, 17, 8, 32), (16, 32, 8, 64)])
def test_lstm_whole_sequence_launch_equals_the_step_launches(T, B, D, H):
    """gx_lstm_seq_fwd / _bwd: all T steps in one launch each way (grid-wide barriers between the steps) against the T
    gx_lstm_step_* launches they replace -- the same arithmetic in the same order: every output bit for bit.  Repeated five
    times on the same barrier counters (a launch has to leave them zero), then 20 replays of a HIP graph holding both."""
    from genesis_amd import hip_ops as hip
    assert T <= hip.lstm_seq_capacity(B, H), (T, hip.lstm_seq_capacity(B, H))
    gx3 = rnd(T, B, 4 * H, seed=1).to(DEV)
    w_hh = rnd(4 * H, H, seed=2, scale=0.2).to(DEV)
    b_hh = rnd(4 * H, seed=3, scale=0.2).to(DEV)
    g = rnd(T, B, H, seed=4).to(DEV)

    def buffers():
        return [torch.full((T, B, 4 * H), float('nan'), device=DEV), torch.full((T, B, H), float('nan'), device=DEV),
                torch.full((T, B, H), float('nan'), device=DEV), torch.full((T, B, 4 * H), float('nan'), device=DEV),
                torch.full((2, B, H), float('nan'), device=DEV)]
    act, c, h, dgates, dc = buffers()
    for t in range(T):
        hip.lstm_step_fwd(gx3[t], h[t - 1] if t else None, c[t - 1] if t else None, w_hh, b_hh, act[t], c[t], h[t])
    for t in reversed(range(T)):
        hip.lstm_step_bwd(g[t], dgates[t + 1] if t + 1 < T else None, w_hh, act[t], c[t], c[t - 1] if t else None,
                          dc[(t + 1) & 1] if t + 1 < T else None, dgates[t], dc[t & 1])
    ref = [act, c, h, dgates]
    assert all(bool(torch.isfinite(r).all()) for r in ref)
    for rep in range(5):
        act2, c2, h2, dg2, dc2 = buffers()
        hip.lstm_seq_fwd(gx3, w_hh, b_hh, act2, c2, h2)
        hip.lstm_seq_bwd(g, w_hh, act2, c2, dg2, dc2)
        for name, a, b in zip(('act', 'c', 'h', 'dgates'), (act2, c2, h2, dg2), ref):
            assert torch.equal(a, b), (rep, name, float((a - b).abs().max()))
    assert int(hip._lstm_bar(torch.device(DEV), 0).abs().sum()) == 0 and int(hip._lstm_bar(torch.device(DEV), 1).abs().sum()) == 0
    # in a HIP graph (the training step replays it): the counters are part of the replayed state
    side = torch.cuda.Stream()
    act3, c3, h3, dg3, dc3 = buffers()
    with torch.cuda.stream(side):
        hip.lstm_seq_fwd(gx3, w_hh, b_hh, act3, c3, h3)         # (this stream's counters exist before the capture)
        hip.lstm_seq_bwd(g, w_hh, act3, c3, dg3, dc3)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            hip.lstm_seq_fwd(gx3, w_hh, b_hh, act3, c3, h3)
            hip.lstm_seq_bwd(g, w_hh, act3, c3, dg3, dc3)
    for rep in range(20):
        dg3.fill_(float('nan')); h3.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(h3, h) and torch.equal(dg3, dgates), rep


def test_kl_mode_gradients():
    """PriorLogPFn with log_q returns log_q - log_p and routes +g to log_q, -g into the prior terms."""
    from genesis_amd import functions as fn
    K, B, D = 4, 5, 16
    z = rnd(K, B, D, seed=1).to(DEV).requires_grad_()
    lin = rnd(K - 1, B, 2 * D, seed=2).to(DEV).requires_grad_()
    lq = rnd(K, B, seed=3).to(DEV).requires_grad_()
    w = rnd(K, B, seed=4).to(DEV)
    (fn.PriorLogPFn.apply(z, lin, lq) * w).sum().backward()
    gz, gl, gq = z.grad.clone(), lin.grad.clone(), lq.grad.clone()
    z.grad = lin.grad = lq.grad = None
    ((lq - fn.PriorLogPFn.apply(z, lin)) * w).sum().backward()
    close(gz, z.grad, rtol=1e-6, atol=1e-7, msg='dz'); close(gl, lin.grad, rtol=1e-6, atol=1e-7, msg='dlin')
    close(gq, lq.grad, rtol=0, atol=0, msg='dlog_q')


@pytest.mark.parametrize('B,R', [(32, 7), (5, 0), (3, 1)])
def test_elbo(B, R):
    """Loss aggregation launch vs train.py:226-242 in torch fp64."""
    from genesis_amd import functions as fn
    err = (rnd(B, seed=1) * 100 + 500).to(DEV).requires_grad_()
    kl = (rnd(R, B, seed=2) * 10).to(DEV).requires_grad_() if R else None
    beta = torch.tensor([0.37], device=DEV)
    tail = torch.zeros(2, device=DEV)
    loss, out = fn.ElboFn.apply(err, kl, beta, tail)
    e = err.detach().cpu().double().mean()
    k = kl.detach().cpu().double().mean(1).sum() if R else torch.zeros((), dtype=torch.float64)
    close(out, torch.stack((e + 0.37 * k, e + k, e, k, torch.tensor(0.37, dtype=torch.float64))), rtol=1e-6, atol=1e-6)
    close(tail, torch.stack((e, k)), rtol=1e-6, atol=1e-6)
    assert float(loss) == float(out[0])
    loss.backward()
    close(err.grad, torch.full((B,), 1.0 / B), rtol=1e-6, atol=0)
    if R:
        close(kl.grad, torch.full((R, B), 0.37 / B), rtol=1e-6, atol=0)


@pytest.mark.parametrize('R,C', [(224, 128), (6, 32), (5, 200)])
def test_pooled_head(R, C):
    """(lin + msum b)/(msum + 1e-5) -> LayerNorm, vs the torch ops of models/genesisv2_config.py:146-154 + z_head[0]
    in fp64; rtol 1e-5 forward, 1e-4 gradients."""
    from genesis_amd import functions as fn
    lin, msum = rnd(R, C, seed=1, scale=30.0), rnd(R, seed=2).abs() * 50 + 0.01
    fb, ga, be, g = rnd(C, seed=3), rnd(C, seed=4) + 1.5, rnd(C, seed=5), rnd(R, C, seed=6)
    ref = [t.double().requires_grad_() for t in (lin, msum, fb, ga, be)]
    obj = (ref[0] + ref[1].unsqueeze(-1) * ref[2]) / (ref[1].unsqueeze(-1) + 1e-5)
    yr = F.layer_norm(obj, (C,), ref[3], ref[4], 1e-5)
    (yr * g.double()).sum().backward()
    dev = [t.to(DEV).requires_grad_() for t in (lin, msum, fb, ga, be)]
    y = fn.PooledHeadFn.apply(*dev, 1e-5)
    close(y, yr, rtol=1e-5, atol=1e-5, msg='y')
    (y * g.to(DEV)).sum().backward()
    for a, b, n in zip(dev, ref, ('dlin', 'dmsum', 'dfbias', 'dgamma', 'dbeta')):
        close(a.grad, b.grad, rtol=1e-4, atol=1e-5, msg=n)


@pytest.mark.parametrize('kind,N,Cin,Cout,S', [('conv3x3', 4, 128, 128, 8), ('conv3x3', 2, 64, 64, 32),
                                                 ('conv3x3', 2, 64, 64, 64), ('deconv', 8, 64, 64, 8),
                                                 ('deconv', 3, 66, 64, 4)])
def test_groupnorm_sums_splitk_partials_bitwise(kind, N, Cin, Cout, S, monkeypatch):
    """gx_*_fwd_parts + gx_gn_relu_fwd_parts (GroupNorm sums the split-K slabs and adds the bias while reading) must
    give the same bits as conv (+ reduce kernel) followed by gx_gn_relu_fwd."""
    x = rnd(N, Cin, S, S, seed=1).to(DEV)
    gamma, beta = (rnd(Cout, seed=4) + 1.5).to(DEV), rnd(Cout, seed=5).to(DEV)
    outs = []
    for fused in (False, True):
        monkeypatch.setattr(hip, 'FUSE_SPLITK_INTO_GN', fused)
        if kind == 'conv3x3':
            w = (rnd(Cout, Cin, 3, 3, seed=2) * 0.05).to(DEV)
            d = torch.empty(N, Cout, S, S, device=DEV)
            y, mean, rstd = hip.conv3x3_gn_relu_fwd(x, w, gamma, beta, 8, 1e-5, (d, 0, 0))
        else:
            w = (rnd(Cin, Cout, 5, 5, seed=2) * 0.05).to(DEV)
            b = rnd(Cout, seed=3).to(DEV)
            d = torch.empty(N, Cout, 2 * S, 2 * S, device=DEV)
            y, mean, rstd = hip.deconv5x5s2_gn_relu_fwd(x, w, b, gamma, beta, 8, 1e-5, (d, 0, 0))
        outs.append((y.clone(), mean.clone(), rstd.clone(), d.clone()))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


@pytest.mark.parametrize('N,C,Cout,S,groups', [(3, 64, 4, 64, 8), (2, 32, 4, 16, 8), (5, 64, 8, 32, 8), (2, 16, 3, 16, 8),
                                               (2, 64, 4, 128, 8), (3, 32, 7, 128, 8)])    # 128 x 128: the chunked (split) path
def test_conv1x1_on_unmaterialised_groupnorm(N, C, Cout, S, groups):
    """Last decoder stage (genesisv2_config.py:97-99): GroupNorm statistics only, the 1x1 conv normalises on load, its
    data gradient is folded into the norm backward -- against the three separate torch ops."""
    y = rnd(N, C, S, S, seed=31, scale=2.0) + 0.2
    gamma = 1 + 0.3 * rnd(C, seed=32)
    beta = 0.2 * rnd(C, seed=33)
    w = rnd(Cout, C, seed=34, scale=0.3)
    b = rnd(Cout, seed=35, scale=0.1)
    g = rnd(N, Cout, S, S, seed=36)
    yr, gr, br, wr, bbr = [t.clone().requires_grad_() for t in (y, gamma, beta, w, b)]
    ref = F.conv2d(F.relu(F.group_norm(yr, groups, gr, br, 1e-5)), wr.view(Cout, C, 1, 1), bbr)
    ref.backward(g)
    yd, gd, bd, wd, bbd, gg = [t.to(DEV) for t in (y, gamma, beta, w, b, g)]
    mean, rstd = hip.gn_relu_fwd(yd, gd, bd, groups, 1e-5, None)            # statistics only
    full_mean, full_rstd = hip.gn_relu_fwd(yd, gd, bd, groups, 1e-5, (torch.empty_like(yd), 0, 0))
    assert torch.equal(mean, full_mean) and torch.equal(rstd, full_rstd)
    out = hip.conv1x1_gn_fwd(yd, mean, rstd, gd, bd, groups, wd, bbd)
    close(out, ref, 1e-5, 1e-5, 'fwd')
    dw, db, _ = hip.conv1x1_gn_wgrad(yd, mean, rstd, gd, bd, groups, gg)
    close(dw, wr.grad, 1e-4, 1e-4, 'dw')
    close(db, bbr.grad, 1e-4, 1e-4, 'db')
    dy, dgamma, dbeta, dbias = hip.gn_relu_bwd_proj(yd, gd, bd, mean, rstd, groups, gg, wd, True)
    close(dy, yr.grad, 1e-4, 1e-5, 'dy')
    close(dgamma, gr.grad, 1e-4, 1e-4, 'dgamma')
    close(dbeta, br.grad, 1e-4, 1e-4, 'dbeta')
    close(dbias, yr.grad.sum((0, 2, 3)), 1e-4, 1e-4, 'dbias')
    fused = hip.conv1x1_gn_bwd_fused(yd, gd, bd, mean, rstd, groups, gg, wd, bbd, None, True)
    if (S * S * Cout * 4 <= 128 * 1024 and S * S >= 256) or S == 128:
        assert fused is not None                       # one pass over y: norm backward + conv weight gradient
    if fused is not None:
        dy2, (dgamma2, dbeta2, dbias2), (dw2, db2, _) = fused
        assert torch.equal(dy2, dy) and torch.equal(dgamma2, dgamma) and torch.equal(dbias2, dbias)
        close(dw2, wr.grad, 1e-4, 1e-4, 'dw (fused)')
        close(db2, bbr.grad, 1e-4, 1e-4, 'db (fused)')


@pytest.mark.parametrize('S', [32, 128])
def test_gated_conv1x1_on_unmaterialised_groupnorm(S):
    """SemiConv colour head on seg_head's never-written activation: gate * conv1x1(relu(gn(y))) + uv and all its
    gradients (gate included) against torch (S = 128: the chunked norm backward of the 128 x 128 configuration)."""
    N, C, Cout, groups = 3, 64, 8, 8
    y = rnd(N, C, S, S, seed=41, scale=2.0) - 0.1
    gamma = 1 + 0.3 * rnd(C, seed=42)
    beta = 0.2 * rnd(C, seed=43)
    w = rnd(Cout, C, seed=44, scale=0.3)
    b = rnd(Cout, seed=45, scale=0.1)
    gate = torch.tensor(0.37)
    uv = rnd(Cout, S, S, seed=46)
    g = rnd(N, Cout, S, S, seed=47)
    yr, gr, br, wr, bbr, gtr = [t.clone().requires_grad_() for t in (y, gamma, beta, w, b, gate)]
    ref = gtr * F.conv2d(F.relu(F.group_norm(yr, groups, gr, br, 1e-5)), wr.view(Cout, C, 1, 1), bbr) + uv
    ref.backward(g)
    yd, gd, bd, wd, bbd, gg, gtd, uvd = [t.to(DEV) for t in (y, gamma, beta, w, b, g, gate, uv)]
    mean, rstd = hip.gn_relu_fwd(yd, gd, bd, groups, 1e-5, None)
    out = hip.conv1x1_gn_fwd(yd, mean, rstd, gd, bd, groups, wd, bbd, gtd, uvd)
    close(out, ref, 1e-5, 1e-5, 'fwd')
    dw, db, dgate = hip.conv1x1_gn_wgrad(yd, mean, rstd, gd, bd, groups, gg, wd, bbd, gtd)
    close(dw, wr.grad, 1e-4, 1e-4, 'dw')
    close(db, bbr.grad, 1e-4, 1e-4, 'db')
    close(dgate, gtr.grad, 1e-4, 1e-4, 'dgate')
    dy, dgamma, dbeta, _ = hip.gn_relu_bwd_proj(yd, gd, bd, mean, rstd, groups, gg, wd, False, gate=gtd)
    close(dy, yr.grad, 1e-4, 1e-5, 'dy')
    close(dgamma, gr.grad, 1e-4, 1e-4, 'dgamma')
    close(dbeta, br.grad, 1e-4, 1e-4, 'dbeta')
    fused = hip.conv1x1_gn_bwd_fused(yd, gd, bd, mean, rstd, groups, gg, wd, bbd, gtd, False)
    assert fused is not None
    dy2, (dgamma2, dbeta2, _), (dw2, db2, dgate2) = fused
    assert torch.equal(dy2, dy) and torch.equal(dbeta2, dbeta)
    close(dw2, wr.grad, 1e-4, 1e-4, 'dw (fused)')
    close(db2, bbr.grad, 1e-4, 1e-4, 'db (fused)')
    close(dgate2, gtr.grad, 1e-4, 1e-4, 'dgate (fused)')


@pytest.mark.parametrize('N,Cin,Cout,Hin', [(9, 64, 64, 32), (3, 64, 64, 16), (2, 66, 64, 4), (5, 32, 16, 32)])
def test_deconv_epilogue_groupnorm_statistics(N, Cin, Cout, Hin):
    """Transposed conv whose epilogue also produces the GroupNorm statistics of its output (or, for shapes the
    epilogue path does not take, a statistics-only pass): same y, mean / rstd as F.group_norm's."""
    groups = 8
    x = rnd(N, Cin, Hin, Hin, seed=51)
    w = rnd(Cin, Cout, 5, 5, seed=52, scale=0.05)
    b = rnd(Cout, seed=53, scale=0.3)
    gamma, beta = 1 + 0.2 * rnd(Cout, seed=54), 0.1 * rnd(Cout, seed=55)
    ref = F.conv_transpose2d(x, w, b, 2, 2, 1)
    y, mean, rstd = hip.deconv5x5s2_gn_stats_fwd(x.to(DEV), w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), groups, 1e-5)
    close(y, ref, 2e-5, 2e-5, 'y')
    rg = ref.double().view(N, groups, -1)
    np.testing.assert_allclose(mean.cpu().double().numpy(), rg.mean(2).flatten().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(rstd.cpu().double().numpy(), (rg.var(2, unbiased=False) + 1e-5).rsqrt().flatten().numpy(),
                               rtol=2e-5)
    y2 = hip.deconv5x5s2_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
    assert torch.equal(y, y2)


@pytest.mark.parametrize('N,Cin,Cout,Hin', [(56, 64, 64, 32), (70, 64, 64, 32), (224, 64, 64, 16), (60, 32, 64, 32),
                                            (56, 64, 40, 32), (52, 48, 72, 32),
                                            (13, 64, 64, 64), (16, 32, 64, 64)])      # 64 x 64 base (cfg 5): four staging rounds, exact LDS planes
def test_transposed_conv_on_the_bf16_pipe_keeps_fp32_accuracy(N, Cin, Cout, Hin):
    """gx_kq_precision: chip-filling transposed-conv forward layers run on the bf16 matrix pipe (fp32 products from six
    bf16 piece products; input split once by the staging, weights by the pack) or on the fp32 pipe.  Both against
    conv_transpose2d (and its autograd: the data gradient) in fp64, with bias, with and without the GroupNorm statistics
    in the epilogue; shapes with a split
    tail (70 images), two / three / four 16-channel chunks, ragged and multi-tile output channels."""
    from genesis_amd import _lib
    groups = 8
    x = rnd(N, Cin, Hin, Hin, seed=71)
    w = rnd(Cin, Cout, 5, 5, seed=72, scale=0.05)
    b = rnd(Cout, seed=73, scale=0.3)
    gamma, beta = 1 + 0.2 * rnd(Cout, seed=74), 0.1 * rnd(Cout, seed=75)
    xr = x.double().requires_grad_()
    ref = F.conv_transpose2d(xr, w.double(), b.double(), 2, 2, 1)
    dy = rnd(N, Cout, 2 * Hin, 2 * Hin, seed=76)
    ref.backward(dy.double())
    ref = ref.detach()
    rg = ref.view(N, groups, -1)
    err, errm, errd = {}, {}, {}
    try:
        for mode in (0, 1, 2):       # 2: three fp16 piece products, per-tensor power-of-two scales (DESIGN.md finding 40)
            _lib.call('gx_kq_precision', mode)
            dx = hip.deconv5x5s2_dgrad(dy.to(DEV), w.to(DEV))
            errd[mode] = float((dx.double().cpu() - xr.grad).norm() / xr.grad.norm())
            y = hip.deconv5x5s2_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
            y2, mean, rstd = hip.deconv5x5s2_gn_stats_fwd(x.to(DEV), w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), groups, 1e-5)
            assert torch.equal(y, y2)
            err[mode] = float((y.double().cpu() - ref).norm() / ref.norm())
            errm[mode] = float((mean.double().cpu() - rg.mean(2).flatten()).abs().max())
            np.testing.assert_allclose(rstd.cpu().double().numpy(), (rg.var(2, unbiased=False) + 1e-5).rsqrt().flatten().numpy(), rtol=2e-5)
    finally:
        _lib.call('gx_kq_precision', -1)
    print('deconv fwd N=%d %d->%d @%d: relative L2 error fp32 pipe %.3e, bf16 x 6 %.3e, fp16 x 3 %.3e; |mean error| %.2e / %.2e / %.2e; '
          'data gradient %.3e / %.3e / %.3e' % (N, Cin, Cout, Hin, err[0], err[1], err[2], errm[0], errm[1], errm[2], errd[0], errd[1], errd[2]))
    for mode in (1, 2):
        assert err[mode] <= 1.5 * err[0] + 1e-7 and err[mode] < 2e-5, err
        assert errd[mode] <= 1.5 * errd[0] + 1e-7 and errd[mode] < 2e-5, errd
        assert errm[mode] <= 2e-6, errm


@pytest.mark.parametrize('N,Cin,Cout,H,W', [(2, 16, 16, 8, 16), (3, 24, 40, 32, 32), (2, 64, 64, 64, 64),
                                            (1, 128, 72, 16, 48), (2, 70, 130, 24, 32)])
def test_conv3x3_winograd(N, Cin, Cout, H, W):
    """Winograd F(2x2,3x3) forward and data gradient (gx_conv3x3_wino) against F.conv2d / its autograd: same fp32
    tolerance as the direct kernels (channel tails, several channel tiles, non-square grids)."""
    x = rnd(N, Cin, H, W, seed=61)
    w = rnd(Cout, Cin, 3, 3, seed=62, scale=0.1)
    dy = rnd(N, Cout, H, W, seed=63)
    xr = x.clone().requires_grad_()
    ref = F.conv2d(xr, w, None, 1, 1)
    ref.backward(dy)
    y = hip.conv3x3_wino(x.to(DEV), w.to(DEV), 0)
    close(y, ref, 2e-5, 2e-5, 'fwd')
    dx = hip.conv3x3_wino(dy.to(DEV), w.to(DEV), 1)
    close(dx, xr.grad, 2e-5, 2e-5, 'dgrad')


def test_conv3x3_entry_points_take_the_winograd_path_when_the_grid_fills_the_chip(monkeypatch):
    """gx_conv3x3_fwd / _dgrad dispatch: chip-filling layers run the Winograd kernel, the others the direct tap loop;
    both agree with each other to fp32 rounding."""
    from genesis_amd import profiling
    x = rnd(32, 64, 32, 32, seed=64).to(DEV)
    w = rnd(64, 64, 3, 3, seed=65, scale=0.1).to(DEV)
    profiling.enable(True)
    y = hip.conv3x3_fwd(x, w)
    dx = hip.conv3x3_dgrad(x, w)
    small = hip.conv3x3_fwd(x[:2], w)
    rows = {r['name']: r['launches'] for r in profiling.collect()}
    profiling.enable(False)
    assert rows.get('wino_conv_kernel') == 2 and rows.get('tapconv_kernel<0>') == 1, rows
    close(small, y[:2], 2e-5, 2e-5, 'winograd vs direct')
    assert torch.isfinite(dx).all()


def _pixel_coords(d):
    """modules/blocks.py:42-47."""
    g1, g2 = torch.meshgrid(torch.linspace(-1, 1, d), torch.linspace(-1, 1, d), indexing='ij')
    return torch.cat((g1.view(1, 1, d, d), g2.view(1, 1, d, d)), 1)


@pytest.mark.parametrize('N,D,d', [(5, 64, 4), (3, 16, 8), (2, 7, 5)])
def test_broadcast_concat(N, D, d):
    """BroadcastLayer + PixelCoords (modules/blocks.py:104-130) in one launch: bit-exact against expand + cat."""
    z, coords = rnd(N, D, seed=70), _pixel_coords(d)
    ref = torch.cat((z.view(N, D, 1, 1).expand(-1, -1, d, d), coords.expand(N, -1, -1, -1)), 1)
    got = hip.broadcast_concat(z.to(DEV), coords.to(DEV))
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize('N,L,Co,S,nl,act', [(3, 16, 32, 16, 4, 'relu'), (2, 16, 32, 64, 4, 'elu'), (5, 6, 8, 8, 2, None)])
def test_broadcast_decoder_first_layer_without_the_canvas(N, L, Co, S, nl, act):
    """modules/decoders.py:25-28 on the BroadcastLayer canvas (blocks.py:104-130): act(VALID conv3x3([z | g_1 | g_2]))
    from z, the weights' tap sums and the coordinate vectors -- forward on the valid conv's output positions (the canvas
    interior) and dz / dW / db against autograd through the materialised canvas."""
    d = S + 2 * nl
    z = rnd(N, L, seed=71).requires_grad_(True)
    w = rnd(Co, L + 2, 3, 3, seed=72, scale=0.3).requires_grad_(True)
    b = rnd(Co, seed=73, scale=0.2).requires_grad_(True)
    coords = _pixel_coords(d)
    canvas = torch.cat((z.view(N, L, 1, 1).expand(-1, -1, d, d), coords.expand(N, -1, -1, -1)), 1)
    pre = F.conv2d(canvas, w, b)                                  # valid: [N, Co, d-2, d-2]
    ref = pre if act is None else (F.relu(pre) if act == 'relu' else F.elu(pre))
    g = rnd(N, Co, d - 2, d - 2, seed=74)
    ref.backward(g)
    rowc, colc = coords[0, 0, :, 0].contiguous().to(DEV), coords[0, 1, 0, :].contiguous().to(DEV)
    zd, wd, bd = z.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV)
    y = hip.bcast_conv3x3_fwd(zd, wd, bd, rowc, colc, act)
    assert y.shape == (N, Co, d, d)
    close(y[:, :, 1:-1, 1:-1], ref, 2e-5, 2e-5, 'broadcast conv fwd (interior)')
    # the border ring of the incoming gradient must not matter: fill it with garbage
    gfull = torch.full((N, Co, d, d), 7.5)
    gfull[:, :, 1:-1, 1:-1] = g
    dz, dw, db = hip.bcast_conv3x3_bwd(y, gfull.to(DEV), zd, wd, rowc, colc, act)
    close(dz, z.grad, 1e-4, 1e-5, 'dz')
    close(dw, w.grad, 1e-4, 1e-5, 'dw')
    close(db, b.grad, 1e-4, 1e-5, 'db')


@pytest.mark.parametrize('T,B,S,last,with_s0', [(7, 3, 16, True, False), (1, 2, 32, False, True), (4, 2, 8, False, False)])
def test_stick_breaking_scan(T, B, S, last, with_s0):
    """modules/attention.py:42-48 / :118-124 (logsigmoid stick breaking), K steps in one launch, forward + backward."""
    from genesis_amd import functions as fn
    l = (rnd(T, B, 1, S, S, seed=80) * 4).requires_grad_(True)
    s0 = (-rnd(B, 1, S, S, seed=81).abs()).requires_grad_(True) if with_s0 else None
    s = torch.zeros(B, 1, S, S) if s0 is None else s0
    ms, ss = [], []
    for t in range(T):
        ms.append(s if (last and t == T - 1) else s + F.logsigmoid(l[t]))
        s = s + F.logsigmoid(-l[t])
        ss.append(s)
    ref_m, ref_s = torch.stack(ms), torch.stack(ss)
    gm, gs = rnd(T, B, 1, S, S, seed=82), rnd(T, B, 1, S, S, seed=83)
    ((ref_m * gm).sum() + (ref_s * gs).sum()).backward()
    ld = l.detach().to(DEV).requires_grad_(True)
    s0d = None if s0 is None else s0.detach().to(DEV).requires_grad_(True)
    got_m, got_s = fn.SBPScanFn.apply(ld, s0d, last)
    close(got_m, ref_m, 1e-5, 1e-5, 'log_m'); close(got_s, ref_s, 1e-5, 1e-5, 'log_s')
    ((got_m * gm.to(DEV)).sum() + (got_s * gs.to(DEV)).sum()).backward()
    close(ld.grad, l.grad, 1e-5, 1e-5, 'g_logits')
    if s0 is not None:
        close(s0d.grad, s0.grad, 1e-5, 1e-5, 'g_log_s0')


@pytest.mark.parametrize('K,B,S,through_r', [(4, 3, 16, False), (7, 2, 32, True), (2, 2, 8, True), (11, 2, 32, True),
                                             (20, 2, 32, True)])      # K > 16: the two-pass form (no limit on K_steps)
def test_categorical_mask_kl(K, B, S, through_r):
    """MONet.kl_m_loss (models/monet_config.py:157-170) as torch.distributions writes it, forward and both gradients
    (the reconstructed-mask side is the detach_mr_in_klm = False branch of genesisv2_config.py:172-176)."""
    from torch.distributions.categorical import Categorical
    from torch.distributions.kl import kl_divergence
    from genesis_amd import functions as fn
    lm = F.log_softmax(rnd(K, B, 1, S, S, seed=84) * 6, 0).requires_grad_(True)       # some masks fall below 1e-5
    lr = F.log_softmax(rnd(K, B, 1, S, S, seed=85) * 6, 0).requires_grad_(True)
    m = torch.max(torch.stack(list(lm), 4).exp(), torch.tensor(1e-5))
    r = torch.max(torch.stack(list(lr), 4).exp(), torch.tensor(1e-5))
    ref = kl_divergence(Categorical(m.view(-1, K)), Categorical(r.view(-1, K))).view(B, -1).sum(1)
    g = rnd(B, seed=86)
    (ref * g).sum().backward()
    lmd = lm.detach().to(DEV).requires_grad_(True)
    lrd = lr.detach().to(DEV).requires_grad_(through_r)
    got = fn.CategoricalKLFn.apply(lmd, lrd)
    close(got, ref, 2e-5, 1e-4, 'kl_m')
    (got * g.to(DEV)).sum().backward()
    close(lmd.grad, lm.grad, 1e-4, 1e-6, 'g_log_m')
    if through_r:
        close(lrd.grad, lr.grad, 1e-4, 1e-6, 'g_log_m_r')
    else:
        assert lrd.grad is None


def test_lstm_cell_with_fed_back_input():
    """LatentSBP's recurrent core (modules/attention.py:103-110): the step's input depends on the previous output, so
    each step is its own autograd node on the HIP LSTM-step kernel; against nn.LSTM stepped one token at a time."""
    from genesis_amd import functions as fn
    torch.manual_seed(4)
    B, Din, H, T = 5, 48, 32, 4
    lstm = torch.nn.LSTM(Din + H, H)
    x0 = rnd(B, Din, seed=87)
    proj = rnd(H, H, seed=88, scale=0.5)

    def run(cell, dev):
        inp_fixed = x0.to(dev)
        fb = torch.zeros(B, H, device=dev)
        state, outs = None, []
        for _ in range(T):
            h, state = cell(torch.cat([inp_fixed, fb], 1), state)
            fb = torch.tanh(h @ proj.to(dev))            # the next input depends on this output
            outs.append(h)
        return torch.stack(outs)

    def ref_cell(inp, state):
        out, state = lstm(inp.unsqueeze(0), state)
        return out[0], state
    ref = run(ref_cell, 'cpu')
    ref.square().sum().backward()
    ref_grads = [p.grad.clone() for p in lstm.parameters()]
    dl = torch.nn.LSTM(Din + H, H).to(DEV)
    dl.load_state_dict(lstm.state_dict())

    def hip_cell(inp, state):
        hs, cs = (None, None) if state is None else state
        h, c = fn.LSTMCellFn.apply(inp, hs, cs, dl.weight_ih_l0, dl.weight_hh_l0, dl.bias_ih_l0, dl.bias_hh_l0)
        return h, (h, c)
    got = run(hip_cell, DEV)
    close(got, ref, 1e-5, 1e-5, 'lstm cell outputs')
    got.square().sum().backward()
    for p, r in zip(dl.parameters(), ref_grads):
        close(p.grad, r, 1e-4, 1e-5, 'lstm cell parameter grad')


def test_streamk_weight_gradients_many_layers():
    """The deferred path launches EVERY queued conv3x3 / transposed-conv weight gradient in one stream-K grid
    (gx_wgq.hip): queue more (layer, 64x64 channel block) jobs than one launch's job table holds (36), of all three tile
    widths, both tap classes and ragged channel counts, flush, and compare every layer with the same entry point called
    on its own (other split points: fp32 round-off) and with PyTorch; flushing the same queue twice gives the same bits."""
    from genesis_amd import _lib
    layers = []   # (kind, N, Cin, Cout, H)
    for i, (N, Cin, Cout, H) in enumerate([(4, 64, 64, 64), (3, 128, 128, 16), (5, 128, 64, 32), (7, 64, 128, 8),
                                           (2, 96, 40, 32), (6, 128, 128, 8), (3, 128, 128, 16), (2, 64, 64, 64),
                                           (9, 128, 128, 8), (2, 128, 128, 32)]):
        layers.append(('c3', N, Cin, Cout, H))
    for (N, Cin, Cout, H) in [(6, 64, 64, 32), (10, 64, 64, 16), (12, 64, 32, 8), (3, 128, 64, 16)]:
        layers.append(('dt', N, Cin, Cout, H))
    data = []
    for i, (kind, N, Cin, Cout, H) in enumerate(layers):
        x = rnd(N, Cin, H, H, seed=10 + i).to(DEV)
        if kind == 'c3':
            dy = rnd(N, Cout, H, H, seed=50 + i).to(DEV)
            alone = hip.conv3x3_wgrad(x, dy)
            ref = torch.nn.grad.conv2d_weight(x.cpu().double(), (Cout, Cin, 3, 3), dy.cpu().double(), padding=1)
        else:
            dy = rnd(N, Cout, 2 * H, 2 * H, seed=50 + i).to(DEV)
            alone = hip.deconv5x5s2_wgrad(x, dy)
            xr = x.cpu().double()
            wr = torch.zeros(Cin, Cout, 5, 5, dtype=torch.float64, requires_grad=True)
            F.conv_transpose2d(xr, wr, None, 2, 2, 1).backward(dy.cpu().double())
            ref = wr.grad
        data.append((kind, x, dy, alone, ref))

    def queued():
        outs = []
        st = hip.defer_state()
        st.on = True
        try:
            for kind, x, dy, alone, _ in data:
                out = torch.zeros_like(alone)        # the batched reduce accumulates into a zeroed gradient
                (hip.conv3x3_wgrad if kind == 'c3' else hip.deconv5x5s2_wgrad)(x, dy, out=out)
                outs.append(out)
            assert _lib.query('gx_defer_pending') == 18      # 10 + 2 x 4 layer jobs = 38 channel-block jobs (> 36)
            hip.defer_flush()
        finally:
            st.on = False
        torch.cuda.synchronize()
        return outs

    a, b = queued(), queued()
    _lib.call('gx_wgq_policy', 2)          # the same kernels, one launch per (tap class, tile width)
    try:
        grouped = queued()
    finally:
        _lib.call('gx_wgq_policy', 1)
    for (kind, x, dy, alone, ref), o, o2, og in zip(data, a, b, grouped):
        assert torch.equal(o, o2)
        close(og, ref, rtol=2e-5, atol=1e-6 * float(ref.abs().max()), msg='grouped %s %s' % (kind, tuple(x.shape)))
        close(o, ref, rtol=2e-5, atol=1e-6 * float(ref.abs().max()), msg='stream-K %s %s' % (kind, tuple(x.shape)))
        close(o, alone, rtol=2e-5, atol=1e-6 * float(ref.abs().max()), msg='stream-K vs alone %s' % (tuple(x.shape),))


@pytest.mark.parametrize('kind,N,Cin,Cout,S', [('conv3x3', 5, 64, 64, 64), ('conv3x3', 1, 128, 64, 64), ('conv3x3', 7, 96, 40, 32),
                                                ('conv3x3', 33, 64, 128, 32), ('deconv', 3, 64, 64, 32), ('deconv', 29, 40, 72, 32),
                                                ('deconv', 1, 64, 64, 32),
                                                # rows of 16 pixels: two image rows per tile (a zero piece at the seam)
                                                ('conv3x3', 9, 64, 200, 16), ('conv3x3', 1, 128, 64, 16), ('deconv', 13, 64, 64, 16),
                                                ('deconv', 1, 40, 72, 16), ('conv3x3', 3, 64, 64, 8),
                                                # rows too wide for the ring (the 128 x 128 model): two column strips per row, the
                                                # dy values next to a strip in the seam bytes of its A rows
                                                ('conv3x3', 3, 64, 64, 128), ('conv3x3', 1, 128, 64, 128), ('conv3x3', 2, 40, 72, 128),
                                                ('deconv', 3, 64, 64, 64), ('deconv', 1, 40, 72, 64),
                                                # channel blocks with an empty half: the waves of a pair share 32 channels and split
                                                # the tile's k-groups (A half, B half, both; a layer's LAST block only)
                                                ('conv3x3', 5, 64, 32, 64), ('conv3x3', 5, 32, 64, 64), ('conv3x3', 3, 4, 32, 64),
                                                ('conv3x3', 9, 32, 32, 32), ('conv3x3', 2, 96, 160, 64), ('deconv', 3, 32, 64, 32),
                                                ('deconv', 5, 64, 32, 32), ('deconv', 1, 24, 96, 32)])
def test_row_ring_weight_gradients(kind, N, Cin, Cout, S):
    """gx_wgq_ring: the row-ring tiles of the bf16-pipe weight gradients (one full-width base row per tile, x rows in a
    rolling four-slot LDS ring, operands split into bf16 planes once, column shifts as funnel shifts of the dy operand)
    against the 64-pixel LDS-DMA tiles and against autograd in fp64 -- odd image counts (stream-K segments start and end
    anywhere inside an image), a single image, ragged channel blocks; same accuracy bar as the other tiles; two runs are
    bit-identical."""
    from genesis_amd import hip_ops as hip, _lib
    if kind == 'conv3x3':
        x, dy = rnd(N, Cin, S, S, seed=1), rnd(N, Cout, S, S, seed=2)
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)
        run = lambda: hip.conv3x3_wgrad(x.to(DEV), dy.to(DEV))  # noqa: E731
    else:
        x, dy = rnd(N, Cin, S, S, seed=1), rnd(N, Cout, 2 * S, 2 * S, seed=2)
        w = torch.zeros(Cin, Cout, 5, 5, dtype=torch.float64, requires_grad=True)
        F.conv_transpose2d(x.double(), w, None, 2, 2, 1).backward(dy.double())
        ref = w.grad
        run = lambda: hip.deconv5x5s2_wgrad(x.to(DEV), dy.to(DEV))  # noqa: E731
    err = {}
    try:
        for mode in (0, 1):
            _lib.call('gx_wgq_ring', mode)
            got = run()
            if mode == 1:
                for _ in range(8):         # (a race between a segment's LDS fills would show as differing bits)
                    assert torch.equal(got, run())
            err[mode] = float((got.double().cpu() - ref).norm() / ref.norm())
    finally:
        _lib.call('gx_wgq_ring', 1)
    print('%s N=%d %d->%d @%d: relative L2 error 64-pixel tiles %.3e, row-ring tiles %.3e' % (kind, N, Cin, Cout, S, err[0], err[1]))
    assert err[1] <= 1.5 * err[0] + 1e-7 and err[1] < 1e-4, err


@pytest.mark.parametrize('N,CA,CB,S', [(3, 64, 32, 64), (7, 64, 64, 32), (1, 128, 64, 32), (5, 40, 24, 64), (33, 64, 32, 32),
                                        (40, 128, 64, 16), (9, 64, 128, 16), (5, 24, 40, 8), (3, 16, 16, 12), (2, 70, 33, 4)])
def test_conv5x5_stride1_weight_gradient_on_the_row_ring_tiles(N, CA, CB, S):
    """gx_conv5x5_wgrad (the gated 5x5 stride-1 (de)convs of third_party/sylvester): dw[a][b][kh][kw] = sum a * shifted b
    on the bf16-pipe row-ring tiles (kernel rows 0-2 and 3-4 as two jobs of one stream-K launch) -- rows of < 32 pixels: on the
    lean fp32-pipe kernel with a 2-pixel halo (kernel rows 0-1, 2-3, 4 as three launches) -- against autograd in
    fp64, in both roles: Conv2d (a = dy, b = x) and stride-1 ConvTranspose2d (a = x, b = dy); bit-identical twice."""
    from genesis_amd import hip_ops as hip
    os.environ['GENESIS_C5_FAST'] = '2'           # every size, not only the layers it pays for
    try:
        _conv5x5_wgrad_case(hip, N, CA, CB, S)
    finally:
        del os.environ['GENESIS_C5_FAST']


def _conv5x5_wgrad_case(hip, N, CA, CB, S):
    assert hip.conv5x5_wgrad_supported(N, CA, CB, S, S)
    a, b = rnd(N, CA, S, S, seed=1), rnd(N, CB, S, S, seed=2)
    # Conv2d: x = b [N, Cin = CB], dy = a [N, Cout = CA], weight [CA, CB, 5, 5]
    ref = torch.nn.grad.conv2d_weight(b.double(), (CA, CB, 5, 5), a.double(), padding=2)
    got = hip.conv5x5_wgrad(a.to(DEV), b.to(DEV))
    assert torch.equal(got, hip.conv5x5_wgrad(a.to(DEV), b.to(DEV)))
    err = float((got.double().cpu() - ref).norm() / ref.norm())
    # stride-1 ConvTranspose2d: x = a [N, Cin = CA], dy = b [N, Cout = CB], weight [CA, CB, 5, 5]
    wt = torch.zeros(CA, CB, 5, 5, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(a.double(), wt, None, 1, 2).backward(b.double())
    err_t = float((got.double().cpu() - wt.grad).norm() / wt.grad.norm())
    print('conv5x5 wgrad N=%d %dx%d @%d: relative L2 error %.3e (conv), %.3e (transposed)' % (N, CA, CB, S, err, err_t))
    assert err < 2e-6 and err_t < 2e-6, (err, err_t)


@pytest.mark.parametrize('N,K,M,S', [(3, 32, 64, 64), (5, 64, 128, 32), (2, 3, 64, 64), (4, 40, 24, 16), (33, 64, 64, 16), (1, 32, 64, 8),
                                     # chip-filling grids with K % 16 == 0: the bf16-pipe kernel (gx_kq.hip Q_C5H, six piece products)
                                     (16, 32, 64, 64), (52, 64, 128, 32), (200, 64, 64, 16), (16, 32, 72, 64), (13, 48, 64, 64)])
def test_conv5x5_stride1_on_the_tapconv_kernel(N, K, M, S):
    """gx_conv5x5s1 (tap-conv MFMA kernel, 25-tap table, 2-pixel halo): both weight roles against fp64 torch --
    flip 0 = F.conv2d(x, w [M,K,5,5], padding=2); flip 1 = F.conv_transpose2d(x, w [K,M,5,5], stride 1, padding 2)."""
    from genesis_amd import hip_ops as hip
    assert hip.conv5x5s1_supported(N, K, M, S, S)
    x = rnd(N, K, S, S, seed=1)
    w0 = rnd(M, K, 5, 5, seed=2, scale=0.1)
    w1 = rnd(K, M, 5, 5, seed=3, scale=0.1)
    close(hip.conv5x5s1(x.to(DEV), w0.to(DEV), M, False), F.conv2d(x.double(), w0.double(), None, 1, 2), rtol=2e-5, atol=2e-5,
          msg='cross-correlation')
    close(hip.conv5x5s1(x.to(DEV), w1.to(DEV), M, True), F.conv_transpose2d(x.double(), w1.double(), None, 1, 2), rtol=2e-5,
          atol=2e-5, msg='convolution (transposed-conv forward)')


def _stress(kind, *shape, seed=0):
    """Operand distributions the six-bf16-piece products are sensitive to (uniform [-1, 1] is the easy case):
    'offset'  |mean| >> std, as post-GroupNorm+ReLU activations or a head's dy (mean / std = 50);
    'relu'    half exact zeros, the rest with mean / std ~ 20;
    'mixed'   per-channel magnitudes log-uniform over 1e-4 .. 1e3 inside one reduction."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(*shape, generator=g)
    if kind == 'offset':
        return 1.0 + 0.02 * u
    if kind == 'relu':
        return torch.relu(torch.randn(*shape, generator=g)).sign() * (2.0 + 0.1 * u)
    if kind == 'mixed':
        mag = 10.0 ** (torch.rand(1, shape[1], 1, 1, generator=g) * 7.0 - 4.0)
        return u * mag
    raise ValueError(kind)


@pytest.mark.parametrize('dist', ['offset', 'relu', 'mixed'])
@pytest.mark.parametrize('kind,N,Cin,Cout,S', [('conv3x3', 32, 64, 64, 64), ('conv3x3', 8, 128, 64, 32), ('deconv', 28, 64, 64, 32),
                                                ('conv5x5', 6, 32, 64, 64)])
def test_bf16_pipe_weight_gradients_on_hard_operands(dist, kind, N, Cin, Cout, S):
    """The accuracy claim of the bf16-pipe kernels (fp32 products from six bf16 piece products, fp32 accumulate) on
    operands that stress it: large common offsets (cancellation in the accumulate is the fp32 pipe's problem too -- the bar
    stays 'no worse than 1.5 x the fp32 pipe'), exact zeros, seven decades of magnitude inside one reduction, and the
    longest reduction of the workload (conv3x3 64 -> 64 at 64 x 64, batch 32: 131 k products per weight)."""
    from genesis_amd import hip_ops as hip, _lib
    x = _stress(dist, N, Cin, S, S, seed=1)
    if kind == 'deconv':
        dy = _stress(dist, N, Cout, 2 * S, 2 * S, seed=2)
        w = torch.zeros(Cin, Cout, 5, 5, dtype=torch.float64, requires_grad=True)
        F.conv_transpose2d(x.double(), w, None, 2, 2, 1).backward(dy.double())
        ref = w.grad
        run = lambda: hip.deconv5x5s2_wgrad(x.to(DEV), dy.to(DEV))  # noqa: E731
    elif kind == 'conv3x3':
        dy = _stress(dist, N, Cout, S, S, seed=2)
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)
        run = lambda: hip.conv3x3_wgrad(x.to(DEV), dy.to(DEV))  # noqa: E731
    else:
        dy = _stress(dist, N, Cout, S, S, seed=2)
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 5, 5), dy.double(), padding=2)
        run = lambda: hip.conv5x5_wgrad(dy.to(DEV), x.to(DEV))  # noqa: E731
    err = {}
    try:
        for mode in ((1,) if kind == 'conv5x5' else (0, 1)):       # (the 5 x 5 class exists on the bf16 pipe only)
            _lib.call('gx_wgq_precision', mode)
            err[mode] = float((run().double().cpu() - ref).norm() / ref.norm())
    finally:
        _lib.call('gx_wgq_precision', -1)
    # what plain fp32 arithmetic (torch on the host, fp32) makes of the same sums
    if kind == 'conv3x3':
        cpu32 = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1)
    elif kind == 'conv5x5':
        cpu32 = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 5, 5), dy, padding=2)
    else:
        w32 = torch.zeros(Cin, Cout, 5, 5, requires_grad=True)
        F.conv_transpose2d(x, w32, None, 2, 2, 1).backward(dy)
        cpu32 = w32.grad
    e32 = float((cpu32.double() - ref).norm() / ref.norm())
    print('%s %s N=%d %d->%d @%d: relative L2 error vs fp64: fp32 pipe %s, bf16 pipe %.3e, torch CPU fp32 %.3e'
          % (kind, dist, N, Cin, Cout, S, ('%.3e' % err[0]) if 0 in err else 'n/a', err[1], e32))
    bar = 1.5 * (err[0] if 0 in err else e32) + 1e-7
    assert err[1] <= max(bar, 1.5 * e32 + 1e-7), (err, e32)


@pytest.mark.parametrize('dist', ['offset', 'relu', 'mixed'])
def test_bf16_pipe_transposed_conv_on_hard_operands(dist):
    """The same stress operands through the bf16-pipe transposed-conv forward and data gradient (gx_kq.hip)."""
    from genesis_amd import hip_ops as hip, _lib
    N, Cin, Cout, Hin = 56, 64, 64, 32
    x = _stress(dist, N, Cin, Hin, Hin, seed=3)
    dy = _stress(dist, N, Cout, 2 * Hin, 2 * Hin, seed=4)
    w = rnd(Cin, Cout, 5, 5, seed=5, scale=0.05)
    xr = x.double().requires_grad_()
    ref = F.conv_transpose2d(xr, w.double(), None, 2, 2, 1)
    ref.backward(dy.double())
    ref = ref.detach()
    err, errd = {}, {}
    try:
        for mode in (0, 1, 2):
            _lib.call('gx_kq_precision', mode)
            y = hip.deconv5x5s2_fwd(x.to(DEV), w.to(DEV), None)
            dx = hip.deconv5x5s2_dgrad(dy.to(DEV), w.to(DEV))
            err[mode] = float((y.double().cpu() - ref).norm() / ref.norm())
            errd[mode] = float((dx.double().cpu() - xr.grad).norm() / xr.grad.norm())
    finally:
        _lib.call('gx_kq_precision', -1)
    print('deconv %s: forward fp32 pipe %.3e, bf16 x 6 %.3e, fp16 x 3 %.3e; data gradient %.3e / %.3e / %.3e' % (dist, err[0], err[1], err[2], errd[0], errd[1], errd[2]))
    for mode in (1, 2):
        assert err[mode] <= 1.5 * err[0] + 1e-7 and errd[mode] <= 1.5 * errd[0] + 1e-7, (err, errd)


@pytest.mark.parametrize('N,Cin,Cout,S,cin_n', [(5, 4, 32, 64, 1), (5, 4, 32, 64, None), (3, 32, 32, 32, None), (4, 32, 64, 16, None),
                                                 (7, 64, 64, 8, None), (2, 6, 10, 12, 3)])
def test_conv3x3_stride2_dgrad_on_the_vector_alus(N, Cin, Cout, S, cin_n):
    """gx_conv3x3s2_dgrad_small (ComponentVAE encoder, modules/encoders.py:31-34): the stride-2 conv3x3 data gradient as 2 x 2
    parity blocks on the vector ALUs against autograd in fp64; cin_n limits the computed channels (the rest stays zero)."""
    from genesis_amd import hip_ops as hip
    x = rnd(N, Cin, S, S, seed=1).double().requires_grad_()
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.2)
    dy = rnd(N, Cout, S // 2, S // 2, seed=3)
    F.conv2d(x, w.double(), None, 2, 1).backward(dy.double())
    dx = hip.conv3x3s2_dgrad_small(dy.to(DEV), w.to(DEV), S, S, cin_n)
    n = Cin if cin_n is None else cin_n
    close(dx[:, :n], x.grad[:, :n], rtol=1e-5, atol=1e-5, msg='dx')
    assert float(dx[:, n:].abs().sum()) == 0.0


@pytest.mark.parametrize('K,B,S,Cout,act', [(7, 32, 64, 32, 'relu'), (3, 2, 16, 8, 'elu'), (1, 5, 8, 4, 'relu')])
def test_component_vae_first_layer_on_its_stacked_input(K, B, S, Cout, act):
    """MaskImageConvActFn (modules/component_vae.py:59-66 + modules/encoders.py:31-34): gx_mask_image_stack writes the slot-major
    [log_m_k | x] batch bit-exactly (it is a copy), the layer equals act(conv2d(cat(log_m_k, x), stride 2, pad 1)), and the
    compact mask gradient (gx_conv3x3s2_dgrad_small_ex) / the parameter gradients equal autograd's through cat + repeat."""
    from genesis_amd import functions as fn
    log_m = (rnd(K, B, 1, S, S, seed=1) - 1.5)
    x = rnd(B, 3, S, S, seed=2) * 0.5 + 0.5
    w = rnd(Cout, 4, 3, 3, seed=3, scale=0.3)
    b = rnd(Cout, seed=4, scale=0.1)
    g = rnd(K * B, Cout, S // 2, S // 2, seed=5)
    want = torch.cat((log_m.flatten(0, 1), x.repeat(K, 1, 1, 1)), 1)
    assert torch.equal(hip.mask_image_stack(log_m.to(DEV), x.to(DEV)).cpu(), want)
    lm_r, w_r, b_r = (t.double().requires_grad_() for t in (log_m, w, b))
    inp = torch.cat((lm_r.flatten(0, 1), x.double().repeat(K, 1, 1, 1)), 1)
    y_r = getattr(F, act)(F.conv2d(inp, w_r, b_r, 2, 1))
    y_r.backward(g.double())
    lm_d = log_m.to(DEV).requires_grad_()
    w_d, b_d = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = fn.MaskImageConvActFn.apply(lm_d, x.to(DEV), w_d, b_d, act)
    close(y, y_r, rtol=1e-5, atol=1e-5, msg='y')
    y.backward(g.to(DEV))
    assert lm_d.grad.shape == log_m.shape and lm_d.grad.is_contiguous()
    close(lm_d.grad, lm_r.grad, rtol=1e-5, atol=1e-5, msg='d log_m')
    close(w_d.grad, w_r.grad, rtol=1e-5, atol=2e-6 * float(w_r.grad.abs().max()), msg='dw')
    close(b_d.grad, b_r.grad, rtol=1e-5, atol=2e-6 * float(b_r.grad.abs().max()), msg='db')


@pytest.mark.parametrize('N,Cin,Cout,S', [(224, 4, 32, 64), (37, 32, 32, 32), (40, 32, 64, 16), (70, 64, 64, 8), (2, 6, 10, 12),
                                          (1, 3, 5, 2)])
def test_conv3x3_stride2_wgrad_on_the_vector_alus(N, Cin, Cout, S):
    """gx_conv3x3s2_wgrad_small (ComponentVAE encoder, modules/encoders.py:31-34): lanes = output pixels, 9 taps x 8 output
    channels per thread, fixed-order split reduction; against autograd in fp64, and bit-reproducible run to run."""
    from genesis_amd import hip_ops as hip
    x = rnd(N, Cin, S, S, seed=1)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    dy = rnd(N, Cout, S // 2, S // 2, seed=3)
    F.conv2d(x.double(), w, None, 2, 1).backward(dy.double())
    dw = hip.conv3x3s2_wgrad_small(x.to(DEV), dy.to(DEV))
    scale = float(w.grad.abs().max())
    close(dw, w.grad, rtol=1e-5, atol=2e-6 * scale, msg='dw')
    assert torch.equal(dw, hip.conv3x3s2_wgrad_small(x.to(DEV), dy.to(DEV)))


@pytest.mark.parametrize('K,B,D', [(7, 32, 16), (3, 5, 64), (1, 9, 7)])
def test_prior_logp_with_a_conditional_prior_on_every_slot(K, B, D):
    """PriorLogPFn(all_slots=True): Genesis' component prior p(z_c | z_m) = N(tanh(mlp[:L]), to_prior_sigma(mlp[L:])) for all K
    slots (models/genesis_config.py:229-247) -- log_q - log_p and the gradients for z, lin, log_q against torch in fp64."""
    from torch.distributions import Normal
    from genesis_amd import functions as fn
    z, lin, lq, w = rnd(K, B, D, seed=1, scale=2.0), rnd(K, B, 2 * D, seed=2, scale=3.0), rnd(K, B, seed=3), rnd(K, B, seed=4)
    zr, lr, qr = z.double().requires_grad_(), lin.double().requires_grad_(), lq.double().requires_grad_()
    mr, sr = lr.chunk(2, dim=2)
    kl_ref = qr - Normal(torch.tanh(mr), torch.sigmoid(sr + 4.0) + 1e-4).log_prob(zr).sum(2)
    (kl_ref * w.double()).sum().backward()
    zg, lg, qg = z.to(DEV).requires_grad_(), lin.to(DEV).requires_grad_(), lq.to(DEV).requires_grad_()
    kl = fn.PriorLogPFn.apply(zg, lg, qg, True)
    (kl * w.to(DEV)).sum().backward()
    close(kl, kl_ref, rtol=1e-5, atol=1e-4, msg='kl')
    close(zg.grad, zr.grad, rtol=1e-4, atol=1e-5, msg='dz')
    close(lg.grad, lr.grad, rtol=1e-4, atol=1e-5, msg='dlin')
    close(qg.grad, qr.grad, rtol=0, atol=0, msg='dlog_q')


@pytest.mark.parametrize('N,C,S', [(7, 32, 8), (3, 5, 4), (1, 64, 1)])
def test_bn_running_statistics_of_a_gated_unit(N, C, S):
    """gx_bn_running_update after GatedNormFn in 'bn' mode against two nn.BatchNorm2d in training mode (momentum 0.1,
    unbiased running variance, num_batches_tracked)."""
    import torch.nn as nn
    from genesis_amd import hip_ops as hip
    y, bias = rnd(N, 2 * C, S, S, seed=1, scale=2.0), rnd(2 * C, seed=2)
    ref_h, ref_g = nn.BatchNorm2d(C), nn.BatchNorm2d(C)
    dev_h, dev_g = nn.BatchNorm2d(C).to(DEV), nn.BatchNorm2d(C).to(DEV)
    for step in range(2):
        yy = y + step
        if N * S * S > 1:
            h, g = (yy + bias.view(1, -1, 1, 1)).chunk(2, 1)
            ref_h(h); ref_g(g)
        ones, zeros = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        out, stats = hip.gated_norm_fwd(yy.to(DEV), bias.to(DEV), 'bn', ones, zeros, ones, zeros)
        hip.bn_running_update(stats, C, N * S * S, dev_h, dev_g)
    if N * S * S > 1:
        for d, r in ((dev_h, ref_h), (dev_g, ref_g)):
            close(d.running_mean, r.running_mean, rtol=1e-5, atol=1e-6, msg='running_mean')
            close(d.running_var, r.running_var, rtol=1e-4, atol=1e-6, msg='running_var')
    assert int(dev_h.num_batches_tracked) == 2 and int(dev_g.num_batches_tracked) == 2


@pytest.mark.parametrize('N,C,S', [(32, 32, 16), (7, 64, 8), (3, 5, 4)])
def test_bn_running_statistics_inside_the_apply_kernel(N, C, S):
    """gx_gated_bn_running: the running-statistics update applied by the unit's own apply kernel (whose workgroups fold the
    statistics pass's partial sums themselves) leaves the bits of gated_norm_fwd + gx_bn_running_update -- output, {mean, rstd},
    running means / variances, num_batches_tracked -- and matches two nn.BatchNorm2d in training mode; an armed request is
    consumed by exactly one call.  layers.py:40-101."""
    import torch.nn as nn
    from genesis_amd import hip_ops as hip
    y, bias = rnd(N, 2 * C, S, S, seed=3, scale=2.0), rnd(2 * C, seed=4)
    ref_h, ref_g = nn.BatchNorm2d(C), nn.BatchNorm2d(C)
    a_h, a_g = nn.BatchNorm2d(C).to(DEV), nn.BatchNorm2d(C).to(DEV)        # separate launch
    f_h, f_g = nn.BatchNorm2d(C).to(DEV), nn.BatchNorm2d(C).to(DEV)        # inside the apply kernel
    gh, bh = rnd(C, seed=5).to(DEV) + 1.0, rnd(C, seed=6).to(DEV)
    gg, bg = rnd(C, seed=7).to(DEV) + 1.0, rnd(C, seed=8).to(DEV)
    for step in range(3):
        yy = y + 0.5 * step
        h, g = (yy + bias.view(1, -1, 1, 1)).chunk(2, 1)
        ref_h(h); ref_g(g)
        out_a, st_a = hip.gated_norm_fwd(yy.to(DEV), bias.to(DEV), 'bn', gh, bh, gg, bg)
        hip.bn_running_update(st_a, C, N * S * S, a_h, a_g)
        hip.gated_bn_running_arm(C, f_h, f_g)
        out_f, st_f = hip.gated_norm_fwd(yy.to(DEV), bias.to(DEV), 'bn', gh, bh, gg, bg)
        assert torch.equal(out_a, out_f) and torch.equal(st_a[:4 * C], st_f[:4 * C])
        out_n, _ = hip.gated_norm_fwd(yy.to(DEV), bias.to(DEV), 'bn', gh, bh, gg, bg)      # (not armed: no update)
        assert torch.equal(out_n, out_f)
        for a, f in ((a_h, f_h), (a_g, f_g)):
            assert torch.equal(a.running_mean, f.running_mean) and torch.equal(a.running_var, f.running_var)
            assert int(a.num_batches_tracked) == int(f.num_batches_tracked) == step + 1
    for d, r in ((f_h, ref_h), (f_g, ref_g)):
        close(d.running_mean, r.running_mean, rtol=1e-5, atol=1e-6, msg='running_mean')
        close(d.running_var, r.running_var, rtol=1e-4, atol=1e-6, msg='running_var')
    # the stand-alone finalize / parameter launches (GENESIS_GATED_FUSE=0) give the same bits, forward and backward
    import os
    dout = rnd(N, C, S, S, seed=9).to(DEV)
    res = {}
    for fuse in ('1', '0'):
        os.environ['GENESIS_GATED_FUSE'] = fuse
        try:
            u_h, u_g = nn.BatchNorm2d(C).to(DEV), nn.BatchNorm2d(C).to(DEV)
            hip.gated_bn_running_arm(C, u_h, u_g)
            out, st = hip.gated_norm_fwd(y.to(DEV), bias.to(DEV), 'bn', gh, bh, gg, bg)
            bwd = hip.gated_norm_bwd(y.to(DEV), bias.to(DEV), 'bn', gh, bh, gg, bg, st, dout)
            res[fuse] = (out, st[:4 * C].clone(), u_h.running_mean.clone(), u_g.running_var.clone()) + tuple(bwd)
        finally:
            os.environ.pop('GENESIS_GATED_FUSE', None)
    for a, b in zip(res['1'], res['0']):
        assert torch.equal(a, b)


@pytest.mark.parametrize('N,Cin,Cout,H,W', [(16, 32, 32, 72, 72), (5, 16, 24, 40, 24), (4, 32, 7, 16, 16), (9, 48, 32, 8, 64)])
def test_conv3x3_to_32_channels_on_the_bf16_pipe(N, Cin, Cout, H, W):
    """gx_kq.hip's Q_C3H (the BroadcastDecoder's canvas convs, modules/decoders.py:21-35): conv3x3 (+ bias + ELU) and its data
    gradient of layers with <= 32 output channels, every fp32 product from six bf16 piece products, on grids that are not
    powers of two.  Against fp64; the error must stay within 1.5 x the fp32-pipe kernels' (+ 1e-7) and the suite's 1e-4."""
    from genesis_amd import hip_ops as hip, _lib
    x, w, b = rnd(N, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(9 * Cin)), rnd(Cout, seed=3)
    dy = rnd(N, Cout, H, W, seed=4)
    xr = x.double().requires_grad_()
    yr = F.elu(F.conv2d(xr, w.double(), b.double(), padding=1))
    dxr, = torch.autograd.grad(F.conv2d(xr, w.double(), None, padding=1), xr, dy.double())
    err = {}
    _lib.call('gx_kq_policy', 2)                  # every eligible shape (the default asks for a chip-filling grid)
    try:
        for mode in (0, 1, 2):       # 2: three fp16 piece products (packs 40 / 41)
            _lib.call('gx_kq_precision', mode)
            y = hip.conv3x3_bias_act_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 'elu')
            dx = hip.conv3x3_dgrad(dy.to(DEV), w.to(DEV))
            err[mode] = (float((y.double().cpu() - yr).norm() / yr.norm()), float((dx.double().cpu() - dxr).norm() / dxr.norm()))
            if mode >= 1:
                close(y, yr, rtol=1e-5, atol=1e-5, msg='y')
                close(dx, dxr, rtol=1e-5, atol=1e-5, msg='dx')
    finally:
        _lib.call('gx_kq_precision', -1)
        _lib.call('gx_kq_policy', 1)
    print('conv3x3 -> %d channels %dx%d: forward fp32 pipe %.3e, bf16 x 6 %.3e, fp16 x 3 %.3e; data gradient %.3e / %.3e / %.3e'
          % (Cout, H, W, err[0][0], err[1][0], err[2][0], err[0][1], err[1][1], err[2][1]))
    for mode in (1, 2):
        assert err[mode][0] <= 1.5 * err[0][0] + 1e-7 and err[mode][1] <= 1.5 * err[0][1] + 1e-7, err


@pytest.mark.parametrize('N,Cin,Cout,H,W,act', [(16, 32, 32, 72, 72, 'relu'), (5, 24, 16, 40, 24, 'elu'), (4, 7, 32, 16, 16, 'relu'),
                                                 (9, 32, 48, 8, 64, 'elu')])
def test_conv3x3_data_gradient_with_the_previous_layers_activation_backward(N, Cin, Cout, H, W, act):
    """gx_conv3x3_dgrad_act (the BroadcastDecoder's Conv2d / ReLU chain, modules/decoders.py:25-32): the data gradient of a
    conv3x3 times act'(the producing layer's output) and that layer's bias gradient -- bit-identical to gx_conv3x3_dgrad
    followed by gx_bias_act_bwd (the activation's backward moved into the conv kernel's epilogue), bias gradient to fp32
    rounding of an fp64 sum; and against autograd in fp64."""
    from genesis_amd import hip_ops as hip, _lib
    w, dy = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(9 * Cin)), rnd(N, Cout, H, W, seed=4)
    pre = rnd(N, Cin, H, W, seed=5)
    xout = F.relu(pre) if act == 'relu' else F.elu(pre)
    pr = pre.double().requires_grad_()
    b = torch.zeros(Cin, dtype=torch.float64, requires_grad=True)
    a = pr + b.view(1, -1, 1, 1)
    F.conv2d(F.relu(a) if act == 'relu' else F.elu(a), w.double(), None, padding=1).backward(dy.double())
    _lib.call('gx_kq_policy', 2)                  # every eligible shape (the default asks for a chip-filling grid)
    try:
        assert hip.conv3x3_dgrad_act_supported(N, Cin, Cout, H, W)
        dxa, db = hip.conv3x3_dgrad_act(dy.to(DEV), w.to(DEV), xout.to(DEV), act)
        da = hip.conv3x3_dgrad(dy.to(DEV), w.to(DEV))
        dxa2, db2 = hip.bias_act_bwd(xout.to(DEV), da, act)
    finally:
        _lib.call('gx_kq_policy', 1)
    assert torch.equal(dxa, dxa2)
    close(db, db2, rtol=1e-6, atol=1e-6, msg='dbias vs the two-call form')
    close(dxa, pr.grad, rtol=1e-5, atol=1e-5, msg='dxa')
    close(db, b.grad, rtol=1e-5, atol=2e-6 * float(b.grad.abs().max()) + 1e-5, msg='dbias')
    assert not hip.conv3x3_dgrad_act_supported(N, 64, Cout, H, W)


@pytest.mark.parametrize('N,Cin,Cout,H,W,act', [(6, 32, 4, 72, 72, 'relu'), (3, 24, 3, 16, 16, 'elu'), (2, 64, 8, 8, 8, 'relu')])
def test_conv1x1_backward_with_the_previous_layers_activation_backward(N, Cin, Cout, H, W, act):
    """gx_conv1x1_bwd_act (the BroadcastDecoder's last ReLU + 1 x 1 conv, modules/decoders.py:31-32): dxa bit-identical to
    gx_conv1x1_bwd followed by gx_bias_act_bwd, dw / db identical, the layer's bias gradient to fp32 rounding of an fp64 sum."""
    from genesis_amd import hip_ops as hip
    pre = rnd(N, Cin, H, W, seed=5)
    x = (F.relu(pre) if act == 'relu' else F.elu(pre)).to(DEV)
    w, b, dy = rnd(Cout, Cin, seed=1).to(DEV), rnd(Cout, seed=2).to(DEV), rnd(N, Cout, H, W, seed=3).to(DEV)
    dxa, dw, db, dbx = hip.conv1x1_bwd_act(x, dy, w, b, act)
    dx, dw2, db2, _ = hip.conv1x1_bwd(x, dy, w, b)
    dxa2, dbx2 = hip.bias_act_bwd(x, dx, act)
    assert torch.equal(dxa, dxa2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    close(dbx, dbx2, rtol=1e-6, atol=1e-6, msg='dbias vs the two-call form')
    close(dbx, dxa.double().sum((0, 2, 3)), rtol=1e-6, atol=1e-6, msg='dbias')


@pytest.mark.parametrize('N,H,W', [(8, 72, 72), (4, 16, 12), (12, 8, 8), (20, 64, 64),
                                   # grids that are no power of two with W % 8 == 0: the column-strip kernel (gx_wstrip.hip) --
                                   # odd strip counts, a segment shorter than the rest, one row segment, the workload's shape
                                   (4, 24, 40), (8, 12, 24), (4, 72, 8), (224, 72, 72)])
def test_conv3x3_weight_gradient_with_four_images_per_tile(N, H, W):
    """gx_conv3x3_wgrad_quad (the BroadcastDecoder's 32 -> 32 canvas convs): four images per workgroup, one per wave, the four
    quadrant slabs summed by the reduce.  Against autograd in fp64 and bit-reproducible."""
    from genesis_amd import hip_ops as hip
    assert hip.conv3x3_wgrad_quad_supported(N, 32, H, W)
    assert not hip.conv3x3_wgrad_quad_supported(N + 1, 32, H, W) and not hip.conv3x3_wgrad_quad_supported(N, 16, H, W)
    x, dy = rnd(N, 32, H, W, seed=1), rnd(N, 32, H, W, seed=2)
    w = torch.zeros(32, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, padding=1).backward(dy.double())
    dw = hip.conv3x3_wgrad_quad(x.to(DEV), dy.to(DEV))
    close(dw, w.grad, rtol=1e-5, atol=2e-6 * float(w.grad.abs().max()), msg='dw')
    out = torch.full((32, 32, 3, 3), 7.0, device=DEV)
    hip.conv3x3_wgrad_quad(x.to(DEV), dy.to(DEV), out=out)
    assert torch.equal(out, dw)
    # ... with the layer's bias gradient (the channel sums of dy) from the same read of dy: gx_conv3x3_wgrad_quad_bias
    db = torch.full((32,), float('nan'), device=DEV)
    out2 = hip.conv3x3_wgrad_quad(x.to(DEV), dy.to(DEV), dbias_out=db)
    assert torch.equal(out2, dw)
    ref = dy.double().sum((0, 2, 3))
    close(db, ref, rtol=1e-5, atol=2e-6 * float(dy.abs().sum((0, 2, 3)).max()), msg='dbias')


def test_philox_noise_is_a_function_of_seed_step_and_position():
    """gx_philox_noise: the training step's rand_pixel (uniform [0, 1)) and eps (standard normal) from one counter-based launch --
    moments and tails of 2^22 draws, reproducible for one (seed, step), different for another step / seed, and the two tensors
    of one call are different streams."""
    from genesis_amd import hip_ops as hip
    step = torch.zeros((), dtype=torch.int64, device=DEV)
    n = 1 << 22
    u, z = hip.philox_noise((n,), (n + 3,), 1234, step)
    u2, z2 = hip.philox_noise((n,), (n + 3,), 1234, step)
    assert torch.equal(u, u2) and torch.equal(z, z2)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.double().mean()) - 0.5) < 1e-3 and abs(float(u.double().var()) - 1.0 / 12) < 1e-3
    hist = torch.histc(u, bins=64, min=0.0, max=1.0) / n
    assert float((hist - 1.0 / 64).abs().max()) < 5e-4
    zd = z.double()
    assert abs(float(zd.mean())) < 2e-3 and abs(float(zd.var()) - 1.0) < 3e-3
    assert abs(float((zd ** 3).mean())) < 1e-2 and abs(float((zd ** 4).mean()) - 3.0) < 3e-2
    assert 4.0 < float(zd.abs().max()) < 6.5 and bool(torch.isfinite(z).all())
    assert abs(float((zd.abs() > 1.959964).double().mean()) - 0.05) < 1e-3
    assert abs(float(torch.corrcoef(torch.stack((zd[:-1], zd[1:])))[0, 1])) < 2e-3       # neighbours (a Box-Muller pair) uncorrelated
    assert abs(float(torch.corrcoef(torch.stack((u.double(), zd[:n])))[0, 1])) < 2e-3     # the two tensors are different streams
    step.add_(1)
    u3, z3 = hip.philox_noise((n,), (n + 3,), 1234, step)
    u4, _ = hip.philox_noise((n,), None, 1235, step)
    assert not torch.equal(u3, u) and not torch.equal(z3, z) and not torch.equal(u4, u3)
    assert abs(float(torch.corrcoef(torch.stack((u.double(), u3.double())))[0, 1])) < 2e-3
    ue, ze = hip.philox_noise((5,), (2, 3), 7, None)           # ragged ends of the 4-wide calls, no step pointer
    assert ue.shape == (5,) and ze.shape == (2, 3) and bool(torch.isfinite(ze).all())


@pytest.mark.parametrize('N,Cin,Cout,S', [(4, 3, 64, 64), (3, 3, 32, 32), (4, 64, 64, 32), (2, 40, 24, 16),
                                           (32, 128, 64, 4), (5, 200, 72, 4)])     # 4 x 4: queued, launched together at the flush
def test_deferred_weight_gradients_of_a_shared_parameter_accumulate(N, Cin, Cout, S):
    """While gx_defer_enable(1) is in effect every conv3x3 weight-gradient call ADDS into its (zeroed) destination
    (include/genesis_hip.h): a weight used twice in one iteration -- a shared UNet -- ends with the SUM of both uses, also on
    the small-Cin kernel (Cin x 9 <= 32, e.g. a 3-channel input layer), whose reduce used to overwrite."""
    from genesis_amd import _lib
    xs = [rnd(N, Cin, S, S, seed=70 + i).to(DEV) for i in range(2)]
    dys = [rnd(N, Cout, S, S, seed=80 + i).to(DEV) for i in range(2)]
    ref = sum(torch.nn.grad.conv2d_weight(x.cpu().double(), (Cout, Cin, 3, 3), dy.cpu().double(), padding=1)
              for x, dy in zip(xs, dys))
    out = torch.zeros(Cout, Cin, 3, 3, device=DEV)
    st = hip.defer_state()
    st.on = True
    try:
        for x, dy in zip(xs, dys):
            hip.conv3x3_wgrad(x, dy, out=out)
        hip.defer_flush()
    finally:
        st.on = False
    close(out, ref, rtol=2e-5, atol=1e-6 * float(ref.abs().max()), msg='two deferred uses')
    # deferral off: a call overwrites (the plain entry-point contract)
    hip.conv3x3_wgrad(xs[0], dys[0], out=out)
    one = torch.nn.grad.conv2d_weight(xs[0].cpu().double(), (Cout, Cin, 3, 3), dys[0].cpu().double(), padding=1)
    close(out, one, rtol=2e-5, atol=1e-6 * float(one.abs().max()), msg='plain call overwrites')


@pytest.mark.parametrize('M,K,N', [(32, 64, 32768), (224, 64, 8192), (7, 16, 1024), (3, 8, 260), (33, 70, 4100)])
def test_matmul_with_the_weight_in_in_out_layout(M, K, N):
    """gx_matmul_nn_*: y = x w, dx = g w^T, dw = x^T g with w [K, N] -- the sylvester stacks' gated ConvTranspose2d 'fc'
    layer on a 1 x 1 input (VAE.py:27-33) -- against fp64; ragged M / K / N (N % 4 == 0) included."""
    x, w, g = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=1 / np.sqrt(K)), rnd(M, N, seed=3)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    y_ref = xr @ wr
    y_ref.backward(g.double())
    y = hip.matmul_nn_fwd(x.to(DEV), w.to(DEV))
    close(y, y_ref, 2e-6, 2e-6, 'fwd')
    dx, dw = hip.matmul_nn_bwd(x.to(DEV), w.to(DEV), g.to(DEV))
    close(dx, xr.grad, 5e-6, 5e-6, 'dx')
    close(dw, wr.grad, 5e-6, 5e-6, 'dw')
    # through autograd, as the gated decoders call it (4-d weight, flattened inside)
    from genesis_amd import functions as fn
    k = int(round((N // 4) ** 0.5))
    if 4 * k * k == N:
        xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).view(K, 4, k, k).clone().requires_grad_()
        out = fn.MatmulNNFn.apply(xd, wd)
        out.backward(g.to(DEV))
        close(out, y_ref, 2e-6, 2e-6, 'fn fwd'); close(xd.grad, xr.grad, 5e-6, 5e-6, 'fn dx')
        close(wd.grad.view(K, N), wr.grad, 5e-6, 5e-6, 'fn dw')


@pytest.mark.parametrize('K,B,S,C', [(7, 32, 64, 4), (3, 2, 16, 4), (1, 3, 8, 1), (5, 2, 12, 2)])
def test_log_softmax_over_the_slots(K, B, S, C):
    """gx_logsoftmax_k_fwd / _bwd against F.log_softmax over the K slots of the decoder output's last channel
    (MONet.get_mask_recon_stack, monet_config.py:137-139)."""
    from genesis_amd import functions as fn
    dec = rnd(K * B, C, S, S, seed=5, scale=4.0)
    g = rnd(K, B, 1, S, S, seed=6)
    dr = dec.double().requires_grad_()
    ref = F.log_softmax(dr[:, C - 1:].reshape(K, B, 1, S, S), dim=0)
    ref.backward(g.double())
    dd = dec.to(DEV).requires_grad_()
    out = fn.LogSoftmaxKFn.apply(dd, K)
    out.backward(g.to(DEV))
    close(out, ref, 2e-6, 2e-6, 'fwd')
    close(dd.grad, dr.grad, 2e-6, 2e-6, 'bwd')
    ref32 = F.log_softmax(dec[:, C - 1:].reshape(K, B, 1, S, S), dim=0)
    assert float((out.cpu() - ref32).abs().max()) <= 4e-7 * max(1.0, float(ref32.abs().max()))     # torch's evaluation order


def test_two_linear_heads_in_one_buffer():
    from genesis_amd import functions as fn
    h, w1, b1, w2, b2 = rnd(9, 256, seed=1), rnd(64, 256, seed=2, scale=0.06), rnd(64, seed=3), rnd(64, 256, seed=4, scale=0.06), rnd(64, seed=5)
    g = rnd(9, 128, seed=6)
    ps = [t.double().requires_grad_() for t in (h, w1, b1, w2, b2)]
    ref = torch.cat((F.linear(ps[0], ps[1], ps[2]), F.linear(ps[0], ps[3], ps[4])), 1)
    ref.backward(g.double())
    pd = [t.to(DEV).requires_grad_() for t in (h, w1, b1, w2, b2)]
    out = fn.TwoHeadLinearFn.apply(*pd)
    out.backward(g.to(DEV))
    close(out, ref, 2e-6, 2e-6, 'fwd')
    for a, b_, n in zip(pd, ps, ('dh', 'dw1', 'db1', 'dw2', 'db2')):
        close(a.grad, b_.grad, 5e-6, 5e-6, n)


@pytest.mark.parametrize('kind', ['wide', 'tiny', 'huge', 'zeros', 'one_large', 'one_huge'])
def test_fp16x3_transposed_conv_scales_follow_the_tensors(kind):
    """gx_kq_precision(2): every fp32 product of the chip-filling transposed convs from THREE fp16 piece products of x * 2^sx and
    w * 2^sw (hi + lo = 22 significant bits; hi*hi + hi*lo + lo*hi) with ONE power-of-two scale per tensor taken from its largest
    magnitude on the device (one small launch ahead of the conv; the weights' at pack time).  What a per-tensor scale has to
    survive: channels spread over 12 binary orders of magnitude with exact zeros (ReLU outputs), tensors that are tiny / huge as a
    whole (2^-30, 2^+30: far outside fp16's own range), an all-zero input (amax 0), one element 2^20 times the rest (everything else
    lands in fp16's subnormal range: absolute, not relative, accuracy -- still the CPU fp32 op's error against fp64 in norm)."""
    from genesis_amd import _lib
    N, Cin, Cout, H = 56, 64, 64, 32
    x, w, b = rnd(N, Cin, H, H, seed=4), rnd(Cin, Cout, 5, 5, seed=5, scale=0.05), rnd(Cout, seed=6)
    dy = rnd(N, Cout, 2 * H, 2 * H, seed=7)
    if kind == 'wide':
        x = torch.relu(x * torch.pow(2.0, -(torch.arange(Cin).float() % 13)).view(1, -1, 1, 1))
        dy = dy * torch.pow(2.0, -(torch.arange(Cout).float() % 13)).view(1, -1, 1, 1)
        w = w * torch.pow(2.0, -(torch.arange(Cout).float() % 7)).view(1, -1, 1, 1)
    elif kind == 'tiny':
        x, dy, w, b = x * 2.0 ** -30, dy * 2.0 ** -30, w * 2.0 ** -20, b * 2.0 ** -50
    elif kind == 'huge':
        x, dy, w, b = x * 2.0 ** 30, dy * 2.0 ** 30, w * 2.0 ** 20, b * 2.0 ** 50
    elif kind == 'zeros':
        x, dy = torch.zeros_like(x), torch.zeros_like(dy)
    elif kind == 'one_large':
        x[3, 5, 7, 9] = 2.0 ** 20
        dy[2, 4, 6, 8] = -2.0 ** 20
    elif kind == 'one_huge':
        # one element 2^30 (1e9) times the rest -- past the documented range of the per-TENSOR scale (include/genesis_hip.h,
        # gx_kq_precision): every other element is below max * 2^-28, i.e. its low piece is flushed and its high piece is an fp16
        # subnormal.  What must still hold is the documented ABSOLUTE accuracy 2^-40 max|tensor| per operand value.
        x[3, 5, 7, 9] = 2.0 ** 30
        dy[2, 4, 6, 8] = -2.0 ** 30
    out = {}
    for dt in (torch.float64, torch.float32):
        xr = x.to(dt).requires_grad_()
        y = F.conv_transpose2d(xr, w.to(dt), b.to(dt), 2, 2, 1)
        y.backward(dy.to(dt))
        out[dt] = (y.detach(), xr.grad)
    ref, c32 = out[torch.float64], out[torch.float32]
    try:
        _lib.call('gx_kq_precision', 2)
        y = hip.deconv5x5s2_fwd(x.to(DEV), w.to(DEV), b.to(DEV))
        dx = hip.deconv5x5s2_dgrad(dy.to(DEV), w.to(DEV))
    finally:
        _lib.call('gx_kq_precision', -1)
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(dx).all())

    def rel(a, r):
        return float((a.double().cpu() - r).norm() / (r.norm() + 1e-300))
    e = (rel(y, ref[0]), rel(dx, ref[1]))
    e32 = (rel(c32[0], ref[0]), rel(c32[1], ref[1]))
    print('fp16 x 3 transposed conv, %s operands: forward %.3e (CPU fp32 %.3e), data gradient %.3e (CPU fp32 %.3e)' % (kind, e[0], e32[0], e[1], e32[1]))
    if kind == 'zeros':
        assert torch.equal(dx, torch.zeros_like(dx)) and e[0] < 1e-6
    elif kind == 'one_huge':
        # per ELEMENT: the CPU fp32 op's own error + the representation floor 2^-38 max|operand| * sum |w| over a receptive field
        # (2^-40 per value, two pieces' worth of slack); in norm the outlier's footprint dominates and hides nothing
        wsum = float(w.abs().sum((0, 2, 3)).max()), float(w.abs().sum((1, 2, 3)).max())
        for got, r, c, amax, ws in ((y, ref[0], c32[0], float(x.abs().max()), wsum[0]), (dx, ref[1], c32[1], float(dy.abs().max()), wsum[1])):
            d = (got.double().cpu() - r).abs()
            bound = 4.0 * (c.double() - r).abs().max() + 2.0 ** -38 * amax * ws      # (measured: 2.8 x the CPU op's, inside the outlier's footprint)
            print('   one_huge: max abs error %.3e, bound %.3e (CPU fp32 max abs error %.3e)' % (float(d.max()), float(bound), float((c.double() - r).abs().max())))
            assert float(d.max()) <= float(bound), (float(d.max()), float(bound))
    else:
        assert e[0] <= 1.5 * e32[0] + 1e-7 and e[1] <= 1.5 * e32[1] + 1e-7, (e, e32)
        # ... and per OUTPUT CHANNEL (a norm over the tensor lets a weak channel hide behind the strong ones)
        def chan(a, r):
            return ((a.double().cpu() - r).pow(2).sum((0, 2, 3)).sqrt() / r.pow(2).sum((0, 2, 3)).sqrt().clamp_min(1e-300))
        for got, r, c in ((y, ref[0], c32[0]), (dx, ref[1], c32[1])):
            ch, c_ = chan(got, r), chan(c.double(), r)
            # (measured worst: 2.2 - 3.1 x on the channel under one_large's outlier, depending on which conv algorithm the host's
            #  fp32 op picked on that box -- its own per-channel error there moves between 2.1e-7 and 3.0e-7; hence the floor)
            floor = 2e-7 if kind == 'one_large' else 1e-7
            assert float((ch / (3.0 * c_ + floor)).max()) <= 1.0, (kind, float(ch.max()), float(c_.max()))
            assert float(ch.max()) <= 1.5 * float(c_.max()) + 1e-7, (kind, float(ch.max()), float(c_.max()))


@pytest.mark.parametrize('N,K,M,S', [(16, 32, 64, 64), (52, 64, 128, 32), (200, 64, 64, 16), (13, 48, 64, 64)])
@pytest.mark.parametrize('dist', ['uniform', 'mixed'])
def test_fp16x3_conv5x5_stride1(N, K, M, S, dist):
    """The 5 x 5 stride-1 convs of the gated stacks (gx_kq.hip Q_C5H) under gx_kq_precision(2): both weight roles against fp64,
    next to the six-bf16-piece form, on uniform operands and on channels spread over seven decades."""
    from genesis_amd import hip_ops as hip, _lib
    x = rnd(N, K, S, S, seed=1) if dist == 'uniform' else _stress('mixed', N, K, S, S, seed=1)
    w0 = rnd(M, K, 5, 5, seed=2, scale=0.1)
    w1 = rnd(K, M, 5, 5, seed=3, scale=0.1)
    r0 = F.conv2d(x.double(), w0.double(), None, 1, 2)
    r1 = F.conv_transpose2d(x.double(), w1.double(), None, 1, 2)
    err = {}
    try:
        for mode in (1, 2):
            _lib.call('gx_kq_precision', mode)
            y0 = hip.conv5x5s1(x.to(DEV), w0.to(DEV), M, False)
            y1 = hip.conv5x5s1(x.to(DEV), w1.to(DEV), M, True)
            err[mode] = (float((y0.double().cpu() - r0).norm() / r0.norm()), float((y1.double().cpu() - r1).norm() / r1.norm()))
    finally:
        _lib.call('gx_kq_precision', -1)
    c32 = (float((F.conv2d(x, w0, None, 1, 2).double() - r0).norm() / r0.norm()),
           float((F.conv_transpose2d(x, w1, None, 1, 2).double() - r1).norm() / r1.norm()))
    print('conv5x5 %s N=%d %d->%d @%d: bf16 x 6 %.3e / %.3e, fp16 x 3 %.3e / %.3e, CPU fp32 %.3e / %.3e'
          % (dist, N, K, M, S, err[1][0], err[1][1], err[2][0], err[2][1], c32[0], c32[1]))
    for i in (0, 1):
        assert err[2][i] <= 1.5 * max(err[1][i], c32[i]) + 1e-7, (err, c32)


def test_amax_link_hands_the_head_gradients_maximum_to_the_data_gradient():
    """gx_kq_amax_link: the decoder head's GroupNorm backward (gx_gn_relu_bwd_proj's two-workgroups-per-CU kernel) writes one partial
    maximum of the dy it stores per workgroup; the transposed conv's data gradient that reads dy next (fp16 x 3 form) reduces those
    instead of launching its own pass over dy.  The maximum is the same number either way: the data gradient is bit-identical with
    and without the link, the hand-over is counted, an unrelated tensor does not take it, and it is one-shot."""
    from genesis_amd import _lib, hip_ops as hip
    N, C, S, Co = 56, 64, 64, 4
    y = rnd(N, C, S, S, seed=1, scale=2.0) + 0.3
    gamma, beta = 1 + 0.3 * rnd(C, seed=2), 0.2 * rnd(C, seed=3)
    g_out = rnd(N, Co, S, S, seed=4)
    ow = rnd(Co, C, seed=5, scale=0.2)
    w = rnd(C, C, 5, 5, seed=6, scale=0.05)
    yd, gd, bd, god, owd, wd = (t.to(DEV) for t in (y, gamma, beta, g_out, ow, w))
    out = torch.empty_like(yd)
    mean, rstd = hip.gn_relu_fwd(yd, gd, bd, 8, 1e-5, (out, 0, 0))

    def run(link):
        h0 = int(_lib.query('gx_kq_amax_link_hits'))
        buf = hip.amax_link(yd.device, yd.numel()) if link else None
        dy = hip.gn_relu_bwd_proj(yd, gd, bd, mean, rstd, 8, god, owd, True)[0]
        dx = hip.deconv5x5s2_dgrad(dy, wd)
        del buf
        return dy, dx, int(_lib.query('gx_kq_amax_link_hits')) - h0
    try:
        _lib.call('gx_kq_precision', 2)
        dy0, dx0, hits0 = run(False)
        dy1, dx1, hits1 = run(True)
        assert hits0 == 0 and hits1 == 1, (hits0, hits1)
        assert torch.equal(dy0, dy1) and torch.equal(dx0, dx1)
        # armed, but the conv's input is another tensor: not taken, and gone afterwards
        h0 = int(_lib.query('gx_kq_amax_link_hits'))
        buf = hip.amax_link(yd.device, yd.numel())
        dy2 = hip.gn_relu_bwd_proj(yd, gd, bd, mean, rstd, 8, god, owd, True)[0]
        other = dy2.clone()
        dx2 = hip.deconv5x5s2_dgrad(other, wd)
        dx3 = hip.deconv5x5s2_dgrad(dy2, wd)
        assert int(_lib.query('gx_kq_amax_link_hits')) == h0
        assert torch.equal(dx2, dx0) and torch.equal(dx3, dx0)
        del buf
    finally:
        _lib.call('gx_kq_precision', -1)
        _lib.call('gx_kq_amax_link', None, 0, 0)


def test_amax_link_from_the_groupnorm_kernels_of_the_decoder_layers():
    """The same hand-over from the register-resident GroupNorm kernels: forward -- the normalised activation's partial maxima go to the
    NEXT layer's transposed conv (hip.deconv5x5s2_gn_relu_fwd(link_out=True)); backward -- dy's go to the layer's data gradient.
    Bit-identical with and without, each hand-over counted once."""
    from genesis_amd import _lib, hip_ops as hip
    N, C, H = 56, 64, 16
    x = rnd(N, C, H, H, seed=1)
    w1, b1 = rnd(C, C, 5, 5, seed=2, scale=0.05), rnd(C, seed=3, scale=0.3)
    w2, b2 = rnd(C, C, 5, 5, seed=4, scale=0.05), rnd(C, seed=5, scale=0.3)
    gamma, beta = 1 + 0.3 * rnd(C, seed=6), 0.2 * rnd(C, seed=7)
    da = rnd(N, C, 2 * H, 2 * H, seed=8)
    xd, w1d, b1d, w2d, b2d, gd, bd, dad = (t.to(DEV) for t in (x, w1, b1, w2, b2, gamma, beta, da))
    hits = lambda: int(_lib.query('gx_kq_amax_link_hits'))      # noqa: E731

    def fwd(link):
        a = torch.empty(N, C, 2 * H, 2 * H, device=DEV)
        y, mean, rstd = hip.deconv5x5s2_gn_relu_fwd(xd, w1d, b1d, gd, bd, 8, 1e-5, (a, 0, 0), link_out=link)
        return a, y, mean, rstd, hip.deconv5x5s2_fwd(a, w2d, b2d)

    def bwd(link, y, mean, rstd):
        buf = hip.amax_link(dad.device, y.numel()) if link else None
        dy = hip.gn_relu_bwd(y, gd, bd, mean, rstd, 8, (dad, 0, 0), None, True)[0]
        dx = hip.deconv5x5s2_dgrad(dy, w1d)
        del buf
        return dy, dx
    try:
        _lib.call('gx_kq_precision', 2)
        _lib.call('gx_kq_policy', 2)          # (the data gradient of 56 images does not fill the chip: every eligible shape)
        h0 = hits()
        a0, y0, mean, rstd, z0 = fwd(False)
        assert hits() == h0
        a1, y1, _, _, z1 = fwd(True)
        assert hits() == h0 + 1
        assert torch.equal(a0, a1) and torch.equal(z0, z1)
        dy0, dx0 = bwd(False, y0, mean, rstd)
        assert hits() == h0 + 1
        dy1, dx1 = bwd(True, y0, mean, rstd)
        assert hits() == h0 + 2
        assert torch.equal(dy0, dy1) and torch.equal(dx0, dx1)
    finally:
        _lib.call('gx_kq_precision', -1)
        _lib.call('gx_kq_policy', 1)
        _lib.call('gx_kq_amax_link', None, 0, 0)


# ------------------------------------------------------------------------------ weight gradients on three fp16 piece products
def _wgq_operands(kind, N, Cin, Cout, S, make_x, make_dy):
    if kind == 'conv3x3':
        x, dy = make_x((N, Cin, S, S)), make_dy((N, Cout, S, S))
    else:
        x, dy = make_x((N, Cin, S, S)), make_dy((N, Cout, 2 * S, 2 * S))
    return x, dy


def _wgq_fp64(kind, x, dy, chunk=8):
    """The weight gradient in fp64 (autograd of the fp64 op), images in chunks (the long-contraction cases are GBs in fp64)."""
    Cin, Cout = x.shape[1], dy.shape[1]
    w = torch.zeros((Cout, Cin, 3, 3) if kind == 'conv3x3' else (Cin, Cout, 5, 5), dtype=torch.float64, requires_grad=True)
    for i in range(0, x.shape[0], chunk):
        xs, ds = x[i:i + chunk].double(), dy[i:i + chunk].double()
        y = F.conv2d(xs, w, None, 1, 1) if kind == 'conv3x3' else F.conv_transpose2d(xs, w, None, 2, 2, 1)
        (y * ds).sum().backward()
    return w.grad


def _wgq_run(kind, x, dy, mode, with_amax):
    from genesis_amd import hip_ops as hip, _lib
    _lib.call('gx_wgq_precision', mode)
    xd, dd = x.to(DEV), dy.to(DEV)
    am = (hip.amax_of(dd), hip.amax_of(xd)) if with_amax else None
    fn = hip.conv3x3_wgrad if kind == 'conv3x3' else hip.deconv5x5s2_wgrad
    return fn(xd, dd, amax=am).double().cpu()


def _per_channel_err(got, ref, dim):
    """relative L2 error per OUTPUT channel (dim 0 of a conv weight, dim 1 of a transposed-conv weight)."""
    dims = [d for d in range(4) if d != dim]
    return ((got - ref).pow(2).sum(dims).sqrt() / ref.pow(2).sum(dims).sqrt().clamp_min(1e-300))


@pytest.mark.parametrize('kind,N,Cin,Cout,S', [('conv3x3', 32, 64, 64, 64), ('conv3x3', 8, 128, 64, 32), ('conv3x3', 16, 64, 128, 16),
                                                ('deconv', 56, 64, 64, 32), ('deconv', 16, 64, 64, 16),
                                                # rows of 128 / 64 input pixels: the strip tiles (seam values in two fp16 pieces)
                                                ('conv3x3', 4, 64, 64, 128), ('deconv', 8, 64, 64, 64),
                                                # 32-channel blocks: the k-split forms (wave pairs split the tile's pixels, finding 24)
                                                ('conv3x3', 32, 32, 64, 64), ('conv3x3', 32, 64, 32, 32), ('conv3x3', 32, 32, 32, 64),
                                                ('deconv', 56, 32, 64, 32)])
def test_weight_gradients_on_three_fp16_piece_products_keep_fp32_accuracy(kind, N, Cin, Cout, S):
    """gx_wgq_precision(2) + gx_wgq_operand_amax: the row-ring tiles with two fp16 pieces per operand value (x * 2^e = hi + lo)
    and three piece products.  Against autograd in fp64, next to the fp32 pipe and the six-bf16-piece form: the fp16 error stays
    within 1.5 x the fp32 pipe's + 1e-7 over the whole tensor AND per output channel; without the operands' maxima the same call
    is the bf16 form bit for bit."""
    from genesis_amd import _lib, profiling
    x, dy = _wgq_operands(kind, N, Cin, Cout, S, lambda s: rnd(*s, seed=1), lambda s: rnd(*s, seed=2))
    ref = _wgq_fp64(kind, x, dy)
    cdim = 0 if kind == 'conv3x3' else 1
    try:
        g32 = _wgq_run(kind, x, dy, 0, False)
        g6 = _wgq_run(kind, x, dy, 1, False)
        g3 = _wgq_run(kind, x, dy, 2, True)
        g3_nohint = _wgq_run(kind, x, dy, 2, False)
    finally:
        _lib.call('gx_wgq_precision', -1)
    assert torch.equal(g3_nohint, g6)
    assert not torch.equal(g3, g6)                  # (the hint was taken: another arithmetic)
    e = {k: float((g - ref).norm() / ref.norm()) for k, g in (('fp32', g32), ('bf16x6', g6), ('fp16x3', g3))}
    c = {k: float(_per_channel_err(g, ref, cdim).max()) for k, g in (('fp32', g32), ('bf16x6', g6), ('fp16x3', g3))}
    print('%s N=%d %d->%d @%d: relative L2 error %s; worst output channel %s' % (
        kind, N, Cin, Cout, S, ' '.join('%s %.3e' % kv for kv in e.items()), ' '.join('%s %.3e' % kv for kv in c.items())))
    assert e['fp16x3'] <= 1.5 * e['fp32'] + 1e-7 and e['fp16x3'] < 1e-4, e
    assert c['fp16x3'] <= 1.5 * c['fp32'] + 1e-7, c


def _flat_rect(shape, seed):
    """Structured operand: five-level flat rectangles per (image, channel) plane (testing.make_rect_input's value distribution) --
    exact zeros, exact ones, large constant regions."""
    g = torch.Generator().manual_seed(seed)
    N, C, H, W = shape
    lv = torch.tensor([0.0, 63.0 / 255.0, 127.0 / 255.0, 191.0 / 255.0, 1.0])
    t = lv[torch.randint(0, 5, (N, C, 1, 1), generator=g)].expand(N, C, H, W).clone()
    for _ in range(3):
        y0, x0 = int(torch.randint(0, H // 2, (1,), generator=g)), int(torch.randint(0, W // 2, (1,), generator=g))
        hh, ww = int(torch.randint(H // 8, H // 2, (1,), generator=g)), int(torch.randint(W // 8, W // 2, (1,), generator=g))
        t[:, :, y0:y0 + hh, x0:x0 + ww] = lv[torch.randint(0, 5, (N, C, 1, 1), generator=g)]
    return t


@pytest.mark.parametrize('case', ['randn', 'relu_like', 'flat_rect', 'weak_channel', 'one_outlier'])
def test_long_contraction_weight_gradient_on_fp16_pieces_per_output_channel(case):
    """The precondition for running the weight gradients on fp16 pieces (review, round 5): a contraction as long as the metric
    step's -- N H W = 224 x 64 x 64 = 9.2e5 terms per output element (the decoder's last layer: K B = 224 images) -- on
      randn         both operands standard normal;
      relu_like     x >= 0 with half of it exactly zero (a post-ReLU activation), dy heavy-tailed (normal^3);
      flat_rect     x = five-level flat rectangles (the structured input set's value distribution), dy normal;
      weak_channel  as randn, with ONE dy channel 2^12 below the others (its gradients must be as accurate as its neighbours');
      one_outlier   as randn, with one dy element 2^20 times the rest (the per-tensor scale follows it: everything else has
                    its low piece in fp16's subnormal range -- the documented range limit of the per-tensor scale).
    Per OUTPUT channel, error against fp64.  What a contraction of this length does (measured, MI355X, worst channel): the fp32
    pipe 1.0 - 1.2e-6 (2.1e-6 on flat_rect), the six-bf16-piece form -- the default of rounds 2 - 5 -- 3.1 - 3.6e-6, three fp16 pieces
    1.6e-6 (randn, weak_channel: 1.3 x the fp32 pipe), 1.9e-6 (flat_rect: 0.9 x), 2.1e-6 (relu_like: 2.1 x), 2.6e-6 (one_outlier:
    2.3 x).  At 9e5 terms every form is bound by the fp32 ACCUMULATION of the matrix instructions (a 16-product MFMA rounds more
    than once), not by the operands' representation: halving the MFMAs of the bf16 form halves that error.  The review's bar for
    moving the weight gradients to fp16 pieces was 1.5 x the fp32 pipe per channel: it holds on three of the five operand sets;
    asserted here is what IS true of all five -- never worse than 2.5 x the fp32 pipe + 1e-7 per output channel, and never worse
    than the bf16 form it replaces."""
    from genesis_amd import _lib
    N, Cin, Cout, S = 224, 64, 64, 64
    shape_x, shape_d = (N, Cin, S, S), (N, Cout, S, S)
    if case == 'randn':
        x, dy = rnd(*shape_x, seed=11), rnd(*shape_d, seed=12)
    elif case == 'relu_like':
        x, dy = rnd(*shape_x, seed=13).clamp_min(0.0), rnd(*shape_d, seed=14).pow(3)
    elif case == 'flat_rect':
        x, dy = _flat_rect(shape_x, 15), rnd(*shape_d, seed=16)
    elif case == 'weak_channel':
        x, dy = rnd(*shape_x, seed=17), rnd(*shape_d, seed=18)
        dy[:, 5] *= 2.0 ** -12
    else:
        x, dy = rnd(*shape_x, seed=19), rnd(*shape_d, seed=20)
        dy[7, 3, 21, 40] = 2.0 ** 20
    ref = _wgq_fp64('conv3x3', x, dy)
    try:
        g32 = _wgq_run('conv3x3', x, dy, 0, False)
        g6 = _wgq_run('conv3x3', x, dy, 1, False)
        g3 = _wgq_run('conv3x3', x, dy, 2, True)
    finally:
        _lib.call('gx_wgq_precision', -1)
    c32, c6, c3 = (_per_channel_err(g, ref, 0) for g in (g32, g6, g3))
    print('%s: worst output channel, relative L2 vs fp64: fp32 pipe %.3e, bf16 x 6 %.3e, fp16 x 3 %.3e; median fp32 %.3e fp16 %.3e'
          % (case, float(c32.max()), float(c6.max()), float(c3.max()), float(c32.median()), float(c3.median())))
    assert not torch.equal(g3, g6)
    assert float((c3 / (2.5 * c32 + 1e-7)).max()) <= 1.0, (case, float(c3.max()), float(c32.max()))
    assert float(c3.max()) <= float(c6.max()) * 1.05 + 1e-7, (case, float(c3.max()), float(c6.max()))
    if case in ('randn', 'weak_channel', 'flat_rect'):
        assert float((c3 / (1.5 * c32 + 1e-7)).max()) <= 1.0, (case, float(c3.max()), float(c32.max()))
    if case == 'weak_channel':
        # the channel 2^12 below the others is as accurate as its neighbours (the scale is per TENSOR: 12 of its 17 spare bits)
        assert float(c3[5]) <= 1.5 * float(c3.median()) + 1e-7, (float(c3[5]), float(c3.median()))


# ------------------------------------------------------------------------------ Winograd conv3x3 on three fp16 piece products
@pytest.mark.parametrize('N,Cin,Cout,S,mode', [(32, 64, 64, 64, 0), (32, 64, 64, 64, 1), (8, 128, 64, 32, 0), (8, 64, 128, 32, 1),
                                               (4, 256, 64, 64, 0), (16, 48, 80, 16, 0)])
def test_winograd_on_three_fp16_piece_products_keeps_fp32_accuracy(N, Cin, Cout, S, mode):
    """gx_wino_precision(2) + gx_conv_input_amax: U = G g G^T packed as two fp16 pieces of U * 2^eU (eU from max |w|), V = B^T d B
    split into two fp16 pieces of V * 2^eV in registers (eV from 4 max |x|, handed in as partial maxima), three piece products.
    Forward (mode 0) and data gradient (mode 1) against fp64, next to the fp32 pipe and the six-bf16-piece form: within 1.5 x the
    fp32 pipe's error + 1e-7 over the tensor and per output channel; without the hint the call is the bf16 form bit for bit;
    operands at 2^-30 / 2^+30 of the usual scale, and a weak input channel, change nothing (the scales follow the tensors)."""
    from genesis_amd import hip_ops as hip, _lib
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.1)
    x = rnd(N, Cin if mode == 0 else Cout, S, S, seed=1)

    def ref_of(xx, ww):
        return F.conv2d(xx.double(), ww.double(), None, 1, 1) if mode == 0 else F.conv_transpose2d(xx.double(), ww.double(), None, 1, 1)

    def run(xx, ww, prec, hint):
        _lib.call('gx_wino_precision', prec)
        xd = xx.to(DEV)
        return hip.conv3x3_wino(xd, ww.to(DEV), mode, amax_in=hip.amax_of(xd) if hint else None).double().cpu()

    def chan_err(got, ref):
        return ((got - ref).pow(2).sum((0, 2, 3)).sqrt() / ref.pow(2).sum((0, 2, 3)).sqrt().clamp_min(1e-300))
    try:
        ref = ref_of(x, w)
        y32, y6, y3, y3n = run(x, w, 0, False), run(x, w, 1, False), run(x, w, 2, True), run(x, w, 2, False)
        assert torch.equal(y3n, y6) and not torch.equal(y3, y6)
        e = {k: float((v - ref).norm() / ref.norm()) for k, v in (('fp32', y32), ('bf16x6', y6), ('fp16x3', y3))}
        c = {k: float(chan_err(v, ref).max()) for k, v in (('fp32', y32), ('bf16x6', y6), ('fp16x3', y3))}
        print('winograd %s N=%d %d->%d @%d: relative L2 %s; worst channel %s' % ('fwd' if mode == 0 else 'dgrad', N, Cin, Cout, S,
              ' '.join('%s %.3e' % kv for kv in e.items()), ' '.join('%s %.3e' % kv for kv in c.items())))
        assert e['fp16x3'] <= 1.5 * e['fp32'] + 1e-7 and e['fp16x3'] < 1e-5, e
        assert c['fp16x3'] <= 1.5 * c['fp32'] + 1e-7, c
        # the scales follow the tensors
        for sx, sw in ((2.0 ** -30, 1.0), (2.0 ** 30, 2.0 ** -20)):
            ys = run(x * sx, w * sw, 2, True)
            assert torch.equal(ys, y3 * (sx * sw)), (sx, sw)
        # one input channel 2^12 below the others: its contribution is as accurate as before (per-tensor scale, 17 spare bits)
        xw = x.clone()
        xw[:, 3] *= 2.0 ** -12
        refw = ref_of(xw, w)
        ew = {k: float((run(xw, w, p, h) - refw).norm() / refw.norm()) for k, p, h in (('fp32', 0, False), ('fp16x3', 2, True))}
        assert ew['fp16x3'] <= 1.5 * ew['fp32'] + 1e-7, ew
    finally:
        _lib.call('gx_wino_precision', -1)
