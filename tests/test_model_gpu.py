"""Model-level parity of the HIP product module (genesis_amd.genesisv2_config.GenesisV2) against the
golden vectors captured from the real reference, and against the oracle on fresh seeded inputs.
fp32 tolerances: forward tensors rtol 1e-4 / atol 1e-5; ELBO within 1e-3 relative (north_star)."""
import numpy as np
import pytest
import torch

from tests.common import SAMPLE_CASES, Golden, SampleGolden

pytestmark = pytest.mark.gpu
DEV = 'cuda'

DEFAULT_CASES = ['tiny', 'tiny_b3k3', 'tiny_noar', 'tiny_klm', 'tiny_klm_nodetach', 'tiny_nosemi', 'tiny_laplacian',
                 'tiny_epanechnikov', 'metric', 'cfg2', 'cfg5']


def build(gold):
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    cfg = AttrDict(dict(gold.cfg, debug=False, multi_gpu=False, dynamic_K=False))
    torch.manual_seed(0)
    model = G.load(cfg)
    sd = gold.weights(model.state_dict())
    model.load_state_dict(sd)
    return model.to(DEV).train()


def run(model, gold, x, rand_pixel, eps_k):
    eps = torch.stack(eps_k).to(DEV)
    out = model(x.to(DEV), rand_pixel.to(DEV), eps)
    seed_idx = torch.stack(list(out[3]['seed_idx'])).cpu().numpy()
    if not np.array_equal(seed_idx, gold.g['seed_idx']):
        # near-tie in the discontinuous argmax: replay with the reference's seeds
        assert float(gold.g['seed_margin'].min()) < 1e-3
        forced = torch.from_numpy(gold.g['seed_idx']).to(DEV)
        out = model(x.to(DEV), rand_pixel.to(DEV), eps, forced)
    return out


@pytest.mark.parametrize('case', DEFAULT_CASES)
def test_forward_and_grads_vs_golden(case):
    gold = Golden(case)
    model = build(gold)
    want = torch.float32 if gold.cfg['kernel'] == 'epanechnikov' else torch.float64
    assert model.att_process.log_sigma.dtype == want   # modules/attention.py:145-155
    x, rand_pixel, eps_k = gold.inputs()
    recon, losses, stats, att, comp = run(model, gold, x, rand_pixel, eps_k)
    # log-masks accumulate K-1 stick-breaking steps of log(1-alpha) (slope up to 100 at the 0.99 clamp)
    gold.check_forward(recon, losses, stats, att, comp, rtol=1e-4, atol=2e-5, mask_atol=1e-3)
    err = losses.err.mean(0)
    kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
    if 'kl_m' in losses:
        kl = kl + losses.kl_m.mean(0)
    elbo_ref = float(gold.g['loss/err']) + float(gold.g['loss/kl'])
    assert abs(float(err + kl) - elbo_ref) <= 1e-3 * abs(elbo_ref)          # north_star: ELBO within 1e-3
    assert abs(float(err + kl) - elbo_ref) <= 2e-5 * abs(elbo_ref)          # what fp32 actually gives
    (err + kl).backward()
    grads = [(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()]
    gold.check_grads(grads, rtol=5e-3, l2_tol=1e-2)
    # reference's own invariant (utils/misc.py:258-270)
    for key in ('log_m_k', 'log_m_r_k'):
        s = torch.stack(stats[key], 4).exp().sum(4)
        assert float((s - 1).abs().max()) < 1e-3


@pytest.mark.parametrize('case', SAMPLE_CASES)
def test_sample_vs_golden(case):
    """GenesisV2.sample (reference models/genesisv2_config.py:227-256) on the HIP kernels -- AR-prior rollout through
    gx_linear_fwd / gx_lstm_step_fwd / gx_latent_prior_sample, then the decoder -- against the reference's own sample()
    outputs on its recorded standard-normal draws (tests/golden/make_golden_sample.py)."""
    gold = SampleGolden(case)
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    from genesis_amd import testing as T
    cfg = AttrDict(dict(gold.cfg, debug=False, multi_gpu=False, dynamic_K=False))
    torch.manual_seed(0)
    model = G.load(cfg)
    model.load_state_dict(T.formula_state_dict(model.state_dict()))
    model = model.to(DEV).eval()
    recon, stats = model.sample(gold.B, gold.K_arg, eps=gold.eps.to(DEV))
    assert len(stats.x_k) == gold.eps.shape[0]
    gold.check_all(recon, stats.x_k, stats.log_m_k, stats.z_k, rtol=1e-4, atol=2e-5, mx_k=stats.mx_k)
    # default path (own noise): same shapes, masks sum to one
    recon2, st2 = model.sample(gold.B, gold.K_arg)
    assert recon2.shape == recon.shape
    assert float((torch.stack(st2.log_m_k, 4).exp().sum(4) - 1).abs().max()) < 1e-3


@pytest.mark.parametrize('case', ['tiny', 'tiny_klm', 'metric', 'cfg5'])
def test_no_grad_forward_vs_golden(case):
    """The evaluation forward (train.py:512-530 runs the model under torch.no_grad() in eval mode) against the golden
    forward tensors: same kernels, no autograd graph."""
    gold = Golden(case)
    model = build(gold).eval()
    x, rand_pixel, eps_k = gold.inputs()
    with torch.no_grad():
        recon, losses, stats, att, comp = run(model, gold, x, rand_pixel, eps_k)
    assert not recon.requires_grad and not losses.err.requires_grad
    gold.check_forward(recon, losses, stats, att, comp, rtol=1e-4, atol=2e-5, mask_atol=1e-3)


def test_output_contract():
    """Shapes / types / access patterns train.py and the scripts rely on."""
    gold = Golden('tiny')
    model = build(gold)
    B, K, S, D = gold.B, gold.K, gold.S, gold.D
    x, _, _ = gold.inputs()
    torch.manual_seed(3)
    recon, losses, stats, att_stats, comp_stats = model(x.to(DEV))
    assert recon.shape == (B, 3, S, S) and losses.err.shape == (B,)
    assert len(losses.kl_l_k) == K and losses['kl_l_k'][0].shape == (B,)
    assert torch.stack(losses.kl_l_k, dim=1).shape == (B, K)
    assert torch.cat(stats.log_m_k, 1).shape == (B, K, S, S)
    assert len(stats.log_s_k) == K and stats.x_r_k[0].shape == (B, 3, S, S)
    assert stats.instance_seg.shape == (B, S, S) and stats.instance_seg.dtype == torch.int64
    assert att_stats.colour.shape == (B, 8, S, S) and att_stats.delta.shape == (B, 2, S, S)
    assert len(att_stats.seeds) == K - 1 and att_stats.seeds[0].shape == (B, 8)
    assert len(comp_stats.z_k) == K and comp_stats.z_k[0].shape == (B, D)
    assert 'kl_l_k' in losses and 'kl_m' not in losses
    assert model.get_features(x.to(DEV)).shape == (B, K * D)
    img, st = model.sample(3)
    assert img.shape == (3, 3, S, S) and len(st.x_k) == K and st.log_m_k[0].shape == (3, 1, S, S)
    img, st = model.sample(2, K_steps=6)
    assert len(st.mx_k) == 6
    assert float((torch.stack(st.log_m_k, 4).exp().sum(4) - 1).abs().max()) < 1e-3


def test_error_budget_vs_fp64_oracle():
    """Rigorous fp32 check: the oracle run in fp64 is ground truth.  Forward tensors: the HIP module's
    relative L2 error must be of the same class as the fp32 CPU oracle's own error (<= 4x + 2e-6).
    Gradients: <= 2e-3 relative L2 -- a single ReLU pre-activation within ~1e-7 of zero flipping
    sign (seen on both the CPU-fp32 and the HIP side, tools/diag_model_decoder.py) moves every
    upstream gradient by ~1e-4..1e-3; the table printed below shows both columns."""
    from oracle import v2_oracle as O
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    from genesis_amd import testing as T
    cfg = O.make_cfg(K_steps=5, img_size=64, feat_dim=32)
    torch.manual_seed(7)
    model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False)))
    with torch.no_grad():
        model.att_process.colour_head.gate.gate.fill_(0.2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    B = 4
    x = T.make_input(99, B, 64)
    rp, eps_k = T.draw_noise(123, B, 64, 32, 5)

    def oracle(dtype):
        p = {k: v.clone().to(dtype if v.dtype == torch.float32 else v.dtype).requires_grad_(True)
             for k, v in sd.items()}
        out = O.v2_forward(p, x.to(dtype), cfg, rp.to(dtype), [e.to(dtype) for e in eps_k], reference_form=False)
        e, kl, _ = O.aggregate_losses(out[1])
        (e + kl).backward()
        return out, {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in p.items()}

    o64, g64 = oracle(torch.float64)
    o32, g32 = oracle(torch.float32)
    model = model.to(DEV)
    recon, losses, stats, att, _ = model(x.to(DEV), rp.to(DEV), torch.stack(eps_k).to(DEV))
    assert torch.equal(torch.stack(list(att['seed_idx'])).cpu(), torch.stack(o64[3]['seed_idx']))
    (losses.err.mean(0) + torch.stack(losses.kl_l_k, 1).mean(0).sum()).backward()

    def relerr(a, ref):
        return float((a.double().cpu() - ref).norm()) / (float(ref.norm()) + 1e-30)

    rows = []
    bad = []
    pairs = [('recon', recon, o32[0], o64[0]), ('err', losses.err, o32[1]['err'], o64[1]['err']),
             ('log_m', torch.stack(list(stats.log_m_k)), torch.stack(o32[2]['log_m_k']), torch.stack(o64[2]['log_m_k'])),
             ('kl', torch.stack(list(losses.kl_l_k)), torch.stack(o32[1]['kl_l_k']), torch.stack(o64[1]['kl_l_k']))]
    for name, got, r32, r64 in pairs:
        e_gpu, e_cpu = relerr(got.detach(), r64.detach().double()), relerr(r32.detach(), r64.detach().double())
        rows.append((name, e_gpu, e_cpu))
        if e_gpu > 4 * e_cpu + 2e-6:
            bad.append(name)
    gmax = max(float(v.norm()) for v in g64.values())
    for n, prm in model.named_parameters():
        got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        ref = g64[n]
        floor = 2e-6 * gmax / (float(ref.norm()) + 1e-30)   # analytically-zero grads: absolute floor
        e_gpu, e_cpu = relerr(got, ref), relerr(g32[n], ref)
        rows.append(('grad ' + n, e_gpu, e_cpu))
        if e_gpu > max(4 * e_cpu + 2e-6, 2e-3) + floor:
            bad.append(n)
    print('\n%-44s %12s %12s' % ('tensor', 'hip-vs-f64', 'cpu32-vs-f64'))
    for r in rows:
        print('%-44s %12.3e %12.3e' % r)
    assert not bad, bad


def test_cpu_input_raises():
    from genesis_amd._lib import GenesisHipError
    gold = Golden('tiny')
    model = build(gold)
    with pytest.raises((GenesisHipError, RuntimeError)):
        model.cpu()(torch.rand(1, 3, 32, 32))


@pytest.mark.parametrize('case', ['metric', 'cfg5'])
def test_golden_parity_with_every_conv3x3_on_the_winograd_kernel(case):
    """The golden cases have B = 2, too small for the Winograd dispatch (it takes the layers that fill the chip): force
    every eligible conv3x3 forward / data gradient onto the Winograd kernel and repeat the reference comparison."""
    from genesis_amd import _lib, profiling
    gold = Golden(case)
    model = build(gold)
    x, rand_pixel, eps_k = gold.inputs()
    _lib.call('gx_conv3x3_wino_policy', 2)
    try:
        profiling.enable(True)
        recon, losses, stats, att, comp = run(model, gold, x, rand_pixel, eps_k)
        gold.check_forward(recon, losses, stats, att, comp, rtol=1e-4, atol=2e-5, mask_atol=1e-3)
        err = losses.err.mean(0)
        kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
        elbo_ref = float(gold.g['loss/err']) + float(gold.g['loss/kl'])
        assert abs(float((err + kl).detach()) - elbo_ref) <= 2e-5 * abs(elbo_ref)
        (err + kl).backward()
        rows = {r['name']: r['launches'] for r in profiling.collect()}
        assert rows.get('wino_conv_kernel', 0) >= 10, rows          # UNet 32x32 / 64x64 levels and both heads, fwd + dgrad
        grads = [(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()]
        gold.check_grads(grads, rtol=5e-3, l2_tol=1e-2)
    finally:
        profiling.enable(False)
        _lib.call('gx_conv3x3_wino_policy', 1)

