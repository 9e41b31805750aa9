"""Model-level parity of the HIP product module (genesis_amd.genesisv2_config.GenesisV2) against the
golden vectors captured from the real reference, and against the oracle on fresh seeded inputs.
fp32 tolerances: forward tensors rtol 1e-4 / atol 1e-5; ELBO within 1e-3 relative (north_star)."""
import numpy as np
import pytest
import torch

from tests.common import DYN_CASES, SAMPLE_CASES, Golden, SampleGolden

pytestmark = pytest.mark.gpu
DEV = 'cuda'

DEFAULT_CASES = ['tiny', 'tiny_b3k3', 'tiny_noar', 'tiny_klm', 'tiny_klm_nodetach', 'tiny_nosemi', 'tiny_laplacian',
                 'tiny_epanechnikov', 'metric', 'cfg2', 'cfg5']


def build(gold):
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    cfg = AttrDict(dict(dict(dynamic_K=False), **dict(gold.cfg, debug=False, multi_gpu=False)))
    torch.manual_seed(0)
    model = G.load(cfg)
    sd = gold.weights(model.state_dict())
    model.load_state_dict(sd)
    return model.to(DEV).train()


SEED_FALLBACKS = []     # (case, number of seed pixels that differed) whenever the reference's seeds had to be injected


def run(model, gold, x, rand_pixel, eps_k):
    eps = torch.stack(eps_k).to(DEV)
    out = model(x.to(DEV), rand_pixel.to(DEV), eps)
    if out[3] is None or 'seed_idx' not in gold.g.files:       # dynamic_K cases carry no seed list
        return out
    seed_idx = torch.stack(list(out[3]['seed_idx'])).cpu().numpy()
    if not np.array_equal(seed_idx, gold.g['seed_idx']):
        # near-tie in the discontinuous argmax: replay with the reference's seeds -- recorded, and
        # test_seed_injection_fallback_never_fires fails if it ever happens on the committed fixtures (their noise seeds
        # were chosen with a top-2 margin > 2e-5, make_golden.py), so a systematic argmax bug cannot hide behind it
        SEED_FALLBACKS.append((gold.name, int((seed_idx != gold.g['seed_idx']).sum())))
        assert float(gold.g['seed_margin'].min()) < 1e-3
        forced = torch.from_numpy(gold.g['seed_idx']).to(DEV)
        out = model(x.to(DEV), rand_pixel.to(DEV), eps, forced)
    return out


@pytest.mark.parametrize('case', DEFAULT_CASES + DYN_CASES)
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
    # gradient tolerance per case, from measurements (tools/tol_probe.py, tests/test_error_budget_gpu.py): on the 32x32
    # cases HIP and the reference agree to <= 8e-5 (1.7e-3 with the undetached mask KL); on the closed-form weights of
    # the 64x64 / 128x128 cases fp32 ITSELF is only good to 8e-3 -- the reference's own arithmetic sits 7.9e-3 from
    # fp64 there, the HIP path 5.6e-3 (test_genesis_v2_on_the_golden_cases_weights_and_inputs) -- so two fp32 results
    # may differ by ~1e-2
    big = case in ('metric', 'cfg2', 'cfg5')
    l2 = 1e-2 if big else (4e-3 if case == 'tiny_klm_nodetach' else 5e-4)
    gold.check_grads(grads, rtol=5e-3 if big else l2, l2_tol=l2)
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
    cfg = AttrDict(dict(dict(dynamic_K=False), **dict(gold.cfg, debug=False, multi_gpu=False)))
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


def test_seed_injection_fallback_never_fires():
    """Runs after the golden comparisons of this module (pytest keeps file order): none of them may have needed the
    reference's seed pixels injected."""
    for case in DEFAULT_CASES:
        gold = Golden(case)
        model = build(gold)
        x, rand_pixel, eps_k = gold.inputs()
        with torch.no_grad():
            run(model, gold, x, rand_pixel, eps_k)
    assert SEED_FALLBACKS == [], SEED_FALLBACKS


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


def test_cpu_input_raises():
    from genesis_amd._lib import GenesisHipError
    gold = Golden('tiny')
    model = build(gold)
    with pytest.raises((GenesisHipError, RuntimeError)):
        model.cpu()(torch.rand(1, 3, 32, 32))


# Gradient bar of this test (review, round 5: "derived from a budget instead of the 3e-2"): per PARAMETER, from the fp32 error budget of
# the fixture's own closed-form weights -- tests/golden/v2_<case>_budget.npz (make_golden_budget.py: how far the oracle's fp32 gradient
# sits from its fp64 gradient; up to 8e-3 on these ill-conditioned weights, median 6e-4) --
#     |HIP - reference| <= (4 + 1) x budget + 5e-5 + 3e-3        (relative L2 on the strided samples, and on the norm)
# i.e. tests/test_fullbatch_gpu.py's bar: 4 x for the HIP path (the fp64 budget tests hold it to that), 1 x for the reference's own
# distance, and the ReLU-decision allowance, which on the closed-form weights is granted to every parameter (one decoder ReLU is known
# to move on them: DESIGN.md finding 12; the full-batch fixtures grant it per layer).
@pytest.mark.parametrize('case,l2_tol', [('metric', None), ('cfg5', None)])
def test_golden_parity_with_every_conv3x3_on_the_winograd_kernel(case, l2_tol):
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
        import os.path as osp
        import numpy as np
        from genesis_amd import testing as T
        from tests.common import GOLDEN
        bud = np.load(osp.join(GOLDEN, 'v2_%s_budget.npz' % case), allow_pickle=False)
        names = [str(n) for n in bud['param_names']]
        gmax = float(bud['grad_max_f64'])
        named = dict(model.named_parameters())
        assert names == [n for n, _ in model.named_parameters()]
        table = []
        for i, name in enumerate(names):
            p = named[name]
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            sm = T.summarize(g)
            ref = gold.g['grad/%s/samples' % name].astype(np.float64)
            den = max(float(bud['grad_norms_f64'][i]), 1e-6 * gmax)
            e_s = float(np.linalg.norm(sm['samples'].astype(np.float64) - ref)) / (den * np.sqrt(len(ref) / max(1, int(sm['n']))))
            e_n = abs(float(g.double().norm()) - float(gold.g['grad_norms'][i])) / den
            bar = 5.0 * float(bud['budget'][i]) + 5e-5 + 3e-3
            table.append((max(e_s, e_n) / bar, name, e_s, e_n, float(bud['budget'][i])))
        table.sort(reverse=True)
        print('%s on the Winograd kernel: worst gradient error / bar = %.3f' % (case, table[0][0]))
        for r in table[:4]:
            print('   %-44s samples %.2e norm %.2e budget %.2e (%.2f of the bar)' % (r[1], r[2], r[3], r[4], r[0]))
        assert table[0][0] <= 1.0, table[0]
    finally:
        profiling.enable(False)
        _lib.call('gx_conv3x3_wino_policy', 1)

