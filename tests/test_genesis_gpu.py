"""BaselineVAE (BASELINE config 1) and GENESIS v1 (config 3) on the HIP path vs golden vectors captured from the
real reference: forward tensors, parameter gradients, three GECO + Adam steps."""
import numpy as np
import pytest
import torch

from tests.test_genesis_oracle import GEN_CASES, VAE_CASES, Gold

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build(gold, mod):
    from genesis_amd.compat.attrdict import AttrDict
    cfg = AttrDict(dict(gold.cfg, debug=False, multi_gpu=False))
    torch.manual_seed(0)
    model = mod.load(cfg)
    model.load_state_dict(gold.weights(model.state_dict()))
    return model.to(DEV).train()


def grads(model):
    return [(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()]


@pytest.mark.parametrize('case', VAE_CASES)
def test_vae_forward_grads_and_steps(case):
    import genesis_amd.vae_config as G
    from genesis_amd.trainer import TrainStep
    gold = Gold('vae', case)
    model = build(gold, G)
    x = gold.x()
    L = gold.cfg['latent_dimension']
    (eps,) = gold.replay([(gold.B, L)])
    recon, losses, stats, _, _ = model(x.to(DEV), eps.to(DEV))
    for k, t in (('err', losses.err), ('kl_l', losses.kl_l), ('recon', recon), ('mu', stats.mu), ('z', stats.z)):
        gold.check(k, t, 1e-4, 2e-5)
    (losses.err.mean(0) + losses.kl_l.mean(0)).backward()
    gold.check_grads(grads(model), rtol=5e-3, l2_tol=1e-2)
    model = build(gold, G)
    ts = TrainStep(model, gold.S, lr=1e-4)
    hist = gold.g['train_hist']
    for it in range(3):
        (e,) = gold.replay([(gold.B, L)], 1 + it)
        out = ts.step(x.to(DEV), eps=e.to(DEV)).cpu().numpy()
        assert abs(out[0] - hist[it, 0]) <= 1e-3 * abs(hist[it, 0]), (it, out, hist[it])
        np.testing.assert_allclose(out[[1, 3]], hist[it, [1, 3]], rtol=5e-4)
    img, st = model.sample(3)
    assert img.shape == (3, 3, gold.S, gold.S) and model.get_features(x.to(DEV)).shape == (gold.B, L)


@pytest.mark.parametrize('case', GEN_CASES)
def test_genesis_forward_grads_and_steps(case):
    import genesis_amd.genesis_config as G
    from genesis_amd.trainer import TrainStep
    gold = Gold('genesis', case)
    cfg = gold.cfg
    K, L, Lc = cfg['K_steps'], cfg['attention_latents'], cfg['comp_ldim']
    model = build(gold, G)
    x = gold.x()
    two = cfg.get('two_stage', True)
    noise = gold.replay([(gold.B, L)] * K + ([(K * gold.B, Lc)] if two else []))
    recon, losses, stats, att, comp = model(x.to(DEV), [n.to(DEV) for n in noise[:K]], noise[K].to(DEV) if two else None)
    st = lambda l: torch.stack(list(l))  # noqa: E731
    gold.check('err', losses.err, 1e-4, 1e-3)
    gold.check('kl_m_k', st(losses.kl_m_k), 1e-3, 5e-3)
    if two:
        gold.check('kl_l_k', st(losses.kl_l_k), 1e-3, 5e-3)
        gold.check('comp_z_k', st(comp.z_k), 1e-4, 5e-5)
    else:
        assert comp is None and 'kl_l_k' not in losses
    gold.check('recon', recon, 1e-4, 2e-5)
    gold.check('log_m_k', st(stats.log_m_k), 1e-4, 1e-3)
    gold.check('x_r_k', st(stats.x_r_k), 1e-4, 2e-5)
    gold.check('att_z_k', st(att.z_k), 1e-4, 5e-5)
    # the visualisation-only statistics are evaluated on first access (models/genesis_config.py:256-270 returns them eagerly)
    mx = st(stats.mx_r_k)
    assert torch.allclose(mx, st(stats.x_r_k) * st(stats.log_m_k).exp())
    pmu, psig = st(att.pmu_k), st(att.psigma_k)
    assert pmu.shape == st(att.mu_k).shape and float(pmu[0].abs().max()) == 0.0 and float((psig[0] - 1).abs().max()) == 0.0
    assert float(pmu.abs().max()) <= 1.0 and float(psig.min()) >= 1e-4
    if two and cfg.get('comp_prior', True):
        assert st(comp.pmu_k).shape == st(comp.mu_k).shape and float(st(comp.psigma_k).min()) >= 1e-4
    err = losses.err.mean(0)
    kl = torch.stack(losses.kl_m_k, 1).mean(0).sum()
    if two:
        kl = kl + torch.stack(losses.kl_l_k, 1).mean(0).sum()
    elbo_ref = float(gold.g['loss/err']) + float(gold.g['loss/kl'])
    assert abs(float(err + kl) - elbo_ref) <= 1e-4 * abs(elbo_ref)
    (err + kl).backward()
    # per-parameter tolerance from the fp32 error budget of this case (tests.common.fp32_budget): 5 x (CPU-fp32 vs fp64) + 5e-4, capped at the round-2 constant
    from tests.common import budget_tolerances, fp32_budget
    from tests.test_genesis_oracle import is_param
    from oracle import genesis_oracle as GO
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    def loss_fn(p, dtype):
        out = GO.genesis_forward(p, x.to(dtype), cfg, [n.to(dtype) for n in noise[:K]], noise[K].to(dtype) if two else None)
        e, kl_l, kl_m = GO.aggregate_losses(out[1])
        return e + kl_l + kl_m
    e_cpu = fp32_budget(loss_fn, sd, is_param=is_param)
    print('GENESIS %s: fp32 budget per parameter: max %.3e, median %.3e' % (case, max(e_cpu.values()), sorted(e_cpu.values())[len(e_cpu) // 2]))
    gold.check_grads(grads(model), per_param=budget_tolerances(e_cpu, floor=5e-4, cap=5e-2))
    assert float((torch.stack(stats.log_m_k, 4).exp().sum(4) - 1).abs().max()) < 1e-3
    # three training steps (BatchNorm running statistics restart from the fixture's values)
    model = build(gold, G)
    ts = TrainStep(model, gold.S, lr=1e-4)
    hist = gold.g['train_hist']
    for it in range(3):
        nz = gold.replay([(gold.B, L)] * K + ([(K * gold.B, Lc)] if two else []), 1 + it)
        out = ts.step(x.to(DEV), eps_m=[n.to(DEV) for n in nz[:K]], eps_c=nz[K].to(DEV) if two else None).cpu().numpy()
        assert abs(out[0] - hist[it, 0]) <= 1e-3 * abs(hist[it, 0]), (it, out, hist[it])
    assert int(model.att_process.core.q_z_nn[0].h_norm.num_batches_tracked) == 3 if cfg['enc_norm'] == 'bn' else True


@pytest.mark.parametrize('K,B,F_,D', [(7, 32, 256, 64), (2, 5, 256, 16), (1, 3, 256, 64)])
def test_latent_sbp_posterior_node_equals_the_chained_functions(K, B, F_, D):
    """LatentSBPPosteriorFn (the recurrent posterior of modules/attention.py:84-118 as one autograd node) against the same
    recurrence written with one Function per op (LSTMCellFn -> LinearFn -> PosteriorFn, torch.cat between them): outputs and
    every gradient, including h's (K + 1 uses) and the cell's parameters (K - 1 uses)."""
    from genesis_amd import functions as fn
    g = torch.Generator().manual_seed(3)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)      # noqa: E731
    H = 2 * D
    names = ['h', 'w_m', 'b_m', 'w_v', 'b_v', 'w_ih', 'w_hh', 'b_ih', 'b_hh', 'w_lin', 'b_lin']
    vals = [r(B, F_), r(D, F_, sc=0.05), r(D, sc=0.1), r(D, F_, sc=0.05), r(D, sc=0.1), r(4 * H, F_ + D, sc=0.05),
            r(4 * H, H, sc=0.08), r(4 * H, sc=0.1), r(4 * H, sc=0.1), r(2 * D, H, sc=0.1), r(2 * D, sc=0.1)]
    eps = r(K, B, D)
    cz, cl, cm, cs = r(K, B, D), r(K, B), r(K, B, D), r(K, B, D)

    def chained(h, w_m, b_m, w_v, b_v, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin):
        lin = torch.cat((fn.linear(h, w_m, b_m), fn.linear(h, w_v, b_v)), 1)
        zs, mus, sgs, lqs = [], [], [], []
        hs = cs_ = None
        for k in range(K):
            if k:
                hs, cs_ = fn.LSTMCellFn.apply(torch.cat([h, zs[-1]], 1), hs, cs_, w_ih, w_hh, b_ih, b_hh)
                lin = fn.linear(hs, w_lin, b_lin)
            z1, m1, s1, q1 = fn.PosteriorFn.apply(lin.unsqueeze(1), eps[k].unsqueeze(0))
            zs.append(z1.view(B, -1)); mus.append(m1.view(B, -1)); sgs.append(s1.view(B, -1)); lqs.append(q1)
        return torch.stack(zs), torch.stack(mus), torch.stack(sgs), torch.cat(lqs, 0)

    def node(h, *params):
        return fn.LatentSBPPosteriorFn.apply(h, eps, *params)

    res = []
    for f in (chained, node):
        leaves = [v.clone().requires_grad_(True) for v in vals]
        z, mu, sg, lq = f(*leaves)
        ((z * cz).sum() + (lq * cl).sum() + (mu * cm).sum() + (sg * cs).sum()).backward()
        res.append(([z, mu, sg, lq], [l.grad for l in leaves]))
    for a, b_ in zip(*[r_[0] for r_ in res]):
        torch.testing.assert_close(b_, a, rtol=1e-5, atol=1e-5)
    for n, a, b_ in zip(names, *[r_[1] for r_ in res]):
        if K == 1 and n in ('w_ih', 'w_hh', 'b_ih', 'b_hh', 'w_lin', 'b_lin'):
            assert b_ is None or float(b_.abs().max()) == 0.0
            continue
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b_).abs().max()) <= 2e-5 * scale + 1e-6, (n, float((a - b_).abs().max()), scale)
