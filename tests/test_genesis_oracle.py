"""Pins oracle/vae_oracle.py (BASELINE config 1) and oracle/genesis_oracle.py (config 3) against golden vectors
captured from the real reference (tests/golden/make_golden_genesis.py)."""
import json
import os.path as osp

import numpy as np
import pytest
import torch

from genesis_amd import testing as T
from oracle import genesis_oracle as GO
from oracle import vae_oracle as VO

GOLDEN = osp.join(osp.dirname(osp.abspath(__file__)), 'golden')
VAE_CASES = ['tiny', 'cfg1', 'tiny_bcast', 'cfg1_b32']
GEN_CASES = ['tiny', 'tiny_in', 'cfg3', 'tiny_noprior', 'tiny_onestage', 'tiny_sym']


class Gold(object):
    def __init__(self, prefix, name):
        self.name = '%s_%s' % (prefix, name)
        self.g = np.load(osp.join(GOLDEN, self.name + '.npz'), allow_pickle=False)
        self.cfg = json.loads(str(self.g['cfg_json']))
        self.B, self.S = int(self.g['B']), self.cfg['img_size']

    def x(self):
        x = T.make_input(int(self.g['x_seed']), self.B, self.S)
        T.check_summary('in/x', x, self.g, 0, 0, self.name)
        return x

    def replay(self, shapes, offset=0):
        state = torch.get_rng_state()
        torch.manual_seed(int(self.g['noise_seed']) + offset)
        out = [torch.normal(torch.zeros(*s), torch.ones(*s)) for s in shapes]
        torch.set_rng_state(state)
        return out

    def weights(self, template):
        assert list(template.keys()) == [str(k) for k in self.g['sd_keys']]
        assert [int(v.numel()) for v in template.values()] == [int(n) for n in self.g['sd_numel']]
        return T.formula_state_dict(template)

    def check(self, key, tensor, rtol, atol):
        full = 'out/' + key
        if full in self.g.files:
            np.testing.assert_allclose(tensor.detach().cpu().float().numpy(), self.g[full], rtol=rtol, atol=atol,
                                       err_msg='%s %s' % (self.name, key))
        else:
            T.check_summary(full, tensor, self.g, rtol, atol, self.name)

    def check_grads(self, named_grads, rtol=2e-3, l2_tol=1e-2, per_param=None):
        """per_param: {name: tolerance} overriding rtol / l2_tol for that parameter (tests.common.budget_tolerances)."""
        names = [str(n) for n in self.g['param_names']]
        norms = self.g['grad_norms']
        big = float(np.max(norms))
        named = dict(named_grads)
        for i, name in enumerate(names):
            g = named[name]
            if per_param is not None:
                rtol = l2_tol = per_param[name]
            got = float(g.double().norm().item())
            assert abs(got - float(norms[i])) <= rtol * float(norms[i]) + 2e-5 + 1e-6 * big, (self.name, name, got, norms[i])
            s = T.summarize(g)
            ref = self.g['grad/%s/samples' % name].astype(np.float64)
            diff = np.linalg.norm(s['samples'].astype(np.float64) - ref)
            floor = (2e-5 + 1e-6 * big) * np.sqrt(len(ref) / max(1, int(s['n'])))
            assert diff <= l2_tol * np.linalg.norm(ref) + floor, (self.name, name, diff / (np.linalg.norm(ref) + 1e-30))


def is_param(k):
    return not (k == 'std' or k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))


@pytest.mark.parametrize('case', VAE_CASES)
def test_vae_forward_and_grads(case):
    gold = Gold('vae', case)
    sd = gold.weights(VO.template_state_dict(gold.cfg))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = gold.x()
    (eps,) = gold.replay([(gold.B, gold.cfg['latent_dimension'])])
    recon, losses, stats, _, _ = VO.vae_forward(p, x, gold.cfg, eps)
    for k, t in (('err', losses['err']), ('kl_l', losses['kl_l']), ('recon', recon), ('mu', stats['mu']), ('z', stats['z'])):
        gold.check(k, t, 2e-5, 2e-5)
    (losses['err'].mean(0) + losses['kl_l'].mean(0)).backward()
    gold.check_grads([(k, v.grad) for k, v in p.items()])


@pytest.mark.parametrize('case', GEN_CASES)
def test_genesis_forward_and_grads(case):
    gold = Gold('genesis', case)
    cfg = gold.cfg
    sd = gold.weights(GO.template_state_dict(cfg))
    p = {k: (v.clone().requires_grad_(True) if is_param(k) else v.clone()) for k, v in sd.items()}
    x = gold.x()
    K, L, Lc = cfg['K_steps'], cfg['attention_latents'], cfg['comp_ldim']
    two = cfg.get('two_stage', True)
    noise = gold.replay([(gold.B, L)] * K + ([(K * gold.B, Lc)] if two else []))
    recon, losses, stats, att, comp = GO.genesis_forward(p, x, cfg, noise[:K], noise[K] if two else None)
    st = lambda l: torch.stack(list(l))  # noqa: E731
    gold.check('err', losses['err'], 2e-5, 2e-5)
    gold.check('kl_m_k', st(losses['kl_m_k']), 5e-5, 5e-4)
    if two:
        gold.check('kl_l_k', st(losses['kl_l_k']), 5e-5, 5e-4)
        gold.check('comp_z_k', st(comp['z_k']), 2e-5, 2e-5)
    else:
        assert comp is None and 'kl_l_k' not in losses
    gold.check('recon', recon, 2e-5, 2e-5)
    gold.check('log_m_k', st(stats['log_m_k']), 5e-5, 5e-5)
    gold.check('x_r_k', st(stats['x_r_k']), 2e-5, 2e-5)
    gold.check('att_z_k', st(att['z_k']), 2e-5, 2e-5)
    err, kl_l, kl_m = GO.aggregate_losses(losses)
    (err + kl_l + kl_m).backward()
    gold.check_grads([(k, v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if is_param(k)],
                     rtol=5e-3, l2_tol=2e-2)
    assert float((torch.stack(stats['log_m_k'], 4).exp().sum(4) - 1).abs().max()) < 1e-3
