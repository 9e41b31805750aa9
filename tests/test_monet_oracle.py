"""Pins oracle/monet_oracle.py (BASELINE config 4) against golden vectors captured from the real reference MONet,
and live against the import when /root/reference is present."""
import json
import os.path as osp

import numpy as np
import pytest
import torch

from genesis_amd import testing as T
from oracle import monet_oracle as M
from oracle import ref_import as R

GOLDEN = osp.join(osp.dirname(osp.abspath(__file__)), 'golden')
CASES = ['tiny', 'tiny_k4', 'cfg4', 'tiny_scope']


class MonetGolden(object):
    def __init__(self, name):
        self.name = name
        self.g = np.load(osp.join(GOLDEN, 'monet_%s.npz' % name), allow_pickle=False)
        self.cfg = json.loads(str(self.g['cfg_json']))
        self.B, self.K, self.S, self.L = int(self.g['B']), self.cfg['K_steps'], self.cfg['img_size'], self.cfg['comp_ldim']

    def inputs(self, offset=0):
        x = T.make_input(int(self.g['x_seed']), self.B, self.S)
        state = torch.get_rng_state()
        torch.manual_seed(int(self.g['noise_seed']) + offset)
        eps = torch.normal(torch.zeros(self.K * self.B, self.L), torch.ones(self.K * self.B, self.L))
        torch.set_rng_state(state)
        if offset == 0:
            T.check_summary('in/x', x, self.g, 0, 0, self.name)
            T.check_summary('in/eps', eps, self.g, 0, 0, self.name)
        return x, eps

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

    def check_forward(self, recon, losses, stats, comp, rtol=1e-4, atol=2e-5, mask_atol=None):
        st = lambda l: torch.stack(list(l))  # noqa: E731
        mask_atol = atol if mask_atol is None else mask_atol
        self.check('err', losses['err'], rtol, atol)
        self.check('kl_m', losses['kl_m'], rtol, 50 * atol)
        self.check('kl_l_k', st(losses['kl_l_k']), rtol, 10 * atol)
        self.check('recon', recon, rtol, atol)
        self.check('log_m_k', st(stats['log_m_k']), rtol, mask_atol)
        self.check('log_s_k', st(stats['log_s_k']), rtol, mask_atol)
        self.check('x_r_k', st(stats['x_r_k']), rtol, atol)
        self.check('log_m_r_k', st(stats['log_m_r_k']), rtol, mask_atol)
        self.check('mu_k', st(comp['mu_k']), rtol, atol)
        self.check('sigma_k', st(comp['sigma_k']), rtol, atol)
        self.check('z_k', st(comp['z_k']), rtol, atol)

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


@pytest.mark.parametrize('case', CASES)
def test_forward_and_grads(case):
    gold = MonetGolden(case)
    sd = gold.weights(M.template_state_dict(gold.cfg))
    p = {k: v.clone().requires_grad_(k != 'std') for k, v in sd.items()}
    x, eps = gold.inputs()
    recon, losses, stats, _, comp = M.monet_forward(p, x, gold.cfg, eps)
    gold.check_forward(recon, losses, stats, comp, rtol=2e-5, atol=2e-6)
    err, kl_l, kl_m = M.aggregate_losses(losses)
    assert abs(float(err) - float(gold.g['loss/err'])) <= 1e-5 * abs(float(err))
    (err + kl_l + kl_m).backward()
    gold.check_grads([(k, v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if k != 'std'])
    for key in ('log_m_k', 'log_m_r_k'):     # utils/misc.py:258-270
        assert float((torch.stack(stats[key], 4).exp().sum(4) - 1).abs().max()) < 1e-3


@pytest.mark.skipif(not R.reference_available(), reason='reference tree not present')
def test_live_against_reference_import():
    mods = R.import_reference()
    cfg = M.make_cfg(K_steps=4, img_size=32)
    torch.manual_seed(2)
    ref = mods['monet_config'].load(R.reference_cfg(**cfg))
    sd = ref.state_dict()
    want = M.param_shapes(cfg)
    assert list(sd.keys()) == list(want.keys())
    for k, (shape, _) in want.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    x = torch.rand(2, 3, 32, 32)
    torch.manual_seed(5)
    r = ref(x)
    torch.manual_seed(5)
    o = M.monet_forward({k: v for k, v in sd.items()}, x, cfg)
    assert torch.allclose(r[0], o[0], atol=1e-6)
    assert torch.allclose(r[1]['err'], o[1]['err'], rtol=1e-6)
    assert torch.allclose(r[1]['kl_m'], o[1]['kl_m'], rtol=1e-5, atol=1e-5)
