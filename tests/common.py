"""Shared helpers for the parity tests: load a golden fixture, rebuild its closed-form
weights / inputs / noise, and compare a forward result against it."""
import json
import os.path as osp

import numpy as np
import torch

from genesis_amd import testing as T

GOLDEN = osp.join(osp.dirname(osp.abspath(__file__)), 'golden')
ALL_CASES = ['tiny', 'tiny_b3k3', 'tiny_noar', 'tiny_klm', 'tiny_klm_nodetach', 'tiny_nosemi', 'tiny_laplacian',
             'tiny_epanechnikov', 'metric', 'cfg2', 'cfg5']


class Golden(object):
    def __init__(self, name):
        self.name = name
        self.g = np.load(osp.join(GOLDEN, 'v2_%s.npz' % name), allow_pickle=False)
        self.cfg = json.loads(str(self.g['cfg_json']))
        self.cfg['pixel_std2'] = self.cfg['pixel_std1']
        self.B = int(self.g['B'])
        self.K = self.cfg['K_steps']
        self.S = self.cfg['img_size']
        self.D = self.cfg['feat_dim']

    def inputs(self):
        x = T.make_input(int(self.g['x_seed']), self.B, self.S)
        rand_pixel, eps_k = T.draw_noise(int(self.g['noise_seed']), self.B, self.S, self.D, self.K)
        # the stored summaries pin that the regenerated inputs are the reference's
        T.check_summary('in/x', x, self.g, 0, 0, self.name)
        T.check_summary('in/rand_pixel', rand_pixel, self.g, 0, 0, self.name)
        T.check_summary('in/eps', torch.stack(eps_k), self.g, 0, 0, self.name)
        return x, rand_pixel, eps_k

    def noise(self, seed_offset):
        return T.draw_noise(int(self.g['noise_seed']) + seed_offset, self.B, self.S, self.D, self.K)

    def weights(self, template):
        keys = [str(k) for k in self.g['sd_keys']]
        assert list(template.keys()) == keys, 'state_dict key order differs from reference'
        numel = [int(v.numel()) for v in template.values()]
        assert numel == [int(n) for n in self.g['sd_numel']]
        return T.formula_state_dict(template)

    def check(self, key, tensor, rtol, atol):
        full = 'out/' + key
        if full in self.g.files:
            np.testing.assert_allclose(tensor.detach().cpu().float().numpy(), self.g[full],
                                       rtol=rtol, atol=atol, err_msg='%s %s' % (self.name, key))
        else:
            T.check_summary(full, tensor, self.g, rtol, atol, self.name)

    def has(self, key):
        return ('out/' + key) in self.g.files or ('out/%s/n' % key) in self.g.files

    def check_forward(self, recon, losses, stats, att, comp, rtol=1e-4, atol=1e-5, mask_atol=None):
        mask_atol = atol if mask_atol is None else mask_atol
        st = lambda l: torch.stack(list(l))  # noqa: E731
        if att is None:                      # dynamic_K on a batch (genesisv2_config.py:122): no att_stats, no log_s_k
            assert not self.has('colour') and not self.has('log_s_k') and stats['log_s_k'] is None
        elif 'seed_idx' in self.g.files:
            seed_idx = torch.stack([i.cpu() for i in att['seed_idx']]).numpy()
            np.testing.assert_array_equal(seed_idx, self.g['seed_idx'], err_msg='seed pixels')
        if 'slots' in self.g.files:          # dynamic_K: the number of mask tensors the reference returned
            assert len(stats['log_m_k']) == int(self.g['slots'])
        self.check('err', losses['err'], rtol, atol)
        self.check('kl_l_k', st(losses['kl_l_k']), rtol, 10 * atol)
        if 'kl_m' in losses:
            self.check('kl_m', losses['kl_m'], rtol, atol)
        self.check('recon', recon, rtol, atol)
        self.check('log_m_k', st(stats['log_m_k']), rtol, mask_atol)
        if self.has('log_s_k'):
            self.check('log_s_k', st(stats['log_s_k']), rtol, mask_atol)
        self.check('x_r_k', st(stats['x_r_k']), rtol, atol)
        self.check('log_m_r_k', st(stats['log_m_r_k']), rtol, atol)
        if self.has('colour'):
            self.check('colour', att['colour'], rtol, atol)
            self.check('seeds', st(att['seeds']), rtol, atol)
        self.check('mu_k', st(comp['mu_k']), rtol, atol)
        self.check('sigma_k', st(comp['sigma_k']), rtol, atol)
        self.check('z_k', st(comp['z_k']), rtol, atol)
        assert int(stats['instance_seg'].sum().item()) == int(self.g['instance_seg_sum'])

    def check_grads(self, named_grads, rtol=2e-3, l2_tol=1e-2):
        """named_grads: iterable of (name, grad tensor) in state_dict order.

        Gradients are compared in relative L2 (norm and strided samples), not element-wise: the ReLU
        derivative is discontinuous, so one borderline pre-activation (|pre| ~ 1e-7, a handful among
        the ~1e7 activations of a step) that flips sign between two correct fp32 implementations
        perturbs every upstream gradient by ~1/sqrt(n) ~ 1e-4..1e-3 relative (measured on the GPU:
        tools/diag_model_decoder.py; the CPU fp32 path shows the same effect against fp64).  The
        per-kernel tests pin the arithmetic itself at 1e-5..1e-4."""
        norms = self.g['grad_norms']
        big = float(np.max(norms))  # floor for analytically-zero grads (pure cancellation noise)
        for i, (name, g) in enumerate(named_grads):
            got = float(g.double().norm().item())
            assert abs(got - float(norms[i])) <= rtol * float(norms[i]) + 2e-5 + 1e-6 * big, \
                '%s grad norm %s: %r vs %r' % (self.name, name, got, float(norms[i]))
            s = T.summarize(g)
            ref = self.g['grad/%s/samples' % name].astype(np.float64)
            assert int(self.g['grad/%s/n' % name]) == int(s['n']), name
            diff = np.linalg.norm(s['samples'].astype(np.float64) - ref)
            floor = (2e-5 + 1e-6 * big) * np.sqrt(len(ref) / max(1, int(s['n'])))
            assert diff <= l2_tol * np.linalg.norm(ref) + floor, \
                '%s grad %s: sample rel-L2 %.3e' % (self.name, name, diff / (np.linalg.norm(ref) + 1e-30))


SAMPLE_CASES = ['tiny', 'tiny_k6', 'tiny_noar', 'metric']
DYN_CASES = ['tiny_dynk', 'tiny_dynk_b1']          # dynamic_K: a batch (padded slots) and one image (fewer slots)


class SampleGolden(object):
    """tests/golden/v2_sample_*.npz: GenesisV2.sample of the real reference with its standard-normal draws recorded
    (tests/golden/make_golden_sample.py)."""

    def __init__(self, name):
        self.name = name
        self.g = np.load(osp.join(GOLDEN, 'v2_sample_%s.npz' % name), allow_pickle=False)
        self.cfg = json.loads(str(self.g['cfg_json']))
        self.cfg['pixel_std2'] = self.cfg['pixel_std1']
        self.B = int(self.g['B'])
        self.K_arg = None if int(self.g['K_arg']) < 0 else int(self.g['K_arg'])
        self.eps = torch.from_numpy(self.g['eps'])

    def check(self, key, tensor, rtol, atol):
        full = 'out/' + key
        if full in self.g.files:
            np.testing.assert_allclose(tensor.detach().cpu().float().numpy(), self.g[full], rtol=rtol, atol=atol,
                                       err_msg='sample %s %s' % (self.name, key))
        else:
            T.check_summary(full, tensor, self.g, rtol, atol, 'sample ' + self.name)

    def check_all(self, recon, x_k, log_m_k, z_k, rtol, atol, mx_k=None):
        st = lambda l: torch.stack(list(l))  # noqa: E731
        self.check('z_k', st(z_k), rtol, atol)
        self.check('recon', recon, rtol, atol)
        self.check('x_k', st(x_k), rtol, atol)
        self.check('log_m_k', st(log_m_k), rtol, 10 * atol)
        if mx_k is not None:
            self.check('mx_k', st(mx_k), rtol, atol)


def fp32_budget(loss_fn, sd, is_param=lambda k: True):
    """Per-parameter error of plain fp32 arithmetic on this problem: loss_fn(p) -> scalar, evaluated by the CPU oracle in
    fp64 (ground truth) and in fp32 (what the reference computes); returns {name: relative L2 error of the fp32 gradient}.
    The golden gradient comparisons derive their tolerances from it: |HIP - reference fp32| <= |HIP - fp64| + |reference -
    fp64| <= (4 + 1) x this + floor (the HIP bar of tests/test_error_budget_gpu.py plus the reference's own error)."""
    grads = {}
    for dtype in (torch.float64, torch.float32):
        p = {k: (v.clone().to(dtype if v.dtype == torch.float32 else v.dtype).requires_grad_(True) if is_param(k) and v.is_floating_point()
                 else v.clone().to(dtype if v.dtype == torch.float32 else v.dtype)) for k, v in sd.items()}
        loss_fn(p, dtype).backward()
        grads[dtype] = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in p.items() if v.requires_grad}
    g64, g32 = grads[torch.float64], grads[torch.float32]
    return {k: float((g32[k] - g64[k]).norm()) / (float(g64[k].norm()) + 1e-30) for k in g64}


def budget_tolerances(e_cpu, floor=2e-4, factor=5.0, sampling=1.5, cap=None):
    """{name: tolerance} for a golden gradient comparison on strided samples: sampling x (factor x e_cpu) + floor, never
    looser than `cap`.  floor: what one ReLU whose pre-activation sits within fp32 round-off of zero on the fixtures'
    closed-form weights moves an upstream gradient by when it falls the other way (1e-4 .. 1e-3, Golden.check_grads)."""
    return {k: min(sampling * factor * v + floor, cap if cap is not None else float('inf')) for k, v in e_cpu.items()}
