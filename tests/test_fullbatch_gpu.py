"""The chip-filling dispatch against the REAL reference at the batch sizes the benchmark runs.

tests/golden/full_*.npz (tests/golden/make_golden_fullbatch.py, generated from the imported reference): GENESIS-V2 metric
configuration B = 32, config 2 (K = 5) B = 64, config 5 (K = 11, 128 x 128) B = 4, GENESIS config 3 B = 32 (BatchNorm over
the K x B = 224 rows: no per-image decomposition exists), MONet config 4 at one rank's B = 32 -- on the models' own
seed-0 initialisation.  Nothing is forced here: the library's default policies pick the kernels (Winograd convs, bf16-pipe
transposed convs / canvas convs, stream-K weight gradients), and the test asserts that they did.

Bars: forward tensors rtol 1e-4 / atol 2e-5 (log-masks 1e-3 absolute, as in the small-fixture tests); ELBO within 1e-4
relative (north_star: 1e-3; measured 1e-8 .. 2e-7); parameter gradients per parameter in relative L2 on the fixture's
strided samples: |HIP - reference| <= GRAD_FACTOR x budget + GRAD_FLOOR + RELU_FLIP, where `budget` is the fixture's
measured distance of the reference's OWN fp32 gradient from the fp64 gradient (same weights, inputs, noise): two correct
fp32 implementations differ by about that much.

RELU_FLIP is granted only where it applies (round 5): the fixtures carry the reference's per-layer ReLU pattern (active outputs of
every nn.ReLU, make_golden_fullbatch.py:hook_relus) and the HIP forward reports its own per layer (hip_ops.RELU_PROBE: the kernels'
own decision expression, gx_gn_relu_active_count; exact counts of the materialised activations elsewhere).  A parameter gets the
allowance only if some layer DOWNSTREAM of it (a layer its gradient is back-propagated through) has a different count; every other
parameter -- and every parameter of a model without ReLUs (GENESIS: ELU and sigmoid gates) -- is held to GRAD_FACTOR x budget +
GRAD_FLOOR.  The test prints the differing layers and the worst error / bar either way.

RELU_FLIP.  A step at these sizes evaluates ~1e8 ReLUs; a pre-activation within fp32 round-off of zero (|y| ~ 1e-7: a
handful per step) may fall on either side in two correct fp32 evaluation orders, and ONE such decision moves every upstream
gradient by ~1e-3 relative (measured, tools/diag_unet_masks.py: the UNet backward at B = 32 sits 1.6e-3 from the fp64
gradient with ONE differing decision at |y| = 1.06e-7 in `up.3`, and 9.7e-7 from the fp64 gradient evaluated on the HIP
forward's own ReLU pattern; B = 64: two decisions, 6.8e-4 / 1.3e-6).  The arithmetic itself is pinned without that
allowance by tests/test_error_budget_gpu.py::test_unet_gradients_equal_fp64_on_the_same_relu_pattern."""
import json
import os.path as osp

import numpy as np
import pytest
import torch

from genesis_amd import testing as T

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLDEN = osp.join(osp.dirname(osp.abspath(__file__)), 'golden')
CASES = ['v2_metric_b32', 'v2_cfg2_b64', 'v2_cfg5_b4', 'v2_cfg5_b32', 'genesis_cfg3_b32', 'monet_cfg4_b32', 'v2_metric_b32_rect']
GRAD_FACTOR = 5.0      # |HIP - reference| <= |HIP - fp64| + |reference - fp64| <= (4 + 1) x budget: the fp64 error-budget tests' own bar for the
                       # HIP path is 4 x the CPU fp32 error (tests/test_error_budget_gpu.py); measured without any ReLU allowance: GENESIS
                       # (no ReLU in the model) 3.5 x on one BatchNorm bias of 32 values, everything else <= 2.4 x
GRAD_FLOOR = 5e-5
RELU_FLIP = 3e-3
# kernels of the chip-filling dispatch that must have run (profiling rows) per case
EXPECT_KERNELS = {
    'v2_metric_b32': ('wino_conv_kernel', 'kq_dth_kernel', 'kq_dgh_kernel', 'wgq_stream_kernel'),
    'v2_cfg2_b64': ('wino_conv_kernel', 'kq_dth_kernel', 'kq_dgh_kernel', 'wgq_stream_kernel'),
    'v2_cfg5_b4': ('wgq_stream_kernel',),
    'v2_cfg5_b32': ('wino_conv_kernel', 'kq_dth_kernel', 'kq_dgh_kernel', 'wgq_stream_kernel'),     # BASELINE config 5 at its per-GPU batch
    'genesis_cfg3_b32': ('kq_c3h_kernel', 'wgq_stream_kernel'),
    'monet_cfg4_b32': ('kq_c3h_kernel', 'wgq_stream_kernel'),
    'v2_metric_b32_rect': ('wino_conv_kernel', 'kq_dth_kernel', 'kq_dgh_kernel', 'wgq_stream_kernel'),  # SURVEY 8(d)'s structured inputs
}


# forward order of the parameter groups per family: a ReLU decision of a layer enters the gradient of every parameter in front
# of it (and of the layer itself).  MONet's attention UNet is recurrent with shared weights: a decision in any pass touches all of it.
STAGES = {
    'v2': ('encoder.down.', 'encoder.mlp.', 'encoder.up.', 'seg_head.', 'att_process.', 'feat_head.', 'z_head.', 'decoder_module.'),
    'monet': ('att_process.', 'comp_vae.encoder_module.module.', 'comp_vae.decoder_module.seq.'),
    'genesis': (),
}


def _stage_index(fam, name):
    """(stage number, layer index inside the stage) of a parameter / ReLU-site name."""
    for i, pre in enumerate(STAGES[fam]):
        if name.startswith(pre):
            head = name[len(pre):].split('.')[0]
            return i, (int(head) if head.isdigit() else 0)
    return None


def upstream_of(fam, site, pname):
    """Does the gradient of parameter `pname` pass through the ReLU of `site`?"""
    s_, p_ = _stage_index(fam, site), _stage_index(fam, pname)
    if s_ is None or p_ is None:
        return False                              # (the AR prior, the output conv behind the last ReLU: behind every ReLU)
    if fam == 'monet' and s_[0] == 0:
        return p_[0] == 0                         # recurrent UNet, shared weights
    if fam == 'v2' and site.startswith('feat_head.') and (pname.startswith('seg_head.') or pname.startswith('att_process.')):
        return False                              # parallel heads on the encoder features
    return p_[0] < s_[0] or (p_[0] == s_[0] and p_[1] <= s_[1])


def relu_sites_of_run(model, probe):
    """{site: [active, outputs]} of a HIP forward from hip_ops.RELU_PROBE records (keyed by the layer's affine / bias parameter)."""
    by_ptr = {p.data_ptr(): n for n, p in model.named_parameters()}
    sites = {}
    for ptr, cnt, numel in probe:
        name = by_ptr[ptr]
        site = name.rsplit('.', 1)[0]
        a = sites.setdefault(site, [0, 0])
        a[0] += int(cnt.item())
        a[1] += int(numel)
    return sites


def st(l):
    return torch.stack(list(l))


def replay(seed, shapes):
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    out = [torch.normal(torch.zeros(*s), torch.ones(*s)) for s in shapes]
    torch.set_rng_state(state)
    return out


class Full(object):
    def __init__(self, name):
        self.name = name
        self.g = np.load(osp.join(GOLDEN, 'full_%s.npz' % name), allow_pickle=False)
        self.cfg = json.loads(str(self.g['cfg_json']))
        self.fam = str(self.g['family'])
        self.B, self.K, self.S = int(self.g['B']), self.cfg['K_steps'], self.cfg['img_size']
        self.nseed = int(self.g['noise_seed'])

    def build(self):
        from genesis_amd.compat.attrdict import AttrDict
        if self.fam == 'v2':
            import genesis_amd.genesisv2_config as G
            cfg = dict(dict(dynamic_K=False), **self.cfg)
        elif self.fam == 'genesis':
            import genesis_amd.genesis_config as G
            cfg = self.cfg
        else:
            import genesis_amd.monet_config as G
            cfg = self.cfg
        torch.manual_seed(0)
        model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False)))
        if self.fam == 'v2':
            with torch.no_grad():
                model.att_process.colour_head.gate.gate.fill_(float(self.g['v2_gate']))
        # the fixture's weights ARE the seed-0 initialisation: prove this process rebuilt them
        sd = model.state_dict()
        assert list(sd.keys()) == [str(k) for k in self.g['sd_keys']]
        assert [int(v.numel()) for v in sd.values()] == [int(n) for n in self.g['sd_numel']]
        np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], self.g['sd_sum'], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], self.g['sd_asum'], rtol=1e-12, atol=1e-12)
        return model.to(DEV).train()

    def x(self):
        kind = str(self.g['input_kind']) if 'input_kind' in self.g.files else 'rand'
        x = T.make_input_of(kind, int(self.g['x_seed']), self.B, self.S)
        T.check_summary('in/x', x, self.g, 0, 0, self.name)
        return x

    def noise(self, offset=0, check=False):
        K, B, S = self.K, self.B, self.S
        if self.fam == 'v2':
            rp, eps = T.draw_noise(self.nseed + offset, B, S, self.cfg['feat_dim'], K)
            nz = [rp] + list(eps)
        elif self.fam == 'genesis':
            nz = replay(self.nseed + offset, [(B, self.cfg['attention_latents'])] * K + [(K * B, self.cfg['comp_ldim'])])
        else:
            nz = replay(self.nseed + offset, [(K * B, self.cfg['comp_ldim'])])
        if check:
            for i, n in enumerate(nz):
                T.check_summary('in/noise%d' % i, n, self.g, 0, 0, self.name)
        return nz

    def forward_kwargs(self, nz):
        d = lambda t: t.to(DEV)   # noqa: E731
        if self.fam == 'v2':
            return dict(rand_pixel=d(nz[0]), eps=torch.stack(nz[1:]).to(DEV))
        if self.fam == 'genesis':
            return dict(eps_m=[d(n) for n in nz[:self.K]], eps_c=d(nz[self.K]))
        return dict(eps=d(nz[0]))

    def forward(self, model, x, nz, seed_idx=None):
        d = lambda t: t.to(DEV)   # noqa: E731
        if self.fam == 'v2':
            return model(d(x), d(nz[0]), torch.stack(nz[1:]).to(DEV), seed_idx)
        if self.fam == 'genesis':
            return model(d(x), [d(n) for n in nz[:self.K]], d(nz[self.K]))
        return model(d(x), d(nz[0]))

    def aggregate(self, l):
        e = l['err'].mean(0)
        kl = 0.0
        for key in ('kl_l_k', 'kl_m_k'):
            if key in l:
                kl = kl + torch.stack(list(l[key]), dim=1).mean(0).sum()
        if 'kl_m' in l:
            kl = kl + l['kl_m'].mean(0)
        return e, kl

    def named(self, out):
        recon, losses, stats, att, comp = out
        n = {'err': losses['err'], 'recon': recon, 'log_m_k': st(stats['log_m_k']), 'x_r_k': st(stats['x_r_k'])}
        if self.fam == 'v2':
            n.update(kl_l_k=st(losses['kl_l_k']), log_s_k=st(stats['log_s_k']), log_m_r_k=st(stats['log_m_r_k']),
                     colour=att['colour'], seeds=st(att['seeds']), mu_k=st(comp['mu_k']), sigma_k=st(comp['sigma_k']),
                     z_k=st(comp['z_k']))
        elif self.fam == 'genesis':
            n.update(kl_m_k=st(losses['kl_m_k']), kl_l_k=st(losses['kl_l_k']), att_mu_k=st(att['mu_k']),
                     att_z_k=st(att['z_k']), comp_mu_k=st(comp['mu_k']), comp_sigma_k=st(comp['sigma_k']),
                     comp_z_k=st(comp['z_k']))
        else:
            n.update(kl_m=losses['kl_m'], kl_l_k=st(losses['kl_l_k']), log_s_k=st(stats['log_s_k']),
                     log_m_r_k=st(stats['log_m_r_k']), mu_k=st(comp['mu_k']), sigma_k=st(comp['sigma_k']),
                     z_k=st(comp['z_k']))
        return n

    def check(self, key, tensor, rtol, atol):
        full = 'out/' + key
        if full in self.g.files:
            np.testing.assert_allclose(tensor.detach().cpu().float().numpy(), self.g[full], rtol=rtol, atol=atol,
                                       err_msg='%s %s' % (self.name, key))
        else:
            T.check_summary(full, tensor, self.g, rtol, atol, self.name)


# forward tolerances per family (rtol, atol): the small-fixture tests' bars (tests/test_model_gpu.py, test_genesis_gpu.py,
# test_monet_gpu.py); the Monte-Carlo KL terms of GENESIS / MONet are differences of log-densities of size ~1e2
FWD_TOL = {
    'v2': {'*': (1e-4, 2e-5), 'log_m_k': (1e-4, 1e-3), 'log_s_k': (1e-4, 1e-3), 'kl_l_k': (1e-4, 2e-4)},
    'genesis': {'*': (1e-4, 5e-5), 'err': (1e-4, 1e-3), 'kl_m_k': (1e-3, 5e-3), 'kl_l_k': (1e-3, 5e-3), 'log_m_k': (1e-4, 1e-3)},
    'monet': {'*': (1e-4, 2e-5), 'log_m_k': (1e-4, 1e-3), 'log_s_k': (1e-4, 1e-3), 'kl_l_k': (1e-4, 2e-4), 'kl_m': (1e-4, 1e-3),
              'err': (1e-4, 1e-3)},
}


def check_gradients(gold, grads, differing, case):
    """Per-parameter gradients against the reference's at the fixture's own budget: |HIP - reference| <= (4 + 1) x budget + 5e-5
    in relative L2 on the strided samples and on the norm, + the ReLU allowance ONLY for parameters in front of a layer whose
    ReLU pattern differs (`differing`: [(site, count difference)])."""
    names = [str(n) for n in gold.g['param_names']]
    norms, budget = gold.g['grad_norms'], gold.g['budget']
    gmax = float(gold.g['grad_max_f64'])
    worst, table = 0.0, []
    for i, name in enumerate(names):
        g = grads[name]
        s = T.summarize(g)
        ref = gold.g['grad/%s/samples' % name].astype(np.float64)
        assert int(gold.g['grad/%s/n' % name]) == int(s['n']), name
        # the budget is relative to max(|g64|, 1e-6 gmax) over the whole tensor; the same scale for its strided samples
        den = max(float(gold.g['grad_norms_f64'][i]), 1e-6 * gmax) * np.sqrt(len(ref) / max(1, int(s['n'])))
        e_samples = float(np.linalg.norm(s['samples'].astype(np.float64) - ref)) / den
        e_norm = abs(float(g.double().norm()) - float(norms[i])) / max(float(gold.g['grad_norms_f64'][i]), 1e-6 * gmax)
        flip = any(upstream_of(gold.fam, site, name) for site, _ in differing)
        bar = GRAD_FACTOR * float(budget[i]) + GRAD_FLOOR + (RELU_FLIP if flip else 0.0)
        table.append((max(e_samples, e_norm) / bar, name, e_samples, e_norm, float(budget[i]), flip))
        worst = max(worst, max(e_samples, e_norm) / bar)
    table.sort(reverse=True)
    print('%s: worst gradient error / bar = %.3f; the five largest (samples rel-L2, norm rel, budget):' % (case, worst))
    for r in table[:5]:
        print('   %-52s %.2e %.2e  budget %.2e  (%.2f of the bar%s)' % (r[1], r[2], r[3], r[4], r[0], ', ReLU allowance' if r[5] else ''))
    assert worst <= 1.0, table[0]
    return worst


def test_every_eligible_conv3x3_on_the_winograd_kernel_vs_reference():
    """The Winograd kernel on EVERY layer it supports (policy 2: also the 16 x 16 ... 64 x 64 levels that the default policy leaves
    to the direct kernels at this batch) of the 128 x 128 configuration, against the reference's full-batch fixture at the same
    bars as the default dispatch.  (The closed-form weights of tests/golden/v2_cfg5.npz are too ill-conditioned for this check:
    the reference's own fp32 gradient sits 8e-3 from fp64 there and a changed rounding moves `encoder.down.5.0.weight` by 2 %.)"""
    from genesis_amd import _lib
    _lib.call('gx_conv3x3_wino_policy', 2)
    try:
        test_default_dispatch_vs_reference_at_benchmark_batch('v2_cfg5_b4', min_wino=16)      # (default policy at this batch: 8)
    finally:
        _lib.call('gx_conv3x3_wino_policy', 1)


@pytest.mark.parametrize('case', CASES)
def test_default_dispatch_vs_reference_at_benchmark_batch(case, min_wino=0):
    from genesis_amd import profiling
    gold = Full(case)
    model = gold.build()
    x, nz = gold.x(), gold.noise(check=True)
    from genesis_amd import hip_ops
    profiling.enable(True)
    hip_ops.RELU_PROBE = probe = []
    try:
        out = gold.forward(model, x, nz)
        flips = 0
        if gold.fam == 'v2':
            # (K-1) x B argmax decisions among S^2 candidates: the fixture's smallest top-2 margin is ~1e-5 relative, so a
            # decision may legitimately fall the other way in another fp32 evaluation order -- only where the reference's
            # own margin is below 1e-4, and the run is then repeated on the reference's seed pixels (the arithmetic is what
            # this test pins; the argmax itself is pinned by tests/test_model_gpu.py on fixtures with margins > 2e-5)
            seed_idx = torch.stack(list(out[3]['seed_idx'])).cpu().numpy()
            ref_idx = gold.g['seed_idx']
            bad = seed_idx != ref_idx
            flips = int(bad.sum())
            print('%s: seed pixels differing from the reference: %d of %d' % (case, flips, bad.size))
            if flips:
                assert flips <= 2 and float(gold.g['seed_margin'][bad].max()) < 1e-4, (flips, gold.g['seed_margin'][bad])
                del probe[:]
                out = gold.forward(model, x, nz, torch.from_numpy(ref_idx).to(DEV))
        hip_ops.RELU_PROBE = None
        recon, losses, stats, att, comp = out
        tol = FWD_TOL[gold.fam]
        for key, t in gold.named(out).items():
            rtol, atol = tol.get(key, tol['*'])
            gold.check(key, t, rtol, atol)
        if gold.fam == 'v2':
            assert int(stats['instance_seg'].sum().item()) == int(gold.g['instance_seg_sum'])
        err, kl = gold.aggregate(losses)
        elbo_ref = float(gold.g['loss/err']) + float(gold.g['loss/kl'])
        rel = abs(float(err + kl) - elbo_ref) / abs(elbo_ref)
        print('%s: ELBO %.4f (reference %.4f, fp64 %.4f): rel %.2e' % (case, float(err + kl), elbo_ref, float(gold.g['loss/elbo_f64']), rel))
        assert rel <= 1e-4
        (err + kl).backward()
        rows = {r['name']: r['launches'] for r in profiling.collect()}
    finally:
        profiling.enable(False)
        hip_ops.RELU_PROBE = None
    # --- the ReLU pattern per layer: HIP forward against the reference's
    ref_sites = {str(k): (int(a), int(n)) for k, a, n in zip(gold.g['relu_sites'], gold.g['relu_active'], gold.g['relu_outputs']) if str(k)}
    got_sites = relu_sites_of_run(model, probe)
    assert set(got_sites) == set(ref_sites), (sorted(set(got_sites) ^ set(ref_sites)))
    differing = []
    for site, (ra, rn) in sorted(ref_sites.items()):
        ha, hn = got_sites[site]
        # (feat_head runs K times in the reference and once here: compare active / outputs as exact fractions)
        assert rn % hn == 0 or hn % rn == 0, (site, rn, hn)
        if ra * hn != ha * rn:
            differing.append((site, ha * rn / hn - ra))
    print('%s: layers whose ReLU pattern differs from the reference\'s (HIP - reference active outputs): %s'
          % (case, ', '.join('%s %+g' % d for d in differing) or 'none of %d' % len(ref_sites)))
    for kname in EXPECT_KERNELS[case]:
        assert any(k.startswith(kname) and v > 0 for k, v in rows.items()), (kname, sorted(rows))
    assert rows.get('wino_conv_kernel', 0) >= min_wino, rows
    # parameter gradients
    named = dict(model.named_parameters())
    check_gradients(gold, {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in named.items()}, differing, case)
    for key in ('log_m_k',) + (('log_m_r_k',) if gold.fam != 'genesis' else ()):
        assert float((torch.stack(list(stats[key]), 4).exp().sum(4) - 1).abs().max()) < 1e-3      # utils/misc.py:258-270


@pytest.mark.parametrize('case', CASES)
def test_three_training_steps_at_benchmark_batch(case):
    """train.py:223-263 -- forward, aggregation, GECO, backward, Adam -- three iterations against the reference's own history
    (ELBO, err, KL, beta used, err_ema) at the benchmark's batch; BatchNorm running statistics included for GENESIS."""
    from genesis_amd.trainer import TrainStep
    gold = Full(case)
    model = gold.build()
    ts = TrainStep(model, gold.S, lr=1e-4)
    x = gold.x().to(DEV)
    hist = gold.g['train_hist']
    try:
        for it in range(3):
            nz = gold.noise(1 + it)
            out = ts.step(x, **gold.forward_kwargs(nz)).cpu().numpy()
            elbo, err, kl, beta = [float(v) for v in out]
            # step 0 is a forward comparison; later ELBOs inherit the first Adam updates (sign-like: lr x g / |g|)
            tol = 1e-4 if it == 0 else 5e-4
            assert abs(elbo - hist[it, 0]) <= tol * abs(hist[it, 0]), (case, it, out, hist[it])
            np.testing.assert_allclose([err, beta], hist[it, [1, 3]], rtol=5e-4)
        assert abs(float(ts.geco.beta) - float(gold.g['train_beta_final'])) <= 1e-5 * float(gold.g['train_beta_final'])
    finally:
        ts.close()
