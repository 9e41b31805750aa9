"""Generates tests/golden/full_*.npz from the REAL reference at the batch sizes the benchmark runs (the chip-filling
dispatch of the HIP path: Winograd convs, bf16-pipe transposed convs, stream-K weight gradients, BatchNorm over the
K x B batch of GENESIS) -- the golden cases of make_golden*.py have B <= 3.

    python tests/golden/make_golden_fullbatch.py             # all five cases (a few minutes of CPU)
    python tests/golden/make_golden_fullbatch.py v2_metric_b32

Weights: the model's OWN initialisation (`torch.manual_seed(0); load(cfg)` -- what bench.py times; the construction
order of genesis_amd's modules reproduces it bit for bit, tests/test_cabi_cpu.py) with, for GENESIS-V2, the SemiConv
gate moved off its zero initial value (0.35: with gate = 0 the seg_head / colour-head gradients vanish identically).
Per-tensor checksums of that state_dict travel in the fixture, so the GPU test proves it rebuilt the same weights.
Unlike the closed-form weights of the small fixtures this is a well-conditioned fp32 problem: the fixture also carries,
per parameter, the distance of the reference's fp32 gradient from the fp64 gradient of the oracle on the same weights,
inputs and noise (`budget/<name>`), from which the GPU test derives its tolerance (no blanket 1e-2).

Stored: input summaries, the reference's forward outputs (sum / abs-sum / 2048 strided samples), per-image err and KL
terms in full, seed pixels + top-2 margins (GENESIS-V2), per-parameter gradient norms + samples, three GECO + Adam
training steps.  The oracle (oracle/*.py) is checked against the reference at these sizes on the way (fp32, 2e-5)."""
import json
import os.path as osp
import sys
import time

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, REPO)

from genesis_amd import testing as T  # noqa: E402
from oracle import ref_import as R  # noqa: E402
from oracle import genesis_oracle as GO  # noqa: E402
from oracle import monet_oracle as MO  # noqa: E402
from oracle import v2_oracle as VO  # noqa: E402

CASES = {
    # name: (family, cfg overrides, B, x seed, noise seed)
    'v2_metric_b32': ('v2', dict(K_steps=7, img_size=64, feat_dim=64), 32, 118, 128),
    'v2_cfg2_b64': ('v2', dict(K_steps=5, img_size=64, feat_dim=64), 64, 119, 129),
    'v2_cfg5_b4': ('v2', dict(K_steps=11, img_size=128, feat_dim=64), 4, 120, 130),
    'v2_cfg5_b32': ('v2', dict(K_steps=11, img_size=128, feat_dim=64), 32, 121, 131),      # config 5 at its per-GPU batch
    'genesis_cfg3_b32': ('genesis', dict(K_steps=7, img_size=64), 32, 155, 165),
    'monet_cfg4_b32': ('monet', dict(K_steps=7, img_size=64), 32, 133, 143),
    # the metric configuration on the STRUCTURED input set (five-level flat-colour regions, testing.make_rect_input)
    'v2_metric_b32_rect': ('v2', dict(K_steps=7, img_size=64, feat_dim=64), 32, 218, 228),
}
INPUT_KIND = {'v2_metric_b32_rect': 'rect'}
V2_GATE = 0.35


def replay(seed, shapes):
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    out = [torch.normal(torch.zeros(*s), torch.ones(*s)) for s in shapes]
    torch.set_rng_state(state)
    return out


def top2_margin(rand_pixel, log_s):
    v = (rand_pixel * log_s.exp()).flatten(1)
    top = v.topk(2, dim=1).values
    return (top[:, 0] - top[:, 1]) / top[:, 0].abs().clamp_min(1e-30)


def st(l):
    return torch.stack(list(l))


def is_param(k):
    return not (k == 'std' or k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))


def hook_relus(model):
    """Per-layer ReLU pattern of the reference's forward: every nn.ReLU inside an nn.Sequential gets its own instance (the
    ComponentVAE hands ONE activation module to all of its layers, modules/component_vae.py:36-47: behaviour unchanged) and a
    forward hook that adds (out > 0).sum() / out.numel() to the site named after the nearest parametrised sibling in front
    of it (`encoder.down.0.1` = the GroupNorm of that block, `z_head.1` = the Linear, `comp_vae.decoder_module.seq.1` = the
    conv).  Returns {site: [active, outputs]}, filled by the forwards that follow (summed over calls: feat_head runs K
    times, the recurrent UNet of MONet K - 1 times)."""
    import torch.nn as nn
    counts = {}
    for sname, seq in list(model.named_modules()):
        if not isinstance(seq, nn.Sequential):
            continue
        for i, child in enumerate(list(seq)):
            if not isinstance(child, nn.ReLU):
                continue
            j = i - 1
            while j >= 0 and not any(True for _ in seq[j].parameters(recurse=False)):
                j -= 1
            assert j >= 0, (sname, i)
            site = '%s.%d' % (sname, j)
            fresh = nn.ReLU(inplace=child.inplace)
            seq[i] = fresh
            counts[site] = [0, 0]

            def hook(mod, inp, out, site=site):
                counts[site][0] += int((out > 0).sum())
                counts[site][1] += out.numel()
            fresh.register_forward_hook(hook)
    return counts


class Family(object):
    """What differs between the three model families: noise shapes, the forward call with replayed noise, the named
    outputs, the aggregation of the loss terms (train.py:226-242) and the oracle entry point."""

    def __init__(self, fam, cfgd, B):
        self.fam, self.cfgd, self.B = fam, cfgd, B
        self.K, self.S = cfgd['K_steps'], cfgd['img_size']

    def noise(self, seed):
        K, B, S = self.K, self.B, self.S
        if self.fam == 'v2':
            rp, eps = T.draw_noise(seed, B, S, self.cfgd['feat_dim'], K)
            return [rp] + list(eps)
        if self.fam == 'genesis':
            return replay(seed, [(B, self.cfgd['attention_latents'])] * K + [(K * B, self.cfgd['comp_ldim'])])
        return replay(seed, [(K * B, self.cfgd['comp_ldim'])])

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

    def oracle(self, p, x, noise, dtype, seed_idx=None):
        nz = [n.to(dtype) for n in noise]
        if self.fam == 'v2':
            return VO.v2_forward(p, x.to(dtype), self.cfgd, nz[0], nz[1:], seed_idx=seed_idx, reference_form=False)
        if self.fam == 'genesis':
            return GO.genesis_forward(p, x.to(dtype), self.cfgd, nz[:self.K], nz[self.K])
        return MO.monet_forward(p, x.to(dtype), self.cfgd, nz[0])


def run_case(name, mods):
    fam, over, B, xseed, nseed = CASES[name]
    t0 = time.time()
    make_cfg = {'v2': VO.make_cfg, 'genesis': GO.make_cfg, 'monet': MO.make_cfg}[fam]
    cfgd = make_cfg(**over)
    cfg = R.reference_cfg(**cfgd)
    F = Family(fam, cfgd, B)
    K, S = F.K, F.S
    torch.manual_seed(0)
    model = mods[{'v2': 'genesisv2_config', 'genesis': 'genesis_config', 'monet': 'monet_config'}[fam]].load(cfg)
    if fam == 'v2':
        with torch.no_grad():
            model.att_process.colour_head.gate.gate.fill_(V2_GATE)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.train()
    x = T.make_input_of(INPUT_KIND.get(name, 'rand'), xseed, B, S)

    if fam == 'v2':
        # a noise seed whose K-1 x B argmax decisions are not near-ties (modules/attention.py:187-188 is discontinuous)
        # -- with (K-1) x B decisions among S^2 candidates each a margin of 2e-5 (the small fixtures' bar) has
        # probability ~0: take the best of a few seeds; the GPU test replays the reference's seed pixels where a
        # decision below 1e-4 falls the other way, and counts them
        with torch.no_grad():
            best = (-1.0, nseed)
            for _ in range(8):
                noise = F.noise(nseed)
                torch.manual_seed(nseed)
                sk = model(x)[2]['log_s_k']
                mm = min(float(top2_margin(noise[0], sk[i]).min()) for i in range(K - 1))
                best = max(best, (mm, nseed))
                if mm > 2e-6:
                    break
                nseed += 100
            nseed = best[1]
    noise = F.noise(nseed)

    out = {'cfg_json': np.array(json.dumps(cfgd)), 'family': np.array(fam), 'B': np.int64(B), 'x_seed': np.int64(xseed),
           'noise_seed': np.int64(nseed), 'sd_keys': np.array(list(sd.keys())),
           'sd_numel': np.array([v.numel() for v in sd.values()], dtype=np.int64),
           'sd_sum': np.array([float(v.double().sum()) for v in sd.values()]),
           'sd_asum': np.array([float(v.double().abs().sum()) for v in sd.values()]),
           'v2_gate': np.float64(V2_GATE), 'input_kind': np.array(INPUT_KIND.get(name, 'rand'))}
    T.pack_summary('in/x', x, out)
    for i, nz in enumerate(noise):
        T.pack_summary('in/noise%d' % i, nz, out)

    relu = hook_relus(model)
    torch.manual_seed(nseed)
    res = model(x)
    relu_snapshot = {k: tuple(v) for k, v in relu.items()}
    for v in relu.values():           # (the training steps further down run the hooks too: only this forward is recorded)
        v[0] = v[1] = -1 << 60
    recon, losses, stats, att, comp = res
    # the per-layer ReLU pattern of THIS forward (the GPU test grants its ReLU-decision allowance only where the HIP
    # forward's count differs): sites in module order, active outputs, outputs
    out['relu_sites'] = np.array(sorted(relu_snapshot) or [''])
    out['relu_active'] = np.array([relu_snapshot[k][0] for k in sorted(relu_snapshot)], dtype=np.int64)
    out['relu_outputs'] = np.array([relu_snapshot[k][1] for k in sorted(relu_snapshot)], dtype=np.int64)
    # the replayed noise is what the reference drew
    if fam == 'v2':
        assert torch.allclose(comp['mu_k'][1] + comp['sigma_k'][1] * noise[2], comp['z_k'][1], atol=1e-6)
        seed_idx, margins = [], []
        for step in range(K - 1):
            v = (noise[0] * stats['log_s_k'][step].exp()).flatten(2)
            seed_idx.append(v.argmax(2).flatten())
            margins.append(top2_margin(noise[0], stats['log_s_k'][step]))
        out['seed_idx'] = torch.stack(seed_idx).numpy()
        out['seed_margin'] = torch.stack(margins).detach().numpy()
        out['instance_seg_sum'] = np.int64(stats['instance_seg'].sum().item())
    elif fam == 'genesis':
        assert torch.allclose(att.mu_k[1] + att.sigma_k[1] * noise[1], att.z_k[1], atol=1e-6)
        assert torch.allclose(torch.cat(list(comp.mu_k)) + torch.cat(list(comp.sigma_k)) * noise[K], torch.cat(list(comp.z_k)), atol=1e-6)
    else:
        assert torch.allclose(torch.cat(list(comp['mu_k'])) + torch.cat(list(comp['sigma_k'])) * noise[0],
                              torch.cat(list(comp['z_k'])), atol=1e-6)
    named = F.named(res)
    for k, v in named.items():
        if v.numel() <= 8192:
            out['out/' + k] = v.detach().numpy().astype(np.float32)
        else:
            T.pack_summary('out/' + k, v, out)
    err, kl = F.aggregate(losses)
    model.zero_grad()
    (err + kl).backward()
    out['loss/err'], out['loss/kl'] = np.float64(err.item()), np.float64(kl.item())
    names, gn, gref = [], [], {}
    for pname, prm in model.named_parameters():
        g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        names.append(pname)
        gn.append(g.double().norm().item())
        gref[pname] = g.detach().clone()
        T.pack_summary('grad/' + pname, g, out)
    out['grad_norms'] = np.array(gn)
    out['param_names'] = np.array(names)
    t1 = time.time()

    # the oracle at this size: fp32 == the reference (pins the restatement at the benchmark's batch), fp64 = ground truth
    grads = {}
    for dtype in (torch.float32, torch.float64):
        p = {k: (v.clone().to(dtype if v.dtype == torch.float32 else v.dtype).requires_grad_(True) if is_param(k) and v.is_floating_point()
                 else v.clone().to(dtype if v.dtype == torch.float32 else v.dtype)) for k, v in sd.items()}
        seeds = list(torch.from_numpy(out['seed_idx']).unbind(0)) if fam == 'v2' else None
        o = F.oracle(p, x, noise, dtype, seeds)
        e, k_ = F.aggregate(o[1])
        (e + k_).backward()
        grads[dtype] = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in p.items() if v.requires_grad}
        if dtype == torch.float32:
            rel = abs(float(e + k_) - float(err + kl)) / abs(float(err + kl))
            assert rel <= 2e-5, ('oracle vs reference ELBO', rel)
            print('   oracle fp32 vs reference: ELBO rel %.2e' % rel)
        else:
            out['loss/elbo_f64'] = np.float64(float(e + k_))
    gmax = max(float(v.norm()) for v in grads[torch.float64].values())
    budget, ovr = [], []
    for n in names:
        g64 = grads[torch.float64][n]
        den = max(float(g64.norm()), 1e-6 * gmax)          # (analytically-zero gradients: measured against the largest)
        budget.append(float((gref[n].double() - g64).norm()) / den)
        ovr.append(float((grads[torch.float32][n] - gref[n].double()).norm()) / den)
    out['budget'] = np.array(budget)              # |reference fp32 gradient - fp64 gradient| / |fp64 gradient|
    out['grad_norms_f64'] = np.array([float(grads[torch.float64][n].norm()) for n in names])
    out['grad_max_f64'] = np.float64(gmax)
    print('   reference fp32 vs fp64 gradients: max %.2e median %.2e; oracle fp32 vs reference: max %.2e'
          % (max(budget), sorted(budget)[len(budget) // 2], max(ovr)))
    for i in np.argsort(-np.array(budget))[:4]:
        print('      %-44s reference-f64 %.2e   oracle32-reference %.2e' % (names[i], budget[i], ovr[i]))
    t2 = time.time()

    # three GECO + Adam steps (train.py:159-175,223-263), noise seeds nseed+1..3
    geco = mods['geco'].GECO(0.5655 * 3 * S * S, 1e-5 * (64 ** 2 / S ** 2), 0.99, 1.0, 1e-10, 10)
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), 1e-4)
    hist = []
    for it in range(3):
        opt.zero_grad()
        torch.manual_seed(nseed + 1 + it)
        _, l, _, _, _ = model(x)
        e, k_ = F.aggregate(l)
        beta = float(geco.beta)
        geco.loss(e, k_).backward()
        opt.step()
        hist.append([float(e + k_), float(e), float(k_), beta, float(geco.err_ema)])
    out['train_hist'] = np.array(hist)
    out['train_beta_final'] = np.float64(float(geco.beta))
    path = osp.join(HERE, 'full_%s.npz' % name)
    np.savez_compressed(path, **out)
    print(name, 'ELBO', float(err + kl), 'err', float(err), 'kl', float(kl),
          ('min margin %.2e' % float(out['seed_margin'].min())) if fam == 'v2' else '',
          osp.getsize(path) // 1024, 'KiB;  reference %.0f s, oracle %.0f s, steps %.0f s' % (t1 - t0, t2 - t1, time.time() - t2))


if __name__ == '__main__':
    mods = R.import_reference()
    for n in (sys.argv[1:] or list(CASES)):
        run_case(n, mods)
