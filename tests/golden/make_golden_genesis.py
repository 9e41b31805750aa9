"""Generates tests/golden/{vae,genesis}_*.npz from the REAL reference (models/vae_config.py, models/genesis_config.py
imported from /root/reference in the build container): closed-form weights, seeded inputs, replayed rsample noise,
forward outputs, parameter gradients, three GECO + Adam steps (training mode: BatchNorm uses batch statistics)."""
import json
import os.path as osp
import sys

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, REPO)

from genesis_amd import testing as T  # noqa: E402
from oracle import ref_import as R  # noqa: E402
from oracle import genesis_oracle as GO  # noqa: E402
from oracle import vae_oracle as VO  # noqa: E402

VAE_CASES = {'tiny': (dict(img_size=32, latent_dimension=16), 2, 51, 61), 'cfg1': (dict(img_size=64), 2, 52, 62),
             'tiny_bcast': (dict(img_size=32, latent_dimension=16, broadcast_decoder=True), 2, 56, 66),
             'cfg1_b32': (dict(img_size=64), 32, 152, 162)}          # BASELINE config 1 at its batch
GEN_CASES = {'tiny': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8), 2, 53, 63),
             'tiny_in': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, enc_norm='in', dec_norm='in'), 3, 54, 64),
             'cfg3': (dict(K_steps=7, img_size=64), 2, 55, 65),
             'tiny_noprior': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, comp_prior=False), 2, 57, 67),
             'tiny_onestage': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, two_stage=False), 2, 58, 68),
             'tiny_sym': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, comp_symmetric=True), 2, 59, 69)}


def replay(seed, shapes):
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    out = [torch.normal(torch.zeros(*s), torch.ones(*s)) for s in shapes]
    torch.set_rng_state(state)
    return out


def finish(prefix, name, model, sd, x, losses_fn, out, geco_mod, S, nseed, extra_named):
    for k, v in extra_named.items():
        if v.numel() <= 4096:
            out['out/' + k] = v.detach().numpy().astype(np.float32)
        else:
            T.pack_summary('out/' + k, v, out)
    err, kl = losses_fn(model, nseed)
    model.zero_grad()
    (err + kl).backward()
    out['loss/err'], out['loss/kl'] = np.float64(err.item()), np.float64(kl.item())
    gn = []
    for pname, prm in model.named_parameters():
        g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        gn.append(g.double().norm().item())
        T.pack_summary('grad/' + pname, g, out)
    out['grad_norms'] = np.array(gn)
    out['param_names'] = np.array([n for n, _ in model.named_parameters()])
    geco = geco_mod.GECO(0.5655 * 3 * S * S, 1e-5 * (64 ** 2 / S ** 2), 0.99, 1.0, 1e-10, 10)
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), 1e-4)
    hist = []
    for it in range(3):
        opt.zero_grad()
        e, k_ = losses_fn(model, nseed + 1 + it)
        beta = float(geco.beta)
        geco.loss(e, k_).backward()
        opt.step()
        hist.append([float(e + k_), float(e), float(k_), beta, float(geco.err_ema)])
    out['train_hist'] = np.array(hist)
    out['train_beta_final'] = np.float64(float(geco.beta))
    path = osp.join(HERE, '%s_%s.npz' % (prefix, name))
    np.savez_compressed(path, **out)
    print(prefix, name, 'err', float(err), 'kl', float(kl), osp.getsize(path) // 1024, 'KiB')


def base(cfgd, B, xseed, nseed, sd):
    return {'cfg_json': np.array(json.dumps(cfgd)), 'B': np.int64(B), 'x_seed': np.int64(xseed),
            'noise_seed': np.int64(nseed), 'sd_keys': np.array(list(sd.keys())),
            'sd_numel': np.array([v.numel() for v in sd.values()], dtype=np.int64)}


def run_vae(name, mods):
    over, B, xseed, nseed = VAE_CASES[name]
    cfgd = VO.make_cfg(**over)
    cfg = R.reference_cfg(**cfgd)
    torch.manual_seed(0)
    model = mods['vae_config'].load(cfg)
    sd = T.formula_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model.train()
    S, L = cfgd['img_size'], cfgd['latent_dimension']
    x = T.make_input(xseed, B, S)
    out = base(cfgd, B, xseed, nseed, sd)
    T.pack_summary('in/x', x, out)
    (eps,) = replay(nseed, [(B, L)])
    T.pack_summary('in/eps', eps, out)
    torch.manual_seed(nseed)
    recon, losses, stats, _, _ = model(x)
    assert torch.allclose(stats.mu + stats.sigma * eps, stats.z, atol=1e-6)

    def losses_fn(m, seed):
        torch.manual_seed(seed)
        _, l, _, _, _ = m(x)
        return l['err'].mean(0), l['kl_l'].mean(0)

    finish('vae', name, model, sd, x, losses_fn, out, mods['geco'], S, nseed,
           {'err': losses['err'], 'kl_l': losses['kl_l'], 'recon': recon, 'mu': stats.mu, 'sigma': stats.sigma, 'z': stats.z})


def run_gen(name, mods):
    over, B, xseed, nseed = GEN_CASES[name]
    cfgd = GO.make_cfg(**over)
    cfg = R.reference_cfg(**cfgd)
    torch.manual_seed(0)
    model = mods['genesis_config'].load(cfg)
    sd = T.formula_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model.train()
    K, S, L, Lc = cfgd['K_steps'], cfgd['img_size'], cfgd['attention_latents'], cfgd['comp_ldim']
    x = T.make_input(xseed, B, S)
    out = base(cfgd, B, xseed, nseed, sd)
    T.pack_summary('in/x', x, out)
    two = cfgd.get('two_stage', True)
    noise = replay(nseed, [(B, L)] * K + ([(K * B, Lc)] if two else []))
    T.pack_summary('in/eps_m', torch.stack(noise[:K]), out)
    if two:
        T.pack_summary('in/eps_c', noise[K], out)
    torch.manual_seed(nseed)
    recon, losses, stats, att, comp = model(x)
    assert torch.allclose(att.mu_k[1] + att.sigma_k[1] * noise[1], att.z_k[1], atol=1e-6), 'noise replay'
    if two:
        assert torch.allclose(torch.cat(list(comp.mu_k)) + torch.cat(list(comp.sigma_k)) * noise[K], torch.cat(list(comp.z_k)), atol=1e-6)
    else:
        assert comp is None and 'kl_l_k' not in losses
    model.load_state_dict(sd)     # undo the BatchNorm running-stat update of this forward

    def losses_fn(m, seed):
        torch.manual_seed(seed)
        _, l, _, _, _ = m(x)
        kl = torch.stack(list(l['kl_m_k']), 1).mean(0).sum()
        if 'kl_l_k' in l:
            kl = kl + torch.stack(list(l['kl_l_k']), 1).mean(0).sum()
        return l['err'].mean(0), kl

    st = lambda l: torch.stack(list(l))  # noqa: E731
    named = {'err': losses['err'], 'kl_m_k': st(losses['kl_m_k']), 'recon': recon,
             'log_m_k': st(stats.log_m_k), 'x_r_k': st(stats.x_r_k), 'att_mu_k': st(att.mu_k), 'att_z_k': st(att.z_k)}
    if two:
        named.update({'kl_l_k': st(losses['kl_l_k']), 'comp_mu_k': st(comp.mu_k), 'comp_sigma_k': st(comp.sigma_k),
                      'comp_z_k': st(comp.z_k)})
    finish('genesis', name, model, sd, x, losses_fn, out, mods['geco'], S, nseed, named)


if __name__ == '__main__':
    mods = R.import_reference()
    which = sys.argv[1:] or ['vae', 'genesis']          # 'vae', 'genesis', or single cases 'vae:tiny_bcast'
    for n in VAE_CASES:
        if 'vae' in which or 'vae:' + n in which:
            run_vae(n, mods)
    for n in GEN_CASES:
        if 'genesis' in which or 'genesis:' + n in which:
            run_gen(n, mods)
