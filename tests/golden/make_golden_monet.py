"""Generates tests/golden/monet_*.npz from the REAL reference MONet (models/monet_config.py imported from
/root/reference in the build container).  Same recipe as make_golden.py: closed-form weights, seeded inputs,
replayed rsample noise (one standard-normal draw of [K*B, ldim], modules/component_vae.py:73), forward outputs,
parameter gradients of err + kl_l + kl_m, and three GECO + Adam steps."""
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
from oracle import monet_oracle as M  # noqa: E402

CASES = {
    'tiny': (dict(K_steps=3, img_size=32), 2, 31, 41, True),
    'tiny_k4': (dict(K_steps=4, img_size=32, filter_start=16, comp_enc_channels=16, comp_dec_channels=16, comp_ldim=8), 3, 32, 42, False),
    'cfg4': (dict(K_steps=7, img_size=64), 2, 33, 43, False),
    'tiny_scope': (dict(K_steps=4, img_size=32, prior_mode='scope', montecarlo_kl=False), 2, 34, 44, False),
}


def draw_eps(seed, n, L):
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    eps = torch.normal(torch.zeros(n, L), torch.ones(n, L))
    torch.set_rng_state(state)
    return eps


def run_case(name, mods):
    over, B, xseed, nseed, full = CASES[name]
    cfgd = M.make_cfg(**over)
    cfg = R.reference_cfg(**cfgd)
    K, S, L = cfg.K_steps, cfg.img_size, cfg.comp_ldim
    torch.manual_seed(0)
    model = mods['monet_config'].load(cfg)
    sd = T.formula_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model.train()
    x = T.make_input(xseed, B, S)
    eps = draw_eps(nseed, K * B, L)
    out = {'cfg_json': np.array(json.dumps(cfgd)), 'B': np.int64(B), 'x_seed': np.int64(xseed),
           'noise_seed': np.int64(nseed), 'sd_keys': np.array(list(sd.keys())),
           'sd_numel': np.array([v.numel() for v in sd.values()], dtype=np.int64)}
    T.pack_summary('in/x', x, out)
    T.pack_summary('in/eps', eps, out)
    torch.manual_seed(nseed)
    recon, losses, stats, _, comp = model(x)
    z_replay = torch.cat(list(comp['mu_k'])) + torch.cat(list(comp['sigma_k'])) * eps
    assert torch.allclose(z_replay, torch.cat(list(comp['z_k'])), atol=1e-6), 'noise replay mismatch'
    named = {'err': losses['err'], 'kl_m': losses['kl_m'], 'kl_l_k': torch.stack(list(losses['kl_l_k'])),
             'recon': recon, 'log_m_k': torch.stack(list(stats['log_m_k'])),
             'log_s_k': torch.stack(list(stats['log_s_k'])), 'x_r_k': torch.stack(list(stats['x_r_k'])),
             'log_m_r_k': torch.stack(list(stats['log_m_r_k'])), 'mu_k': torch.stack(list(comp['mu_k'])),
             'sigma_k': torch.stack(list(comp['sigma_k'])), 'z_k': torch.stack(list(comp['z_k']))}
    for k, v in named.items():
        if full or v.numel() <= 4096:
            out['out/' + k] = v.detach().numpy().astype(np.float32)
        else:
            T.pack_summary('out/' + k, v, out)
    err = losses['err'].mean(0)
    kl = torch.stack(list(losses['kl_l_k']), dim=1).mean(0).sum() + losses['kl_m'].mean(0)
    model.zero_grad()
    (err + kl).backward()
    out['loss/err'] = np.float64(err.item())
    out['loss/kl'] = np.float64(kl.item())
    gn = []
    for pname, prm in model.named_parameters():
        g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        gn.append(g.double().norm().item())
        T.pack_summary('grad/' + pname, g, out)
    out['grad_norms'] = np.array(gn)
    out['param_names'] = np.array([n for n, _ in model.named_parameters()])
    geco = mods['geco'].GECO(0.5655 * 3 * S * S, 1e-5 * (64 ** 2 / S ** 2), 0.99, 1.0, 1e-10, 10)
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), 1e-4)
    hist = []
    for it in range(3):
        opt.zero_grad()
        torch.manual_seed(nseed + 1 + it)
        _, l, _, _, _ = model(x)
        e = l['err'].mean(0)
        k_ = torch.stack(list(l['kl_l_k']), dim=1).mean(0).sum() + l['kl_m'].mean(0)
        beta = float(geco.beta)
        geco.loss(e, k_).backward()
        opt.step()
        hist.append([float(e + k_), float(e), float(k_), beta, float(geco.err_ema)])
    out['train_hist'] = np.array(hist)
    out['train_beta_final'] = np.float64(float(geco.beta))
    path = osp.join(HERE, 'monet_%s.npz' % name)
    np.savez_compressed(path, **out)
    print(name, 'err', losses['err'].tolist(), 'kl', float(kl), osp.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    mods = R.import_reference()
    for n in (sys.argv[1:] or list(CASES)):
        run_case(n, mods)
