"""Generates tests/golden/*.npz from the REAL reference (imported from /root/reference
in the build container -- it cannot travel to the GPU box; only these data files do).

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py tiny       # one case

Each fixture holds: the config, the inputs (x, injected rand_pixel / eps -- replayed
from the seed in the reference's own draw order), the reference's forward outputs
(full tensors for the tiny case, sum/abs-sum/strided samples for large ones), the
gradients of err.mean() + sum_k mean_b kl_l_k (+ kl_m) w.r.t. every parameter
(norm + samples) and three GECO+Adam training steps (elbo / err / kl / beta).
Weights are closed-form (genesis_amd.testing.formula_state_dict) so no weight file is
needed.
"""
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

CASES = {
    # name: (cfg overrides, B, x seed, noise seed, full tensors?)
    'tiny': (dict(K_steps=4, img_size=32, feat_dim=16), 2, 11, 21, True),
    'tiny_b3k3': (dict(K_steps=3, img_size=32, feat_dim=8), 3, 12, 22, True),
    'tiny_noar': (dict(K_steps=4, img_size=32, feat_dim=16, autoreg_prior=False), 2, 13, 23, False),
    'tiny_klm': (dict(K_steps=4, img_size=32, feat_dim=16, klm_loss=True), 2, 14, 24, False),
    'tiny_klm_nodetach': (dict(K_steps=4, img_size=32, feat_dim=16, klm_loss=True, detach_mr_in_klm=False), 2, 31, 41, False),
    'tiny_nosemi': (dict(K_steps=4, img_size=32, feat_dim=16, semiconv=False), 2, 15, 25, False),
    'tiny_laplacian': (dict(K_steps=4, img_size=32, feat_dim=16, kernel='laplacian'), 2, 16, 26, False),
    'tiny_epanechnikov': (dict(K_steps=4, img_size=32, feat_dim=16, kernel='epanechnikov'), 2, 17, 27, False),
    'metric': (dict(K_steps=7, img_size=64, feat_dim=64), 2, 18, 28, False),
    'cfg2': (dict(K_steps=5, img_size=64, feat_dim=64), 2, 19, 29, False),
    'cfg5': (dict(K_steps=11, img_size=128, feat_dim=64), 1, 20, 30, False),
    # dynamic_K (genesisv2_config.py:118-137): a batch (finished images padded with -1e10 masks, no att_stats) and a
    # single image (the lists end early)
    'tiny_dynk': (dict(K_steps=5, img_size=32, feat_dim=16, dynamic_K=True), 3, 42, 52, True),
    'tiny_dynk_b1': (dict(K_steps=5, img_size=32, feat_dim=16, dynamic_K=True), 1, 42, 52, True),
}


def top2_margin(rand_pixel, log_s):
    v = (rand_pixel * log_s.exp()).flatten(1)
    top = v.topk(2, dim=1).values
    return (top[:, 0] - top[:, 1]) / top[:, 0].abs().clamp_min(1e-30)


def run_case(name, mods):
    over, B, xseed, nseed, full = CASES[name]
    cfg = R.reference_cfg(**over)
    K, S, D = cfg.K_steps, cfg.img_size, cfg.feat_dim
    torch.manual_seed(0)
    model = mods['genesisv2_config'].load(cfg)
    sd = T.formula_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model.train()
    x = T.make_input(xseed, B, S)
    # pick a noise seed whose argmax seeds are not near-ties (modules/attention.py:187-188
    # is discontinuous): first of nseed, nseed+100, ... with top-2 margin > 2e-5.
    dyn = bool(over.get('dynamic_K', False))
    with torch.no_grad():
        for _ in range(50):
            rand_pixel, eps_k = T.draw_noise(nseed, B, S, D, K)
            if dyn:
                break          # (no scope list to test the margins on: att_stats / log_s_k are None for a batch)
            torch.manual_seed(nseed)
            st = model(x)[2]
            mm = min(float(top2_margin(rand_pixel, st['log_s_k'][i]).min()) for i in range(K - 1))
            if mm > 2e-5:
                break
            nseed += 100

    out = {}
    out['cfg_json'] = np.array(json.dumps(
        dict({k: cfg[k] for k in ('K_steps', 'img_size', 'feat_dim', 'kernel', 'semiconv',
                                  'klm_loss', 'detach_mr_in_klm', 'pixel_bound',
                                  'autoreg_prior', 'pixel_std1')}, **({'dynamic_K': True} if dyn else {}))))
    out['B'] = np.int64(B)
    out['x_seed'] = np.int64(xseed)
    out['noise_seed'] = np.int64(nseed)
    out['sd_keys'] = np.array(list(sd.keys()))
    out['sd_numel'] = np.array([v.numel() for v in sd.values()], dtype=np.int64)
    T.pack_summary('in/x', x, out)
    T.pack_summary('in/rand_pixel', rand_pixel, out)
    T.pack_summary('in/eps', torch.stack(eps_k), out)

    torch.manual_seed(nseed)
    recon, losses, stats, att, comp = model(x)
    # sanity: the replayed noise is what the reference drew
    z_replay = comp['mu_k'][0] + comp['sigma_k'][0] * eps_k[0]
    assert torch.allclose(z_replay, comp['z_k'][0], atol=1e-6), 'noise replay mismatch'

    named = {
        'err': losses['err'], 'kl_l_k': torch.stack(list(losses['kl_l_k'])),
        'recon': recon, 'log_m_k': torch.stack(list(stats['log_m_k'])),
        'x_r_k': torch.stack(list(stats['x_r_k'])),
        'log_m_r_k': torch.stack(list(stats['log_m_r_k'])),
        'mu_k': torch.stack(list(comp['mu_k'])), 'sigma_k': torch.stack(list(comp['sigma_k'])),
        'z_k': torch.stack(list(comp['z_k'])),
    }
    if dyn:
        lm = named['log_m_k']
        out['slots'] = np.int64(lm.shape[0])
        out['seed_margin'] = np.array([1.0])
        print('   dynamic_K: %d mask tensors; padded (-1e10) slots per image:' % lm.shape[0],
              (lm.flatten(2).max(2).values < -1e9).sum(0).flatten().tolist())
    if stats['log_s_k'] is not None and att is not None and not dyn:
        seed_idx = []
        margins = []
        flat_rand = rand_pixel
        for step in range(K - 1):
            v = (flat_rand * stats['log_s_k'][step].exp()).flatten(2)
            seed_idx.append(v.argmax(2).flatten())
            margins.append(top2_margin(flat_rand, stats['log_s_k'][step]))
        out['seed_idx'] = torch.stack(seed_idx).numpy()
        out['seed_margin'] = torch.stack(margins).detach().numpy()
    if stats['log_s_k'] is not None:
        named['log_s_k'] = torch.stack(list(stats['log_s_k']))
    if att is not None:
        named['colour'] = att['colour']
        named['seeds'] = torch.stack(list(att['seeds']))
    else:
        att = {'delta': None}
    if att['delta'] is not None:
        named['delta'] = att['delta']
    if 'kl_m' in losses:
        named['kl_m'] = losses['kl_m']
    out['instance_seg_sum'] = np.int64(stats['instance_seg'].sum().item())
    for k, v in named.items():
        if full or v.numel() <= 4096:
            out['out/' + k] = v.detach().numpy().astype(np.float32)
        else:
            T.pack_summary('out/' + k, v, out)

    # gradients of the beta=1 objective (train.py:227-242 aggregation)
    err = losses['err'].mean(0)
    kl = torch.stack(list(losses['kl_l_k']), dim=1).mean(0).sum()
    if 'kl_m' in losses:
        kl = kl + losses['kl_m'].mean(0)
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

    # three GECO + Adam steps (train.py:159-175,223-263), noise seeds nseed+1..3
    geco_mod = mods['geco']
    geco = geco_mod.GECO(0.5655 * 3 * S * S, 1e-5 * (64 ** 2 / S ** 2), 0.99, 1.0, 1e-10, 10)
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), 1e-4)
    hist = []
    for it in range(3):
        opt.zero_grad()
        torch.manual_seed(nseed + 1 + it)
        _, l, _, _, _ = model(x)
        e = l['err'].mean(0)
        k_ = torch.stack(list(l['kl_l_k']), dim=1).mean(0).sum()
        if 'kl_m' in l:
            k_ = k_ + l['kl_m'].mean(0)
        beta = float(geco.beta)
        loss = geco.loss(e, k_)
        loss.backward()
        opt.step()
        hist.append([float(e + k_), float(e), float(k_), beta, float(geco.err_ema)])
    out['train_hist'] = np.array(hist)
    out['train_beta_final'] = np.float64(float(geco.beta))
    path = osp.join(HERE, 'v2_%s.npz' % name)
    np.savez_compressed(path, **out)
    print(name, 'err', losses['err'].tolist(), 'kl', float(kl), 'min margin',
          float(out['seed_margin'].min()), osp.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    mods = R.import_reference()
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n, mods)
