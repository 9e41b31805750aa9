"""Generates tests/golden/v2_sample_*.npz from the REAL reference's GenesisV2.sample
(models/genesisv2_config.py:227-256), imported from /root/reference in the build container.

The reference draws its noise inside sample() (Normal(0,1).sample, then one p_z.sample() per later slot); to
replay it, torch.normal is wrapped for the duration of the call so that every draw is `mean + std * e` with a
recorded standard-normal `e` (the arithmetic torch.normal itself performs).  Fixture = config, batch size, seed,
the recorded draws eps [K,B,D], and the reference's outputs (recon, x_k, log_m_k; z_k recovered from the draws
of the wrapped calls).  Weights are closed-form (genesis_amd.testing.formula_state_dict).

    python tests/golden/make_golden_sample.py
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
    # name: (cfg overrides, batch size, K_steps passed to sample (None = model's), seed, full tensors?)
    'tiny': (dict(K_steps=4, img_size=32, feat_dim=16), 3, None, 41, True),
    'tiny_k6': (dict(K_steps=4, img_size=32, feat_dim=16), 2, 6, 42, True),
    'tiny_noar': (dict(K_steps=4, img_size=32, feat_dim=16, autoreg_prior=False), 2, None, 43, False),
    'metric': (dict(K_steps=7, img_size=64, feat_dim=64), 2, None, 44, False),
}


def run_case(name, mods):
    over, B, Ks, seed, full = CASES[name]
    cfg = R.reference_cfg(**over)
    torch.manual_seed(0)
    model = mods['genesisv2_config'].load(cfg)
    sd = T.formula_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model.eval()
    draws, samples = [], []
    real_normal = torch.normal

    def recording_normal(mean, std, *a, **k):
        assert not a and not k, 'unexpected torch.normal signature in the reference sample()'
        e = torch.randn(mean.shape)
        draws.append(e)
        z = mean + std * e
        samples.append(z)
        return z

    torch.manual_seed(seed)
    torch.normal = recording_normal
    try:
        with torch.no_grad():
            recon, stats = model.sample(B, Ks)
    finally:
        torch.normal = real_normal
    K = len(stats['x_k'])
    assert len(draws) == K
    out = {'cfg_json': np.array(json.dumps({k: cfg[k] for k in ('K_steps', 'img_size', 'feat_dim', 'kernel', 'semiconv',
                                                              'klm_loss', 'detach_mr_in_klm', 'pixel_bound',
                                                              'autoreg_prior', 'pixel_std1')})),
           'B': np.int64(B), 'K_sample': np.int64(K), 'K_arg': np.int64(-1 if Ks is None else Ks),
           'eps': torch.stack(draws).numpy().astype(np.float32),
           'out/z_k': torch.stack(samples).numpy().astype(np.float32)}
    named = {'recon': recon, 'x_k': torch.stack(list(stats['x_k'])), 'log_m_k': torch.stack(list(stats['log_m_k'])),
             'mx_k': torch.stack(list(stats['mx_k']))}
    for k, v in named.items():
        if full or v.numel() <= 4096:
            out['out/' + k] = v.detach().numpy().astype(np.float32)
        else:
            T.pack_summary('out/' + k, v, out)
    path = osp.join(HERE, 'v2_sample_%s.npz' % name)
    np.savez_compressed(path, **out)
    print(name, 'K', K, 'recon mean %.6f' % float(recon.mean()), osp.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    mods = R.import_reference()
    for n in (sys.argv[1:] or list(CASES)):
        run_case(n, mods)
