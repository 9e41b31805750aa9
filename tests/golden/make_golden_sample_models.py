"""Generates tests/golden/{genesis,monet,vae}_sample_*.npz from the REAL reference's sample() methods
(models/genesis_config.py:345-425, models/monet_config.py:172-198, models/vae_config.py:89-96), imported from
/root/reference in the build container -- the same recipe as make_golden_sample.py (GENESIS-V2): closed-form weights
(genesis_amd.testing.formula_state_dict, incl. BatchNorm running statistics), evaluation mode (every caller of the
reference's sample() is in eval mode: train.py:425,463, scripts/compute_fid.py:104,125,
scripts/visualise_generation.py:83,85), and torch.normal wrapped for the duration of the call so that every draw is
`mean + std * e` with a recorded standard-normal `e` (the arithmetic torch.normal itself performs).

Fixture = config, batch size, the recorded draws in call order, the sampled latents and the reference's outputs
(generated image, x_k, log_m_k, mx_k; log_s_k for GENESIS).

    python tests/golden/make_golden_sample_models.py [family[:case] ...]
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
from oracle import genesis_oracle as GO  # noqa: E402
from oracle import monet_oracle as MO  # noqa: E402
from oracle import vae_oracle as VO  # noqa: E402

# name: (cfg overrides, batch size, K_steps passed to sample (None = omitted), seed of the draws, train mode?)
GEN_CASES = {
    'tiny': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8), 3, None, 71, False),
    'tiny_in': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, enc_norm='in', dec_norm='in'), 2, 3, 72, False),
    'tiny_noprior': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, comp_prior=False), 2, None, 73, False),
    'tiny_onestage': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, two_stage=False), 2, None, 74, False),
    'tiny_sym': (dict(K_steps=3, img_size=32, attention_latents=16, comp_ldim=8, comp_symmetric=True), 2, None, 75, False),
    'tiny_train': (dict(K_steps=4, img_size=32, attention_latents=16, comp_ldim=8), 3, None, 76, True),
    'cfg3': (dict(K_steps=7, img_size=64), 2, 7, 77, False),
}
MONET_CASES = {
    'tiny': (dict(K_steps=3, img_size=32), 3, None, 81, False),
    'tiny_k5': (dict(K_steps=3, img_size=32), 2, 5, 82, False),
    'tiny_scope': (dict(K_steps=4, img_size=32, prior_mode='scope', montecarlo_kl=False), 2, None, 83, False),
    'cfg4': (dict(K_steps=7, img_size=64), 2, None, 84, False),
}
VAE_CASES = {
    'tiny': (dict(img_size=32, latent_dimension=16), 3, None, 91, False),
    'tiny_bcast': (dict(img_size=32, latent_dimension=16, broadcast_decoder=True), 2, None, 92, False),
    'cfg1': (dict(img_size=64), 2, None, 93, False),
}


def recorded_sample(model, seed, *args):
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
            img, stats = model.sample(*args)
    finally:
        torch.normal = real_normal
    return img, stats, draws, samples


def store(out, named, full):
    for k, v in named.items():
        if full or v.numel() <= 4096:
            out['out/' + k] = v.detach().numpy().astype(np.float32)
        else:
            T.pack_summary('out/' + k, v, out)


def prepare(mod, cfgd, train):
    cfg = R.reference_cfg(**cfgd)
    torch.manual_seed(0)
    model = mod.load(cfg)
    sd = T.formula_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model.train() if train else model.eval()
    return model, sd


def head(cfgd, B, Ks, seed, train, sd):
    return {'cfg_json': np.array(json.dumps(cfgd)), 'B': np.int64(B), 'K_arg': np.int64(-1 if Ks is None else Ks),
            'seed': np.int64(seed), 'train_mode': np.int64(int(train)), 'sd_keys': np.array(list(sd.keys())),
            'sd_numel': np.array([v.numel() for v in sd.values()], dtype=np.int64)}


def run_genesis(name, mods):
    over, B, Ks, seed, train = GEN_CASES[name]
    cfgd = GO.make_cfg(**over)
    model, sd = prepare(mods['genesis_config'], cfgd, train)
    args = (B,) if Ks is None else (B, Ks)
    img, stats, draws, samples = recorded_sample(model, seed, *args)
    K = cfgd['K_steps']
    two = cfgd.get('two_stage', True)
    assert len(draws) == (2 * K if two else K), len(draws)
    out = head(cfgd, B, Ks, seed, train, sd)
    out['eps_m'] = torch.stack(draws[:K]).numpy().astype(np.float32)
    out['out/zm_k'] = torch.stack(samples[:K]).numpy().astype(np.float32)
    if two:
        out['eps_c'] = torch.stack(draws[K:]).numpy().astype(np.float32)
        out['out/zc_k'] = torch.stack(samples[K:]).numpy().astype(np.float32)
    st = lambda l: torch.stack(list(l))  # noqa: E731
    store(out, {'img': img, 'x_k': st(stats['x_k']), 'log_m_k': st(stats['log_m_k']), 'log_s_k': st(stats['log_s_k']),
                'mx_k': st(stats['mx_k'])}, S_full(cfgd))
    save('genesis_sample_%s' % name, out, img)


def run_monet(name, mods):
    over, B, Ks, seed, train = MONET_CASES[name]
    cfgd = MO.make_cfg(**over)
    model, sd = prepare(mods['monet_config'], cfgd, train)
    args = (B,) if Ks is None else (B, Ks)
    img, stats, draws, samples = recorded_sample(model, seed, *args)
    assert len(draws) == 1
    K = cfgd['K_steps'] if Ks is None else Ks
    assert draws[0].shape == (B * K, cfgd['comp_ldim'])
    out = head(cfgd, B, Ks, seed, train, sd)
    out['eps'] = draws[0].numpy().astype(np.float32)
    st = lambda l: torch.stack(list(l))  # noqa: E731
    store(out, {'img': img, 'x_k': st(stats['x_k']), 'log_m_k': st(stats['log_m_k']), 'mx_k': st(stats['mx_k'])},
          S_full(cfgd))
    save('monet_sample_%s' % name, out, img)


def run_vae(name, mods):
    over, B, Ks, seed, train = VAE_CASES[name]
    cfgd = VO.make_cfg(**over)
    model, sd = prepare(mods['vae_config'], cfgd, train)
    img, stats, draws, samples = recorded_sample(model, seed, B)
    assert len(draws) == 1 and torch.equal(samples[0], stats['z'])
    out = head(cfgd, B, Ks, seed, train, sd)
    out['eps'] = draws[0].numpy().astype(np.float32)
    store(out, {'img': img}, S_full(cfgd))
    save('vae_sample_%s' % name, out, img)


def S_full(cfgd):
    return cfgd['img_size'] <= 32


def save(stem, out, img):
    path = osp.join(HERE, stem + '.npz')
    np.savez_compressed(path, **out)
    print(stem, 'img mean %.6f' % float(img.mean()), osp.getsize(path) // 1024, 'KiB')


FAMILIES = {'genesis': (GEN_CASES, run_genesis), 'monet': (MONET_CASES, run_monet), 'vae': (VAE_CASES, run_vae)}

if __name__ == '__main__':
    mods = R.import_reference()
    todo = sys.argv[1:] or list(FAMILIES)
    for item in todo:
        fam, _, case = item.partition(':')
        cases, runner = FAMILIES[fam]
        for n in ([case] if case else list(cases)):
            runner(n, mods)
