"""ORACLE (test infrastructure) -- BaselineVAE, BASELINE config 1 (models/vae_config.py:40-101): the reference's own
CPU-runnable plumbing case.  Gated-conv VAE (oracle/sylvester_oracle.py), no norms, deconv decoder (the
broadcast_decoder flag is off by default)."""
import math

import torch

from . import sylvester_oracle as S
from . import v2_oracle as V

DEFAULT_CFG = dict(latent_dimension=64, broadcast_decoder=False, pixel_bound=True, pixel_std=0.7)


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    return cfg


def vae_forward(p, x, cfg, eps=None):
    """BaselineVAE.forward, vae_config.py:63-87.  eps [B, ldim]: the rsample noise (VAE.py:131-132)."""
    S_, L = cfg['img_size'], cfg['latent_dimension']
    h = S.encode(p, x, S_, 'vae', None)
    mu, var = S.posterior(p, h, 'vae')
    sigma = var.sqrt()
    if eps is None:
        eps = torch.normal(torch.zeros(x.size(0), L), torch.ones(x.size(0), L))
    z = mu + sigma * eps
    recon = S.decode(p, z, S_, 'vae', None)
    if cfg.get('pixel_bound', True):
        recon = torch.sigmoid(recon)
    err = -V.normal_log_prob(x, recon, cfg.get('pixel_std', 0.7)).sum(dim=(1, 2, 3))
    kl = (V.normal_log_prob(z, mu, sigma) - V.normal_log_prob(z, 0.0, 1.0)).sum(1)
    return recon, {'err': err, 'kl_l': kl}, dict(mu=mu, sigma=sigma, z=z), None, None


def param_shapes(cfg):
    return S.param_shapes('vae', cfg['latent_dimension'], 3, cfg['img_size'], 3, None, None)


def template_state_dict(cfg):
    return {k: torch.zeros(s, dtype=dt) for k, (s, dt) in param_shapes(cfg).items()}
