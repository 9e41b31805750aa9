"""ORACLE (test infrastructure) -- BaselineVAE, BASELINE config 1 (models/vae_config.py:40-101): the reference's own
CPU-runnable plumbing case.  Gated-conv VAE (oracle/sylvester_oracle.py), no norms; deconv decoder, or with
broadcast_decoder=True (vae_config.py:53-61) Flatten -> BroadcastDecoder(ldim -> 64, 4 layers, ELU) -> ELU and a
64 -> 3 1x1 p_x_mean."""
import math

import torch

import torch.nn.functional as F

from . import monet_oracle as M
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
    if cfg.get('broadcast_decoder', False):
        h = F.elu(M.broadcast_decoder(p, z, S_, 4, act=F.elu, prefix='vae.p_x_nn.1.seq'))
        recon = F.conv2d(h, p['vae.p_x_mean.weight'], p['vae.p_x_mean.bias'])
    else:
        recon = S.decode(p, z, S_, 'vae', None)
    if cfg.get('pixel_bound', True):
        recon = torch.sigmoid(recon)
    err = -V.normal_log_prob(x, recon, cfg.get('pixel_std', 0.7)).sum(dim=(1, 2, 3))
    kl = (V.normal_log_prob(z, mu, sigma) - V.normal_log_prob(z, 0.0, 1.0)).sum(1)
    return recon, {'err': err, 'kl_l': kl}, dict(mu=mu, sigma=sigma, z=z), None, None


def vae_sample(p, cfg, eps):
    """BaselineVAE.sample, vae_config.py:89-96: z = the standard-normal draw eps [B, ldim], decoded (the VAE has no
    norm layers, so train / eval mode do not differ)."""
    S_ = cfg['img_size']
    if cfg.get('broadcast_decoder', False):
        h = F.elu(M.broadcast_decoder(p, eps, S_, 4, act=F.elu, prefix='vae.p_x_nn.1.seq'))
        x = F.conv2d(h, p['vae.p_x_mean.weight'], p['vae.p_x_mean.bias'])
    else:
        x = S.decode(p, eps, S_, 'vae', None)
    return torch.sigmoid(x) if cfg.get('pixel_bound', True) else x


def param_shapes(cfg):
    sh = S.param_shapes('vae', cfg['latent_dimension'], 3, cfg['img_size'], 3, None, None)
    if cfg.get('broadcast_decoder', False):
        # vae_config.py:53-61 replaces p_x_nn / p_x_mean: the gated deconvs leave the state dict, in their place (and at
        # their position in the key order) the BroadcastDecoder's convs; p_x_mean becomes 64 -> 3
        f32 = torch.float32
        out = {}
        done = False
        for k, v in sh.items():
            if k.startswith('vae.p_x_nn.') or k.startswith('vae.p_x_mean.'):
                if not done:
                    L = cfg['latent_dimension']
                    for l, cin in enumerate((L + 2, 64, 64, 64)):
                        out['vae.p_x_nn.1.seq.%d.weight' % (1 + 2 * l)] = ((64, cin, 3, 3), f32)
                        out['vae.p_x_nn.1.seq.%d.bias' % (1 + 2 * l)] = ((64,), f32)
                    out['vae.p_x_nn.1.seq.9.weight'] = ((64, 64, 1, 1), f32)
                    out['vae.p_x_nn.1.seq.9.bias'] = ((64,), f32)
                    out['vae.p_x_mean.weight'] = ((3, 64, 1, 1), f32)
                    out['vae.p_x_mean.bias'] = ((3,), f32)
                    done = True
                continue
            out[k] = v
        sh = out
    return sh


def template_state_dict(cfg):
    return {k: torch.zeros(s, dtype=dt) for k, (s, dt) in param_shapes(cfg).items()}
