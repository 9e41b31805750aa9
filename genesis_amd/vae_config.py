"""BaselineVAE model config (BASELINE config 1) -- mirror of the reference's `models/vae_config.py` (flags :28-33,
`load(cfg)` :36-37, `BaselineVAE` :40-101) on the HIP path: same `state_dict` (`vae.*`), same
`forward(x) -> (recon, {err, kl_l}, stats, None, None)`, `sample`, `get_features`."""
import math

import torch
import torch.nn as nn

from genesis_amd import autostep
from genesis_amd import compat as _compat

_compat.install()

from attrdict import AttrDict  # noqa: E402
from forge import flags  # noqa: E402

from genesis_amd.genesisv2_config import _cfg_get, _normal_log_prob  # noqa: E402
from genesis_amd.sylvester import SylvesterVAE  # noqa: E402

# GatedConvVAE (models/vae_config.py:28-33)
flags.DEFINE_integer('latent_dimension', 64, 'Latent channels.')
flags.DEFINE_boolean('broadcast_decoder', False, 'Use broadcast decoder instead of deconv.')
flags.DEFINE_boolean('pixel_bound', True, 'Bound pixel values to [0, 1].')
flags.DEFINE_float('pixel_std', 0.7, 'StdDev of reconstructed pixels.')


def load(cfg):
    return BaselineVAE(cfg)


class BaselineVAE(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        cfg.K_steps = None                    # vae_config.py:44
        self.ldim = cfg.latent_dimension
        self.pixel_std = _cfg_get(cfg, 'pixel_std', 0.7)
        self.pixel_bound = _cfg_get(cfg, 'pixel_bound', True)
        self.debug = _cfg_get(cfg, 'debug', False)
        self.img_size = cfg.img_size
        self.vae = SylvesterVAE(self.ldim, [3, cfg.img_size, cfg.img_size], 3)
        self._zero_logw = None
        self.broadcast_decoder = bool(_cfg_get(cfg, 'broadcast_decoder', False))
        if self.broadcast_decoder:
            # vae_config.py:53-61: the deconv decoder is REPLACED (after it was constructed: same RNG consumption) by
            # Flatten -> BroadcastDecoder(ldim -> 64, h 64, 4 layers, ELU) -> ELU, and p_x_mean by a 64 -> 3 1x1 conv
            from genesis_amd.monet_config import _BroadcastDecoderParams
            self.vae.p_x_nn = nn.Sequential(nn.Flatten(), _BroadcastDecoderParams(self.ldim, 64, 64, 4), nn.ELU())
            self.vae.p_x_mean = nn.Conv2d(64, 3, 1, 1, 0)
            self._coords = {}

    def forward(self, x, eps=None):
        """x [B,3,S,S] on the GPU; eps [B, ldim] injects the rsample noise (VAE.py:131-132)."""
        if x.is_cuda:
            autostep.arm(self)      # the unchanged train.py loop: this iteration on TrainStep's launch structure (autostep.py)
        if not x.is_cuda:
            from genesis_amd._lib import GenesisHipError
            raise GenesisHipError('BaselineVAE: the HIP path needs device tensors; there is no CPU fallback')
        from genesis_amd import functions as fn
        B = x.shape[0]
        h = self.vae.encode_features(x)
        zh = self.vae.posterior_heads(h)                               # [B, 2 ldim] = (mu | pre-sigma)
        if eps is None:
            eps = torch.randn(B, self.ldim, device=x.device)
        # sigma = to_sigma(pre-sigma) (= sqrt(ToVar), vae_config.py:66), z = mu + sigma eps (rsample, :67) and log q(z): one launch
        z, mu, sigma, log_q = (t[0] for t in fn.PosteriorFn.apply(zh.unsqueeze(1), eps.unsqueeze(0)))
        x_mean = self._decode(z)
        # likelihood (vae_config.py:72-77): one component with log-weight 0 of the mixture kernel -- -sum log N(x; recon, std),
        # recon = sigmoid(x_mean) under pixel_bound
        key = (B,) + tuple(x.shape[2:]) + (str(x.device),)
        if self._zero_logw is None or self._zero_logw[0] != key:
            self._zero_logw = (key, torch.zeros(1, B, 1, *x.shape[2:], device=x.device))
        err, recon, _ = fn.MixtureWFn.apply(x, x_mean, self._zero_logw[1], 1, float(self.pixel_std), float(self.pixel_std),
                                            bool(self.pixel_bound))
        # Monte-Carlo KL against N(0, 1) at z (vae_config.py:79-81): log q(z) - log p(z)
        kl = fn.PriorLogPFn.apply(z.unsqueeze(0), None, log_q.unsqueeze(0))[0]
        stats = AttrDict(x=x_mean, mu=mu, sigma=sigma, z=z)
        return recon, AttrDict(err=err, kl_l=kl), stats, None, None

    def _decode(self, z):
        if not self.broadcast_decoder:
            return self.vae.decode(z)
        from genesis_amd import functions as fn
        from genesis_amd.genesisv2_config import pixel_coords
        dm = self.vae.p_x_nn[1]
        key = str(z.device)
        if key not in self._coords:
            self._coords[key] = pixel_coords(self.img_size + 2 * dm.num_layers).contiguous().to(z.device)
        # spatial broadcast + coordinates + 4 valid 3x3 convs (ELU) on the canvas-free HIP path; the decoder's 1x1 conv
        # and the Sequential's trailing nn.ELU are one launch (out_act)
        h = fn.BroadcastDecoderFn.apply(z, self._coords[key], 'elu', 'elu', *dm.flat_params())
        return fn.Conv1x1Fn.apply(h, self.vae.p_x_mean.weight, self.vae.p_x_mean.bias)

    @torch.no_grad()
    def sample(self, batch_size, *args, eps=None, **kwargs):
        """models/vae_config.py:89-96; `eps` [B, ldim] injects the standard-normal draw (parity tests)."""
        dev = self.vae.p_x_mean.weight.device
        z = torch.randn(batch_size, self.ldim, device=dev) if eps is None else eps.to(dev).contiguous()
        assert z.shape == (batch_size, self.ldim)
        x = self._decode(z)
        if self.pixel_bound:
            x = torch.sigmoid(x)
        return x, AttrDict(z=z)

    def get_features(self, image_batch):
        with torch.no_grad():
            _, _, stats, _, _ = self.forward(image_batch)
        return stats.z
