"""MONet model config (BASELINE config 4) -- MI355X-native drop-in for the reference's `models/monet_config.py`
(flags :36-37, `load(cfg)` :40-41, `MONet` :44-198): same `load(cfg)` / `forward(x)` 5-tuple / `sample` /
`get_features` contract and the same `state_dict` (root buffer `std`, `att_process.core.*` UNet with InstanceNorm
and the 1x1 final_conv, `comp_vae.encoder_module.module.*`, `comp_vae.decoder_module.seq.*`).

HIP path: the recurrent UNet(IN) attention (K-1 sequential passes on [x | log_s], modules/attention.py:31-51) runs
on the same fp32-MFMA conv + fused norm kernels as GENESIS-V2 (InstanceNorm = GroupNorm with one group per
channel); the ComponentVAE (modules/component_vae.py:45-93) runs its BroadcastDecoder valid-conv chain on the
tap-conv kernel over the (S+2L)^2 broadcast canvas, its four stride-2 encoder convs on the direct-conv kernel, and
the mixture likelihood with the attention masks as mixing weights in one kernel.  Tiny dense / pointwise pieces
(the two encoder Linears, logsigmoid stick-breaking, the categorical mask KL) are plain torch-ROCm ops."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions.normal import Normal

from genesis_amd import autostep
from genesis_amd import compat as _compat

_compat.install()

from attrdict import AttrDict  # noqa: E402
from forge import flags  # noqa: E402

from genesis_amd import functions as fn  # noqa: E402
from genesis_amd import hip_ops as hip  # noqa: E402
from genesis_amd.genesisv2_config import _UNetParams, _cfg_get, _normal_log_prob, pixel_coords  # noqa: E402
from genesis_amd.lazy import Lazy, LazyAttrDict, SlotList  # noqa: E402

# Attention network (models/monet_config.py:36-37)
flags.DEFINE_integer('filter_start', 32, 'Starting number of channels in UNet.')
flags.DEFINE_string('prior_mode', 'softmax', '{scope, softmax}')
# ComponentVAE flags the reference inherits from models/genesis_config.py:41-52
flags.DEFINE_integer('comp_enc_channels', 32, 'Starting number of channels.')
flags.DEFINE_integer('comp_ldim', 16, 'Latent dimension of the VAE.')
flags.DEFINE_integer('comp_dec_channels', 32, 'Num channels in Broadcast Decoder.')
flags.DEFINE_integer('comp_dec_layers', 4, 'Num layers in Broadcast Decoder.')
flags.DEFINE_boolean('montecarlo_kl', True, 'Evaluate KL via MC samples.')
flags.DEFINE_boolean('pixel_bound', True, 'Bound pixel values to [0, 1].')
flags.DEFINE_float('pixel_std1', 0.7, 'StdDev of reconstructed pixels.')
flags.DEFINE_float('pixel_std2', 0.7, 'StdDev of reconstructed pixels.')


def load(cfg):
    return MONet(cfg)


class _SBPParams(nn.Module):
    """attention.SimpleSBP(core): holds the UNet as `core` (modules/attention.py:25-29)."""

    def __init__(self, core):
        super().__init__()
        self.core = core


class _CompEncoderParams(nn.Module):
    """MONetCompEncoder, modules/encoders.py:22-40 (key layout `module.{0,2,4,6,9,11}`)."""

    def __init__(self, img_size, c, ldim, nin=3):
        super().__init__()
        nin_mlp = 2 * c * (img_size // 16) ** 2
        nhid = max(256, 2 * ldim)
        act = nn.ReLU
        self.module = nn.Sequential(nn.Conv2d(nin + 1, c, 3, 2, 1), act(), nn.Conv2d(c, c, 3, 2, 1), act(),
                                    nn.Conv2d(c, 2 * c, 3, 2, 1), act(), nn.Conv2d(2 * c, 2 * c, 3, 2, 1), act(),
                                    nn.Flatten(), nn.Linear(nin_mlp, nhid), act(), nn.Linear(nhid, 2 * ldim))


class _BroadcastDecoderParams(nn.Module):
    """BroadcastDecoder, modules/decoders.py:21-35 (key layout `seq.{1,3,...,2L-1}` convs, `seq.{2L+1}` 1x1)."""

    def __init__(self, in_chnls, out_chnls, h_chnls, num_layers):
        super().__init__()
        mods = [nn.Identity(), nn.Conv2d(in_chnls + 2, h_chnls, 3), nn.ReLU()]
        for _ in range(num_layers - 1):
            mods.extend([nn.Conv2d(h_chnls, h_chnls, 3), nn.ReLU()])
        mods.append(nn.Conv2d(h_chnls, out_chnls, 1))
        self.seq = nn.Sequential(*mods)
        self.num_layers = num_layers

    def flat_params(self):
        p = []
        for l in range(self.num_layers):
            p.extend((self.seq[1 + 2 * l].weight, self.seq[1 + 2 * l].bias))
        last = self.seq[1 + 2 * self.num_layers]
        p.extend((last.weight, last.bias))        # (the Parameter itself: its gradient is written straight into .grad)
        return p


class _ComponentVAEParams(nn.Module):
    def __init__(self, cfg, nout):
        super().__init__()
        self.ldim = cfg.comp_ldim
        self.encoder_module = _CompEncoderParams(cfg.img_size, cfg.comp_enc_channels, cfg.comp_ldim)
        self.decoder_module = _BroadcastDecoderParams(cfg.comp_ldim, nout, cfg.comp_dec_channels, cfg.comp_dec_layers)
        self.pixel_bound = False      # monet_config.py:68


class MONet(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.K_steps = cfg.K_steps
        self.img_size = cfg.img_size
        self.prior_mode = _cfg_get(cfg, 'prior_mode', 'softmax')
        self.mckl = _cfg_get(cfg, 'montecarlo_kl', True)
        self.debug = _cfg_get(cfg, 'debug', False)
        self.pixel_bound = _cfg_get(cfg, 'pixel_bound', True)
        if self.prior_mode not in ('softmax', 'scope'):
            raise ValueError('No valid prior mode.')          # monet_config.py:154-155
        filter_start = _cfg_get(cfg, 'filter_start', 32)
        core = _UNetParams(int(np.log2(cfg.img_size) - 1), cfg.img_size, filter_start, 4, 1, norm='in',
                           keep_final_conv=True)
        self.att_process = _SBPParams(core)
        self.comp_vae = _ComponentVAEParams(cfg, nout=4)
        std = _cfg_get(cfg, 'pixel_std2', 0.7) * torch.ones(1, 1, 1, 1, self.K_steps)
        std[0, 0, 0, 0, 0] = _cfg_get(cfg, 'pixel_std1', 0.7)
        self.register_buffer('std', std)
        self._std12 = (float(_cfg_get(cfg, 'pixel_std1', 0.7)), float(_cfg_get(cfg, 'pixel_std2', 0.7)))
        self._coords = {}

    def _canvas_coords(self, device):
        key = str(device)
        if key not in self._coords:
            d = self.img_size + 2 * self.comp_vae.decoder_module.num_layers
            self._coords[key] = pixel_coords(d).contiguous().to(device)
        return self._coords[key]

    def _decode(self, z):
        dm = self.comp_vae.decoder_module
        return fn.BroadcastDecoderFn.apply(z, self._canvas_coords(z.device), 'relu', None, *dm.flat_params())

    def _attention(self, x):
        """SimpleSBP.forward, modules/attention.py:31-51."""
        core = self.att_process.core
        log_s_k = [torch.zeros_like(x[:, :1])]
        log_m_k = []
        for step in range(self.K_steps - 1):
            feat = fn.UNetEncoderFn.apply(torch.cat((x, log_s_k[step]), 1), core.num_blocks, 0, *core.flat_params())
            a = fn.Conv1x1Fn.apply(feat, core.final_conv.weight, core.final_conv.bias)
            lm, ls = fn.SBPScanFn.apply(a.unsqueeze(0), log_s_k[step], False)     # one stick-breaking step, one launch
            # (views, not [0]: a select's backward is a zero fill + a copy per use)
            log_m_k.append(lm.view(lm.shape[1:]))
            log_s_k.append(ls.view(ls.shape[1:]))
        log_m_k.append(log_s_k[-1])
        return log_m_k, log_s_k

    def forward(self, x, eps=None):
        """x [B,3,H,W] on the GPU; eps [K*B, ldim] injects the rsample noise (component_vae.py:73)."""
        if x.is_cuda:
            autostep.arm(self)      # the unchanged train.py loop: this iteration on TrainStep's launch structure (autostep.py)
        B = x.shape[0]
        K, L = self.K_steps, self.comp_vae.ldim
        log_m_k, log_s_k = self._attention(x)
        log_m = torch.stack(log_m_k, 0)                               # [K,B,1,H,W]
        # --- ComponentVAE: K slots batched slot-major, mask as first channel (component_vae.py:59-66)
        em = self.comp_vae.encoder_module.module
        # (first layer: [log_m_k | x] stacked by its own kernel; only the mask channel carries a gradient)
        h = fn.MaskImageConvActFn.apply(log_m, x, em[0].weight, em[0].bias, 'relu')
        for i in (2, 4, 6):
            h = fn.DirectConvActFn.apply(h, em[i].weight, em[i].bias, 2, 1, 'relu', None)
        h = fn.linear(h.flatten(1), em[9].weight, em[9].bias, 'relu')
        enc_out = fn.linear(h, em[11].weight, em[11].bias)
        if eps is None:
            eps = torch.randn(K * B, L, device=x.device)
        # (mu | sigma_ps) -> z = mu + to_sigma(sigma_ps) eps and log q(z) in one launch
        z, mu, sigma, log_q = (t.view(K * B, -1) for t in fn.PosteriorFn.apply(enc_out.unsqueeze(1), eps.unsqueeze(0)))
        dec = self._decode(z)                                          # [K*B,4,H,W]
        err, recon, x_r = fn.MixtureWFn.apply(x, dec, log_m, K, self._std12[0], self._std12[1], bool(self.pixel_bound))
        # reconstructed masks (MONet.get_mask_recon_stack, monet_config.py:136-155): log_softmax over K of the logit
        # channel, or (prior_mode 'scope') a stick-breaking pass over the K logits whose last mask is the remaining scope
        logits = dec[:, 3:].reshape(K, B, 1, *x.shape[2:])
        if self.prior_mode == 'softmax':
            log_m_r = fn.LogSoftmaxKFn.apply(dec, K)
        else:
            log_m_r, _ = fn.SBPScanFn.apply(logits, None, True)
        losses = AttrDict()
        losses['err'] = err
        # Categorical KL between attention and reconstructed masks (monet_config.py:157-170)
        losses['kl_m'] = fn.CategoricalKLFn.apply(log_m, log_m_r)
        # KL of the component latents against N(0,1) (utils/misc.py:238-255): Monte-Carlo estimate at z, or
        # (montecarlo_kl off) the closed form of kl_divergence(Normal(mu, sigma), Normal(0, 1))
        if self.mckl:
            kl = fn.PriorLogPFn.apply(z.view(1, K * B, -1), None, log_q.view(1, K * B))    # [1, K*B]: log q - log N(0, 1)
        else:
            kl = (-torch.log(sigma) + 0.5 * (sigma * sigma + mu * mu) - 0.5).sum(1)
        kl = kl.view(K, B)
        losses['kl_l_k'] = SlotList(kl.unbind(0), stacked=kl)
        x_r_k = list(x_r.unbind(0))
        stats = LazyAttrDict(recon=recon, log_m_k=log_m_k, log_s_k=log_s_k, x_r_k=x_r_k, log_m_r_k=list(log_m_r.unbind(0)),
                             mx_r_k=Lazy(lambda: list((x_r * log_m.exp()).unbind(0))))      # (visualisation only: on first access)
        comp_stats = AttrDict(mu_k=mu.chunk(K, 0), sigma_k=sigma.chunk(K, 0), z_k=z.chunk(K, 0))
        return recon, losses, stats, AttrDict(), comp_stats

    def get_features(self, image_batch):
        with torch.no_grad():
            _, _, _, _, comp_stats = self.forward(image_batch)
            return torch.cat(comp_stats.z_k, dim=1)

    @torch.no_grad()
    def sample(self, batch_size, K_steps=None, eps=None):
        """models/monet_config.py:172-198; `eps` [B*K, ldim] injects the reference's one standard-normal draw (parity
        tests), default torch.randn."""
        K = self.K_steps if K_steps is None else K_steps
        dev = self.std.device
        z = torch.randn(batch_size * K, self.comp_vae.ldim, device=dev) if eps is None else eps.to(dev).contiguous()
        assert z.shape == (batch_size * K, self.comp_vae.ldim)
        dec = self._decode(z)
        x0 = torch.zeros(batch_size, 3, self.img_size, self.img_size, device=dev)
        _, gen_image, x_r, log_m_r = hip.mixture_fwd(x0, dec.contiguous(), K, 0.7, bool(self.pixel_bound))
        if self.prior_mode == 'scope':
            logits = dec[:, 3:].reshape(K, batch_size, 1, self.img_size, self.img_size).contiguous()
            log_m_r, _ = hip.sbp_scan_fwd(logits, None, True)
            gen_image = (log_m_r.exp() * x_r).sum(0)
        x_r_k, log_m_r_k = list(x_r.unbind(0)), list(log_m_r.unbind(0))
        stats = AttrDict(gen_image=gen_image, x_k=x_r_k, log_m_k=log_m_r_k,
                         mx_k=[x * m.exp() for x, m in zip(x_r_k, log_m_r_k)])
        return gen_image, stats
