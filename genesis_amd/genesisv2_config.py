"""GENESIS-V2 model config -- MI355X-native drop-in for the reference's
`models/genesisv2_config.py` (flags :35-42, `load(cfg)` :45-46, `GenesisV2` :49-256).

Same Forge-style contract: importing this file registers the model flags, `load(cfg)` returns an
`nn.Module` whose `forward(x) -> (recon, losses, stats, att_stats, comp_stats)` and
`sample(batch_size, K_steps)` match the reference's signatures, attribute names and `state_dict`
layout (78 tensors for feat_dim 64 / 64x64, identical keys / shapes / dtypes, `att_process.log_sigma`
fp64), so `train.py` and the visualise / compute_* scripts drop in unchanged.

Underneath, every spatial op group runs as hand-written gfx950 HIP kernels behind the C ABI
(include/genesis_hip.h) -- see genesis_amd/functions.py.  The torch.nn layer classes used below are
parameter containers only (they give the reference's parameter names, shapes and default
initialisation order); their own forward()s are never called for the conv / norm / attention path.
There is no CPU path: calling forward on CPU tensors raises.
"""
import math

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
from genesis_amd.lazy import Lazy, LazyAttrDict, SlotList  # noqa: E402
from genesis_amd import hip_ops as hip  # noqa: E402

# Architecture (models/genesisv2_config.py:35-42)
flags.DEFINE_integer('feat_dim', 64, 'Number of features and latents.')
# Segmentation
flags.DEFINE_string('kernel', 'gaussian', '{laplacian, gaussian, epanechnikov')
flags.DEFINE_boolean('semiconv', True, 'Use semi-convolutional embeddings.')
flags.DEFINE_boolean('dynamic_K', False, 'Dynamic K.')
# Auxiliary mask consistency loss
flags.DEFINE_boolean('klm_loss', False, 'KL mask regulariser.')
flags.DEFINE_boolean('detach_mr_in_klm', True, 'Detach reconstructed masks.')
# Flags the reference inherits from models/genesis_config.py:33-52 (imported by genesisv2_config.py:27)
flags.DEFINE_boolean('autoreg_prior', True, 'Autoregressive prior.')
flags.DEFINE_boolean('pixel_bound', True, 'Bound pixel values to [0, 1].')
flags.DEFINE_float('pixel_std1', 0.7, 'StdDev of reconstructed pixels.')
flags.DEFINE_float('pixel_std2', 0.7, 'StdDev of reconstructed pixels.')


# The AR prior's LSTM runs on the fused HIP cell kernels (functions.LSTMFn: dense input projection + one
# recurrent-GEMM/cell launch per step); there is no torch fallback.


def load(cfg):
    return GenesisV2(cfg)


def _cfg_get(cfg, name, default):
    try:
        return cfg[name]
    except (KeyError, TypeError):
        return getattr(cfg, name, default)


_LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


def _normal_log_prob(x, mu, sigma):
    """torch.distributions.Normal.log_prob without the constructor's argument validation (a device->host
    sync that cannot be captured in a HIP graph)."""
    if torch.is_tensor(sigma):
        return -((x - mu) ** 2) / (2 * sigma ** 2) - sigma.log() - _LOG_SQRT_2PI
    return -((x - mu) ** 2) / (2 * sigma ** 2) - math.log(sigma) - _LOG_SQRT_2PI


def pixel_coords(size):
    """Row grid then column grid, linspace(-1, 1): modules/blocks.py:42-47 ('ij' meshgrid)."""
    lin = torch.linspace(-1, 1, size)
    return torch.stack((lin.view(size, 1).expand(size, size), lin.view(1, size).expand(size, size)), 0).unsqueeze(0)


class _ConvGNReLU(nn.Sequential):
    """Parameter container with the key layout of modules/blocks.py:159-165 (ConvGNReLU) / :151-157 (ConvINReLU)."""

    def __init__(self, nin, nout, norm='gn'):
        layer = nn.GroupNorm(8, nout) if norm == 'gn' else nn.InstanceNorm2d(nout, affine=True)
        super().__init__(nn.Conv2d(nin, nout, 3, 1, 1, bias=False), layer, nn.ReLU(inplace=True))

    def params(self):
        return (self[0].weight, self[1].weight, self[1].bias)


class _UNetParams(nn.Module):
    """Parameter container with the key layout / init order of modules/unet.py:23-67."""

    def __init__(self, num_blocks, img_size, filter_start, in_chnls, out_chnls, norm='gn', keep_final_conv=False):
        super().__init__()
        c = filter_start
        if num_blocks == 4:
            enc_in, enc_out = [in_chnls, c, 2 * c, 2 * c], [c, 2 * c, 2 * c, 2 * c]
            dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c], [2 * c, 2 * c, c, c]
        elif num_blocks == 5:
            enc_in, enc_out = [in_chnls, c, c, 2 * c, 2 * c], [c, c, 2 * c, 2 * c, 2 * c]
            dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c, 2 * c], [2 * c, 2 * c, c, c, c]
        elif num_blocks == 6:
            enc_in, enc_out = [in_chnls, c, c, c, 2 * c, 2 * c], [c, c, c, 2 * c, 2 * c, 2 * c]
            dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c, 2 * c, 2 * c], [2 * c, 2 * c, c, c, c, c]
        else:
            raise ValueError('UNet supports 4, 5 or 6 blocks (img_size 32, 64, 128)')
        self.num_blocks = num_blocks
        self.down = nn.ModuleList([_ConvGNReLU(i, o, norm) for i, o in zip(enc_in, enc_out)])
        self.up = nn.ModuleList([_ConvGNReLU(i, o, norm) for i, o in zip(dec_in, dec_out)])
        self.featuremap_size = img_size // 2 ** (num_blocks - 1)
        flat = 2 * c * self.featuremap_size ** 2
        self.mlp = nn.Sequential(nn.Flatten(), nn.Linear(flat, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(),
                                 nn.Linear(128, flat), nn.ReLU())
        # the reference builds final_conv (consuming init RNG) and then replaces it by Identity
        # (models/genesisv2_config.py:70)
        final = nn.Conv2d(c, out_chnls, 1)
        self.final_conv = final if keep_final_conv else nn.Identity()

    def flat_params(self):
        p = []
        for blk in list(self.down) + list(self.up):
            p.extend(blk.params())
        for j in (1, 3, 5):
            p.extend((self.mlp[j].weight, self.mlp[j].bias))
        return p


class _ScalarGate(nn.Module):
    def __init__(self, init=0.0):
        super().__init__()
        self.gate = nn.Parameter(torch.tensor(init))


class _SemiConvParams(nn.Module):
    def __init__(self, nin, nout):
        super().__init__()
        self.conv = nn.Conv2d(nin, nout, 1)
        self.gate = _ScalarGate()


class _ICSBPParams(nn.Module):
    """modules/attention.py:138-160."""

    def __init__(self, kernel, K_steps, feat_dim, semiconv, colour_dim=8):
        super().__init__()
        if kernel == 'laplacian':
            sigma_init = 1.0 / (np.sqrt(K_steps) * np.log(2))
        elif kernel == 'gaussian':
            sigma_init = 1.0 / (K_steps * np.log(2))
        elif kernel == 'epanechnikov':
            sigma_init = 2.0 / K_steps
        else:
            raise ValueError('No valid kernel.')
        self.kernel = kernel
        self.colour_dim = colour_dim
        self.log_sigma = nn.Parameter(torch.tensor(sigma_init).log())  # float64 0-dim, as in the reference
        self.semiconv = semiconv
        if semiconv:
            self.colour_head = _SemiConvParams(feat_dim, colour_dim)
        else:
            self.colour_head = nn.Conv2d(feat_dim, colour_dim, 1)


class GenesisV2(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.K_steps = cfg.K_steps
        self.img_size = cfg.img_size
        self.pixel_bound = _cfg_get(cfg, 'pixel_bound', True)
        self.feat_dim = cfg.feat_dim
        self.klm_loss = _cfg_get(cfg, 'klm_loss', False)
        self.detach_mr_in_klm = _cfg_get(cfg, 'detach_mr_in_klm', True)
        self.dynamic_K = _cfg_get(cfg, 'dynamic_K', False)
        self.debug = _cfg_get(cfg, 'debug', False)
        self.multi_gpu = _cfg_get(cfg, 'multi_gpu', False)
        D = cfg.feat_dim
        self.encoder = _UNetParams(int(np.log2(cfg.img_size) - 1), cfg.img_size, min(D, 64), 3, D)
        self.att_process = _ICSBPParams(_cfg_get(cfg, 'kernel', 'gaussian'), self.K_steps, D,
                                        _cfg_get(cfg, 'semiconv', True))
        self.seg_head = _ConvGNReLU(D, D)
        self.feat_head = nn.Sequential(_ConvGNReLU(D, D), nn.Conv2d(D, 2 * D, 1))
        self.z_head = nn.Sequential(nn.LayerNorm(2 * D), nn.Linear(2 * D, 2 * D), nn.ReLU(inplace=True),
                                    nn.Linear(2 * D, 2 * D))
        c = D
        cm = min(c, 64)
        self.decoder_module = nn.Sequential(
            nn.Identity(),  # BroadcastLayer(img_size // 16): no parameters (modules/blocks.py:104-117)
            nn.ConvTranspose2d(D + 2, c, 5, 2, 2, 1), nn.GroupNorm(8, c), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(c, c, 5, 2, 2, 1), nn.GroupNorm(8, c), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(c, cm, 5, 2, 2, 1), nn.GroupNorm(8, cm), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(cm, cm, 5, 2, 2, 1), nn.GroupNorm(8, cm), nn.ReLU(inplace=True),
            nn.Conv2d(cm, 4, 1))
        self.autoreg_prior = _cfg_get(cfg, 'autoreg_prior', True)
        self.prior_lstm, self.prior_linear = None, None
        if self.autoreg_prior and self.K_steps > 1:
            self.prior_lstm = nn.LSTM(D, 4 * D)
            self.prior_linear = nn.Linear(4 * D, 2 * D)
        assert _cfg_get(cfg, 'pixel_std1', 0.7) == _cfg_get(cfg, 'pixel_std2', 0.7)
        self.std = _cfg_get(cfg, 'pixel_std1', 0.7)
        # coordinate grids are plain attributes, not buffers (modules/blocks.py:123-126,172-174):
        # they never enter the state_dict.  Cached per device.
        self._grids = {}

    # ------------------------------------------------------------------ helpers
    def _grid(self, device):
        key = str(device)
        if key not in self._grids:
            S, cd = self.img_size, self.att_process.colour_dim
            uv = torch.cat((torch.zeros(1, cd - 2, S, S), pixel_coords(S)), 1)[0].contiguous()
            dec_coords = pixel_coords(self.img_size // 16).contiguous()
            self._grids[key] = (uv.to(device), dec_coords.to(device))
        return self._grids[key]

    def _decoder_params(self):
        p = []
        for ci, gi in ((1, 2), (4, 5), (7, 8), (10, 11)):
            p.extend((self.decoder_module[ci].weight, self.decoder_module[ci].bias,
                      self.decoder_module[gi].weight, self.decoder_module[gi].bias))
        p.extend((self.decoder_module[13].weight, self.decoder_module[13].bias))
        return p

    def _decode(self, z_kbd, x=None, grad_through_masks=False):
        """z [K,B,D] -> (dec [K*B,4,H,W]) -> mixture.  Returns (err, recon, x_r [K,...], log_m_r [K,...]).
        grad_through_masks: log_m_r carries a gradient back into the decoder (the mixture kernel's own outputs are
        non-differentiable by-products)."""
        K, B, D = z_kbd.shape
        _, dec_coords = self._grid(z_kbd.device)
        dec = fn.DecoderFn.apply(z_kbd.reshape(K * B, D), dec_coords, *self._decoder_params())
        if x is None:
            x = torch.zeros(B, 3, self.img_size, self.img_size, device=z_kbd.device)
        err, recon, x_r, log_m_r = fn.MixtureFn.apply(x, dec, K, float(self.std), bool(self.pixel_bound))
        if grad_through_masks:
            log_m_r = fn.MaskReconFn.apply(dec, log_m_r)
        return err, recon, x_r, log_m_r

    def _prior_hidden(self, z_kbd):
        """LSTM of the AR prior from the zero state over z_1..z_{K-1} (models/genesis_config.py:297-307)."""
        K, B, D = z_kbd.shape
        L = self.prior_lstm
        return fn.LSTMFn.apply(z_kbd[:-1], L.weight_ih_l0, L.weight_hh_l0, L.bias_ih_l0, L.bias_hh_l0)

    def _component_kl(self, z, log_q):
        """[K,B]: log q(z_k|x) - log p(z_k|z_<k) per slot (Genesis.mask_latent_loss, models/genesis_config.py:288-343)."""
        lin = None
        if self.prior_lstm is not None:
            if z.shape[0] > 1:
                L, P = self.prior_lstm, self.prior_linear
                return fn.ARPriorKLFn.apply(z, log_q, L.weight_ih_l0, L.weight_hh_l0, L.bias_ih_l0, L.bias_hh_l0,
                                            P.weight, P.bias)
            lin = fn.linear(self._prior_hidden(z), self.prior_linear.weight, self.prior_linear.bias)  # [K-1,B,2D]
        return fn.PriorLogPFn.apply(z, lin, log_q)

    # ------------------------------------------------------------------ forward
    noise = None      # TrainStep installs its noise source here: (uniform shape, normal shape, device) -> (rand_pixel, eps)

    def forward(self, x, rand_pixel=None, eps=None, seed_idx=None):
        """x [B,3,H,W] in [0,1] on the GPU.  The optional arguments inject the noise the reference draws
        internally (rand_pixel [B,1,H,W] uniform, modules/attention.py:177-178; eps [K,B,D] standard
        normal, models/genesisv2_config.py:157) and, for tie-break tests, the seed pixels [K-1,B]."""
        if x.is_cuda:
            if rand_pixel is None and eps is None and seed_idx is None:
                # the unchanged train.py loop, third iteration on: the forward pass as ONE replayed HIP graph (autostep.py)
                t = autostep.graph_forward(self, x)
                if t is not None:
                    return self._assemble(t)
            autostep.arm(self, x if (rand_pixel is None and eps is None and seed_idx is None) else None)      # the unchanged train.py loop: this iteration on TrainStep's launch structure (autostep.py)
        return self._assemble(self._compute(x, rand_pixel, eps, seed_idx))

    def _compute(self, x, rand_pixel=None, eps=None, seed_idx=None):
        """The tensors of a forward pass (everything that launches kernels): a dict for _assemble."""
        B, _, H, W = x.shape
        K, D = self.K_steps, self.feat_dim
        dev = x.device
        uv, _ = self._grid(dev)
        # --- Extract features (F.relu on the ReLU'd UNet output, genesisv2_config.py:115, is the identity)
        enc_feat = fn.UNetEncoderFn.apply(x, self.encoder.num_blocks, 8, *self.encoder.flat_params())
        # --- Predict attention masks
        if rand_pixel is None and eps is None and getattr(self, 'noise', None) is not None:
            # a training step's noise source (TrainStep: one Philox launch for both tensors, replayable in its HIP graph)
            rand_pixel, eps = self.noise((B, 1, H, W), (K, B, D), dev)
        if rand_pixel is None:
            rand_pixel = torch.rand(B, 1, H, W, device=dev)
        ap = self.att_process
        if ap.semiconv:
            cw, cb, gate, addend = ap.colour_head.conv.weight, ap.colour_head.conv.bias, ap.colour_head.gate.gate, uv
        else:
            cw, cb, gate, addend = ap.colour_head.weight, ap.colour_head.bias, None, None
        seg_params = self.seg_head.params()
        # dynamic_K (genesisv2_config.py:118-137, attention.py:218-219): an image stops at the first step whose mask
        # would hold fewer than 20 pixels; the kernel does that per image of the batch in one launch
        min_mass = 20.0 if self.dynamic_K else 0.0
        f = None
        if fn.heads_pairable(enc_feat, seg_params[0], cw, self.feat_head[0].params()[0]):
            # seg_head and feat_head[0] read the same features: one conv launch forward, one data-gradient launch backward
            res = fn.SegFeatHeadsFn.apply(enc_feat, *seg_params, cw, cb, gate, addend, ap.log_sigma, rand_pixel, K,
                                          ap.kernel, seed_idx, min_mass, *self.feat_head[0].params())
            res, f = res[:-1], res[-1]
        elif fn.seg_head_fusable(enc_feat, seg_params[0], cw):
            res = fn.SegICSBPFn.apply(enc_feat, *seg_params, cw, cb, gate, addend, ap.log_sigma, rand_pixel, K, ap.kernel,
                                      seed_idx, min_mass)
        else:
            seg = fn.ConvGNReLUFn.apply(enc_feat, *seg_params)
            res = fn.ICSBPFn.apply(seg, cw, cb, gate, addend, ap.log_sigma, rand_pixel, K, ap.kernel, seed_idx, min_mass)
        log_m, log_s, colour, seeds, idx = res[:5]
        dyn_batched = False
        if self.dynamic_K:
            if B == 1:
                # one image: the reference's lists simply end early -- K shrinks for the rest of the forward pass
                # (a host read of the step count; the reference syncs at every step's `< 20` test)
                n = int(res[5][0])
                K = n + 1
                log_m, log_s = log_m[:K], log_s[:K]
                seeds, idx = seeds[:min(n + 1, self.K_steps - 1)], idx[:min(n + 1, self.K_steps - 1)]
                if eps is not None:
                    eps = eps[:K]
            else:
                dyn_batched = True      # masks of finished images are padded with -1e10; no att_stats / log_s_k (:122)
        # --- Object features: feat_head[0] once (the reference recomputes it K times, :149), pooled per
        #     slot; the 1x1 conv feat_head[1] commutes with the masked sum and is applied to the pooled sums.
        if f is None:
            f = fn.ConvGNReLUFn.apply(enc_feat, *self.feat_head[0].params())
        S, msum = fn.MaskPoolFn.apply(f, log_m)                      # [B,K,D], [B,K]
        ln = self.z_head[0]
        zh = fn.PooledHeadFn.apply(fn.linear(S, self.feat_head[1].weight), msum, self.feat_head[1].bias,
                                   ln.weight, ln.bias, ln.eps)              # (pooled / mask mass) -> LayerNorm
        # --- Posterior
        if eps is None:
            eps = torch.randn(K, B, D, device=dev)
        zh = fn.linear(zh, self.z_head[1].weight, self.z_head[1].bias, 'relu')
        zh = fn.linear(zh, self.z_head[3].weight, self.z_head[3].bias)
        z, mu, sigma, log_q = fn.PosteriorFn.apply(zh, eps)                  # [K,B,D] x3, [K,B]
        # --- Component KL (Genesis.mask_latent_loss, models/genesis_config.py:288-343); optionally forked onto the
        #     side stream so that its chain of tiny kernels runs beside the decoder (step_state().side_prior)
        forked = fn.step_state().side_prior and self.prior_lstm is not None and torch.is_grad_enabled()
        if forked:
            with fn.side_branch(z, log_q):
                kl = self._component_kl(z, log_q)
        # --- Decode latents, reconstruction loss
        err, recon, x_r, log_m_r = self._decode(z, x, self.klm_loss and not self.detach_mr_in_klm)
        if forked:
            fn.join_branch()
        else:
            kl = self._component_kl(z, log_q)
        t = dict(err=err, kl=kl, recon=recon, log_m=log_m, log_s=log_s, x_r=x_r, log_m_r=log_m_r, colour=colour, seeds=seeds,
                 idx=idx, mu=mu, sigma=sigma, z=z)
        # -- Optional: Attention mask loss (MONet.kl_m_loss, models/monet_config.py:157-170)
        if self.klm_loss:
            # genesisv2_config.py:172-176: the reconstructed masks are detached unless detach_mr_in_klm is off
            t['kl_m'] = fn.CategoricalKLFn.apply(log_m, log_m_r.detach() if self.detach_mr_in_klm else log_m_r)
        t['dyn_batched'] = dyn_batched
        return t

    def _assemble(self, t):
        """The reference's five return values (models/genesisv2_config.py:184-203) from the tensors of _compute: views, lists and
        lazily evaluated visualisation outputs -- no arithmetic."""
        err, kl, recon, log_m, log_s, x_r, log_m_r = t['err'], t['kl'], t['recon'], t['log_m'], t['log_s'], t['x_r'], t['log_m_r']
        colour, seeds, idx, mu, sigma, z, dyn_batched = t['colour'], t['seeds'], t['idx'], t['mu'], t['sigma'], t['z'], t['dyn_batched']
        ap = self.att_process
        uv, _ = self._grid(recon.device)
        losses = AttrDict()
        losses['err'] = err
        log_m_k = list(log_m.unbind(0))
        log_s_k = None if dyn_batched else list(log_s.unbind(0))
        x_r_k = list(x_r.unbind(0))
        log_m_r_k = list(log_m_r.unbind(0))
        if 'kl_m' in t:
            losses['kl_m'] = t['kl_m']
        losses['kl_l_k'] = SlotList(kl.unbind(0), stacked=kl)

        # derived visualisation outputs are evaluated on first access (a training step never reads them)
        stats = LazyAttrDict(
            recon=recon, log_m_k=log_m_k, log_s_k=log_s_k, x_r_k=x_r_k, log_m_r_k=log_m_r_k,
            mx_r_k=Lazy(lambda: list((x_r * log_m_r.exp()).unbind(0))),
            instance_seg=Lazy(lambda: torch.argmax(log_m.squeeze(2), dim=0)),
            instance_seg_r=Lazy(lambda: torch.argmax(log_m_r.squeeze(2), dim=0)))
        att_stats = LazyAttrDict()
        att_stats.update({'colour': colour,
                          'delta': Lazy(lambda: colour[:, -2:] - uv[-2:]) if ap.semiconv else None,
                          'seeds': list(seeds.unbind(0)), 'seed_idx': list(idx.unbind(0))})
        comp_stats = AttrDict(mu_k=list(mu.unbind(0)), sigma_k=list(sigma.unbind(0)), z_k=list(z.unbind(0)),
                              kl_l_k=[], q_z_k=[Normal(m, s, validate_args=False)
                                                for m, s in zip(mu.unbind(0), sigma.unbind(0))])
        if self.multi_gpu:
            del comp_stats['q_z_k']
        if dyn_batched:
            att_stats = None             # genesisv2_config.py:122
        return recon, losses, stats, att_stats, comp_stats

    def decode_latents(self, z_k):
        """models/genesisv2_config.py:205-225."""
        _, recon, x_r, log_m_r = self._decode(torch.stack(list(z_k), 0))
        return recon, list(x_r.unbind(0)), list(log_m_r.unbind(0))

    @torch.no_grad()
    def sample(self, batch_size, K_steps=None, eps=None):
        """models/genesisv2_config.py:227-256: ancestral rollout of the AR prior (LSTM cell -> prior_linear ->
        N(tanh, to_prior_sigma) sample per slot), then decode.  The rollout runs on the training path's kernels
        (gx_linear_fwd, gx_lstm_step_fwd, gx_latent_prior_sample); `eps` [K,B,D] injects the standard-normal draws
        (parity tests: the reference draws them with Normal.sample in this order), default torch.randn."""
        K_steps = self.K_steps if K_steps is None else K_steps
        dev = self.decoder_module[13].weight.device
        D = self.feat_dim
        if eps is None:
            eps = torch.randn(K_steps, batch_size, D, device=dev)
        eps = eps.to(dev).contiguous()
        assert eps.shape == (K_steps, batch_size, D)
        if self.autoreg_prior and self.prior_lstm is not None:
            z_k = [eps[0]]
            w_ih, w_hh = self.prior_lstm.weight_ih_l0, self.prior_lstm.weight_hh_l0
            b_ih, b_hh = self.prior_lstm.bias_ih_l0, self.prior_lstm.bias_hh_l0
            H = w_hh.shape[1]
            h_prev = c_prev = None
            for k in range(1, K_steps):
                gx = hip.linear_fwd(z_k[-1].contiguous(), w_ih, b_ih)                  # [B, 4H]
                act = torch.empty(batch_size, 4 * H, device=dev)
                c = torch.empty(batch_size, H, device=dev)
                h = torch.empty(batch_size, H, device=dev)
                hip.lstm_step_fwd(gx, h_prev, c_prev, w_hh, b_hh, act, c, h)
                lin = hip.linear_fwd(h, self.prior_linear.weight, self.prior_linear.bias)   # [B, 2D]
                z_k.append(hip.latent_prior_sample(lin, eps[k]))
                h_prev, c_prev = h, c
        else:
            z_k = list(eps.unbind(0))
        recon, x_r_k, log_m_r_k = self.decode_latents(z_k)
        stats = AttrDict(x_k=x_r_k, log_m_k=log_m_r_k, mx_k=[x * m.exp() for x, m in zip(x_r_k, log_m_r_k)],
                         z_k=z_k)
        return recon, stats

    def get_features(self, image_batch):
        """models/genesis_config.py:427-436 (same accessor the other model configs expose)."""
        with torch.no_grad():
            _, _, _, _, comp_stats = self.forward(image_batch)
            return torch.cat(comp_stats.z_k, dim=1)
