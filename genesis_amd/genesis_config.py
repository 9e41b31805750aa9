"""GENESIS (v1) model config (BASELINE config 3) -- mirror of the reference's `models/genesis_config.py` (flags
:33-52, `load(cfg)` :55-56, `Genesis` :59-436) on the HIP path for the default configuration (two_stage,
autoreg_prior, comp_prior, comp_symmetric=False, K_steps > 1): same `state_dict` (root buffer `std`,
`att_process.core.*` sylvester VAE with BatchNorm / InstanceNorm, `att_process.{lstm,linear}`, `comp_vae.*`,
`prior_lstm`, `prior_linear`, `prior_mlp`), same forward 5-tuple with losses {err, kl_m_k, kl_l_k}.

HIP mapping: LatentSBP's gated-conv attention VAE (modules/attention.py:84-133) runs on the direct (de)conv +
fused gated-norm kernels (genesis_amd/sylvester.py); the ComponentVAE (ELU) on the direct-conv encoder kernels and
the MFMA BroadcastDecoder canvas; the mixture likelihood on gx_mixture_w_* (RGB-only decoder output, attention
masks as mixing weights).  LSTMs / Linears / pointwise stick-breaking are library or torch ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from genesis_amd import autostep
from genesis_amd import compat as _compat

_compat.install()

from attrdict import AttrDict  # noqa: E402
from forge import flags  # noqa: E402

from genesis_amd import functions as fn  # noqa: E402
from genesis_amd.genesisv2_config import _cfg_get, _normal_log_prob, pixel_coords  # noqa: E402
from genesis_amd.lazy import Lazy, LazyAttrDict, SlotList  # noqa: E402
from genesis_amd.monet_config import _BroadcastDecoderParams, _ComponentVAEParams  # noqa: E402
from genesis_amd.sylvester import (GatedConv2d, GatedConvTranspose2d, SylvesterVAE, gc_decoder_forward,  # noqa: E402
                                  gc_encoder_forward)

SYM_STRIDES = [1, 2, 1, 2, 1]           # genesis_config.py:109,118

# models/genesis_config.py:33-52
flags.DEFINE_boolean('two_stage', True, 'Use two stages if two, else only one.')
flags.DEFINE_boolean('autoreg_prior', True, 'Autoregressive prior.')
flags.DEFINE_boolean('comp_prior', True, 'Component prior.')
flags.DEFINE_integer('attention_latents', 64, 'Latent dimension.')
flags.DEFINE_string('enc_norm', 'bn', '{bn, in} - norm type in encoder.')
flags.DEFINE_string('dec_norm', 'bn', '{bn, in} - norm type in decoder.')
flags.DEFINE_integer('comp_enc_channels', 32, 'Starting number of channels.')
flags.DEFINE_integer('comp_ldim', 16, 'Latent dimension of the VAE.')
flags.DEFINE_integer('comp_dec_channels', 32, 'Num channels in Broadcast Decoder.')
flags.DEFINE_integer('comp_dec_layers', 4, 'Num layers in Broadcast Decoder.')
flags.DEFINE_boolean('comp_symmetric', False, 'Use same encoder/decoder as in attention VAE.')
flags.DEFINE_boolean('pixel_bound', True, 'Bound pixel values to [0, 1].')
flags.DEFINE_float('pixel_std1', 0.7, 'StdDev of reconstructed pixels.')
flags.DEFINE_float('pixel_std2', 0.7, 'StdDev of reconstructed pixels.')
flags.DEFINE_boolean('montecarlo_kl', True, 'Evaluate KL via MC samples.')


def load(cfg):
    return Genesis(cfg)


class _LatentSBPParams(nn.Module):
    """attention.LatentSBP (modules/attention.py:77-82): core VAE + LSTM(z+256 -> 2z) + Linear(2z -> 2z)."""

    def __init__(self, core):
        super().__init__()
        self.core = core
        self.lstm = nn.LSTM(core.z_size + 256, 2 * core.z_size)
        self.linear = nn.Linear(2 * core.z_size, 2 * core.z_size)


class Genesis(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.K_steps = cfg.K_steps
        self.img_size = cfg.img_size
        self.two_stage = _cfg_get(cfg, 'two_stage', True)
        self.autoreg_prior = _cfg_get(cfg, 'autoreg_prior', True)
        self.comp_prior = _cfg_get(cfg, 'comp_prior', True)
        self.ldim = _cfg_get(cfg, 'attention_latents', 64)
        self.pixel_bound = _cfg_get(cfg, 'pixel_bound', True)
        self.debug = _cfg_get(cfg, 'debug', False)
        # genesis_config.py:71-75: the component prior exists only in the two-stage model
        self.comp_prior = bool(self.two_stage and self.K_steps > 1 and self.comp_prior)
        if not _cfg_get(cfg, 'montecarlo_kl', True):
            raise AssertionError('ALWAYS use MC for estimating KL')            # genesis_config.py:82
        if not self.autoreg_prior:
            # the reference's forward passes self.prior_lstm unconditionally (genesis_config.py:214-216): with
            # autoreg_prior off that attribute does not exist and its first forward raises
            raise AttributeError("'Genesis' object has no attribute 'prior_lstm' (autoreg_prior=False is not runnable "
                                 'in the reference either)')
        if self.K_steps <= 1:
            raise NotImplementedError('Genesis HIP path: K_steps > 1 (the reference\'s train.py cannot aggregate the '
                                      '0-dim kl_m of its K_steps == 1 branch either)')
        self.comp_symmetric = bool(_cfg_get(cfg, 'comp_symmetric', False)) and bool(self.two_stage)
        att_core = SylvesterVAE(self.ldim, [3, cfg.img_size, cfg.img_size], 1, _cfg_get(cfg, 'enc_norm', 'bn'),
                                _cfg_get(cfg, 'dec_norm', 'bn'))
        self.att_steps = self.K_steps
        self.att_process = _LatentSBPParams(att_core)
        if self.two_stage:
            self.comp_vae = _ComponentVAEParams(cfg, nout=3)
            self.comp_vae.pixel_bound = self.pixel_bound
            self._dec_layers = self.comp_vae.decoder_module.num_layers
            if self.comp_symmetric:
                # genesis_config.py:104-123: encoder / decoder of the component VAE REPLACED (after construction: same
                # RNG consumption, same state_dict positions) by the attention VAE's gated-conv stacks
                en, dn = _cfg_get(cfg, 'enc_norm', 'bn'), _cfg_get(cfg, 'dec_norm', 'bn')
                k = att_core.last_kernel_size
                enc = [GatedConv2d(i, o, 5, s_, 2, en, en)
                       for i, o, s_ in zip([4, 32, 32, 64, 64], [32, 32, 64, 64, 64], SYM_STRIDES)]
                enc.append(GatedConv2d(64, 2 * cfg.comp_ldim, k, 1, 0))
                self.comp_vae.encoder_module = nn.Sequential(nn.Sequential(*enc), nn.Flatten())
                dec = [GatedConvTranspose2d(cfg.comp_ldim, 64, k, 1, 0)]
                dec += [GatedConvTranspose2d(i, o, 5, s_, 2, s_ - 1, dn, dn)
                        for i, o, s_ in zip([64, 64, 32, 32, 32], [64, 32, 32, 32, 32], SYM_STRIDES)]
                self.comp_vae.decoder_module = nn.Sequential(nn.Identity(), nn.Sequential(*dec), nn.Conv2d(32, 3, 1))
        else:
            # one stage (genesis_config.py:124-129): components decoded from the attention latents
            self.decoder = _BroadcastDecoderParams(self.ldim, 3, _cfg_get(cfg, 'comp_dec_channels', 32),
                                                   _cfg_get(cfg, 'comp_dec_layers', 4))
            self._dec_layers = self.decoder.num_layers
        self.prior_lstm = nn.LSTM(self.ldim, 256)
        self.prior_linear = nn.Linear(256, 2 * self.ldim)
        if self.comp_prior:
            self.prior_mlp = nn.Sequential(nn.Linear(self.ldim, 256), nn.ELU(), nn.Linear(256, 256), nn.ELU(),
                                           nn.Linear(256, 2 * cfg.comp_ldim))
        std = _cfg_get(cfg, 'pixel_std2', 0.7) * torch.ones(1, 1, 1, 1, self.K_steps)
        std[0, 0, 0, 0, 0] = _cfg_get(cfg, 'pixel_std1', 0.7)
        self.register_buffer('std', std)
        self._std12 = (float(_cfg_get(cfg, 'pixel_std1', 0.7)), float(_cfg_get(cfg, 'pixel_std2', 0.7)))
        self._coords = {}

    def _canvas_coords(self, device):
        key = str(device)
        if key not in self._coords:
            d = self.img_size + 2 * self._dec_layers
            self._coords[key] = pixel_coords(d).contiguous().to(device)
        return self._coords[key]

    def _attention(self, x, eps_m):
        """LatentSBP.forward (modules/attention.py:84-133) + the K+1 -> K mask fix-up (genesis_config.py:167-169)."""
        K, B = self.K_steps, x.shape[0]
        ap, core = self.att_process, self.att_process.core
        h = core.encode_features(x)
        # the first posterior N(q_z_mean(h), to_var(q_z_var(h))) and the K - 1 recurrent ones (the sampled z is fed back through
        # lstm(cat(h, z)) -> linear: one cell per step) as one autograd node; sqrt(to_var(x)) == to_sigma(x) = softplus(x + 0.5)
        # + 1e-8 (blocks.py:22-26)
        L = ap.lstm
        if not torch.is_tensor(eps_m):
            eps_m = torch.stack(list(eps_m), 0)
        z3, mu3, sig3, log_q = fn.LatentSBPPosteriorFn.apply(
            h, eps_m, core.q_z_mean.weight, core.q_z_mean.bias, core.q_z_var[0].weight, core.q_z_var[0].bias,
            L.weight_ih_l0, L.weight_hh_l0, L.bias_ih_l0, L.bias_hh_l0, ap.linear.weight, ap.linear.bias)
        mu_k, sigma_k, z_k = list(mu3.unbind(0)), list(sig3.unbind(0)), list(z3.unbind(0))
        z = z3.view(K * B, -1)
        logits = core.decode(z).view(K, B, 1, self.img_size, self.img_size)
        # K stick-breaking steps in one launch; the last mask is the remaining scope (genesis_config.py:167-169)
        log_m, log_s = fn.SBPScanFn.apply(logits, None, True)
        log_m_k = list(log_m.unbind(0))
        log_s_k = Lazy(lambda: [torch.zeros_like(x[:, :1])] + list(log_s.unbind(0)))     # (a returned statistic only)
        return log_m, log_m_k, log_s_k, mu_k, sigma_k, z_k, z3, log_q

    def _prior_m(self, z_kbd):
        L = self.prior_lstm
        out = fn.LSTMFn.apply(z_kbd[:-1], L.weight_ih_l0, L.weight_hh_l0, L.bias_ih_l0, L.bias_hh_l0)
        mu_raw, sig_raw = fn.linear(out, self.prior_linear.weight, self.prior_linear.bias).chunk(2, dim=2)
        return torch.tanh(mu_raw), torch.sigmoid(sig_raw + 4.0) + 1e-4

    def forward(self, x, eps_m=None, eps_c=None):
        """x [B,3,S,S] on the GPU.  eps_m: K x [B, ldim], eps_c: [K*B, comp_ldim] inject the rsample noise."""
        if x.is_cuda:
            autostep.arm(self)      # the unchanged train.py loop: this iteration on TrainStep's launch structure (autostep.py)
        if not x.is_cuda:
            from genesis_amd._lib import GenesisHipError
            raise GenesisHipError('Genesis: the HIP path needs device tensors; there is no CPU fallback')
        B, K = x.shape[0], self.K_steps
        if eps_m is None:
            eps_m = torch.randn(K, B, self.ldim, device=x.device)
        log_m, log_m_k, log_s_k, mu_k, sigma_k, z_k, z, log_q = self._attention(x, eps_m)
        if self.two_stage:
            # --- ComponentVAE (ELU), slot-major batch, mask as first channel
            Lc = self.comp_vae.ldim
            if self.comp_symmetric:
                inp = torch.cat((log_m.flatten(0, 1), x.repeat(K, 1, 1, 1)), 1)
                enc_out = gc_encoder_forward(self.comp_vae.encoder_module[0], inp, SYM_STRIDES, self.training)
            else:
                em = self.comp_vae.encoder_module.module
                # (first layer: [log_m_k | x] stacked by its own kernel; only the mask channel carries a gradient)
                h = fn.MaskImageConvActFn.apply(log_m, x, em[0].weight, em[0].bias, 'elu')
                for i in (2, 4, 6):
                    h = fn.DirectConvActFn.apply(h, em[i].weight, em[i].bias, 2, 1, 'elu', None)
                h = fn.linear(h.flatten(1), em[9].weight, em[9].bias, 'elu')
                enc_out = fn.linear(h, em[11].weight, em[11].bias)
            if eps_c is None:
                eps_c = torch.randn(K * B, Lc, device=x.device)
            # (mu | sigma_ps) -> z_c = mu + to_sigma(sigma_ps) eps and log q(z_c) in one launch
            z_c, mu_c, sig_c, log_q_c = (t.view(K * B, -1) for t in fn.PosteriorFn.apply(enc_out.unsqueeze(1), eps_c.unsqueeze(0)))
            dm, z_dec = self.comp_vae.decoder_module, z_c
        else:
            # --- one stage (genesis_config.py:183-191): the components come from the attention latents
            dm, z_dec = self.decoder, z.flatten(0, 1)
        if self.two_stage and self.comp_symmetric:
            h = gc_decoder_forward(dm[1], z_dec, SYM_STRIDES, self.training)
            dec = fn.Conv1x1Fn.apply(h, dm[2].weight, dm[2].bias)
        else:
            dec = fn.BroadcastDecoderFn.apply(z_dec, self._canvas_coords(x.device), 'elu', None, *dm.flat_params())   # [K*B,3,S,S]
        err, recon, x_r = fn.MixtureWFn.apply(x, dec, log_m, K, self._std12[0], self._std12[1], bool(self.pixel_bound))
        losses = AttrDict()
        losses['err'] = err
        # -- Attention mask KL (mask_latent_loss, genesis_config.py:288-343)
        Lm, Pm = self.prior_lstm, self.prior_linear
        if K > 1:
            # LSTM -> linear -> log q - log p as one autograd node (shared with GENESIS-V2)
            kl_m, lin_p = fn.ARPriorKLFn.apply(z, log_q, Lm.weight_ih_l0, Lm.weight_hh_l0, Lm.bias_ih_l0, Lm.bias_hh_l0,
                                               Pm.weight, Pm.bias, True)
        else:
            kl_m, lin_p = fn.PriorLogPFn.apply(z, None, log_q), None
        losses['kl_m_k'] = SlotList(kl_m.unbind(0), stacked=kl_m)
        comp_stats = None
        if self.two_stage:
            if self.comp_prior:
                # -- Component KL with the learned component prior (genesis_config.py:229-247)
                pm_ = self.prior_mlp
                o = fn.linear(z.view(K * B, -1), pm_[0].weight, pm_[0].bias, 'elu')
                o = fn.linear(o, pm_[2].weight, pm_[2].bias, 'elu')
                o = fn.linear(o, pm_[4].weight, pm_[4].bias)              # [K*B, 2*Lc], slot-major like z_c
                kl_l = fn.PriorLogPFn.apply(z_c.view(K, B, -1), o.view(K, B, -1), log_q_c.view(K, B), True)
                o_d = o.detach()
            else:
                # -- N(0, 1) component prior (genesis_config.py:248-254): every row is a "first slot"
                kl_l = fn.PriorLogPFn.apply(z_c.view(1, K * B, -1), None, log_q_c.view(1, K * B))
            kl_l = kl_l.view(K, B)
            losses['kl_l_k'] = SlotList(kl_l.unbind(0), stacked=kl_l)
            # (the priors' means / scales are returned statistics only: evaluated on first access, a training step reads none)
            comp_stats = LazyAttrDict(mu_k=mu_c.chunk(K, 0), sigma_k=sig_c.chunk(K, 0), z_k=z_c.chunk(K, 0))
            if self.comp_prior:
                comp_stats.update({
                    'pmu_k': Lazy(lambda: torch.tanh(o_d.chunk(2, dim=1)[0]).chunk(K, 0)),
                    'psigma_k': Lazy(lambda: (torch.sigmoid(o_d.chunk(2, dim=1)[1] + 4.0) + 1e-4).chunk(K, 0))})
        x_r_k = list(x_r.unbind(0))
        stats = LazyAttrDict(recon=recon, log_m_k=log_m_k, log_s_k=log_s_k, x_r_k=x_r_k,
                             mx_r_k=Lazy(lambda: list((x_r * log_m.exp()).unbind(0))))
        lin_d = None if lin_p is None else lin_p.detach()

        def _pmu():
            return [torch.zeros_like(mu_k[0])] + ([] if lin_d is None else list(torch.tanh(lin_d.chunk(2, dim=2)[0]).unbind(0)))

        def _psig():
            return [torch.ones_like(mu_k[0])] + ([] if lin_d is None else
                                                 list((torch.sigmoid(lin_d.chunk(2, dim=2)[1] + 4.0) + 1e-4).unbind(0)))
        att_stats = LazyAttrDict(mu_k=mu_k, sigma_k=sigma_k, z_k=z_k, pmu_k=Lazy(_pmu), psigma_k=Lazy(_psig))
        return recon, losses, stats, att_stats, comp_stats

    @torch.no_grad()
    def sample(self, batch_size, K_steps=None, eps_m=None, eps_c=None):
        """models/genesis_config.py:345-425: ancestral rollout of the mask latents through the AR prior (prior_lstm ->
        prior_linear -> N(raw mean, to_prior_sigma): the reference's sample() does NOT squash the mean here, unlike its
        mask_latent_loss), masks from the attention decoder + stick-breaking (LatentSBP.masks_from_zm_k,
        modules/attention.py:53-75; last mask = remaining scope), component latents from prior_mlp(zm) (tanh mean) or
        N(0, 1), decoded by the component decoder.  Runs on the training path's kernels (gx_linear_fwd,
        gx_lstm_step_fwd, gx_latent_prior_sample_ex, gx_sbp_scan_fwd, the BroadcastDecoder / gated-deconv stack,
        gx_mixture_w_fwd for sigmoid + mask-weighted sum).  `eps_m` [K,B,ldim] / `eps_c` [K,B,comp_ldim] inject the
        standard-normal draws in the reference's order (parity tests); default torch.randn."""
        from genesis_amd import hip_ops as hip
        K = self.K_steps
        K_arg = K if K_steps is None else K_steps
        if K_arg != K:
            # the reference rolls the masks out over self.att_steps and asserts len(zm_k) == self.K_steps
            # (genesis_config.py:351,379-380); another K_steps only ever reaches its N(0,1) component branch, whose
            # K_steps latents are then chunked into self.K_steps pieces and trip the next assertion
            raise AssertionError('Genesis.sample: K_steps must equal the model\'s K_steps (%d)' % K)
        S, B = self.img_size, batch_size
        dev = self.prior_linear.weight.device
        if eps_m is None:
            eps_m = torch.randn(K, B, self.ldim, device=dev)
        eps_m = eps_m.to(dev).contiguous()
        assert eps_m.shape == (K, B, self.ldim)
        # --- mask latents
        zm_k = [eps_m[0]]
        L = self.prior_lstm
        H = L.weight_hh_l0.shape[1]
        h_prev = c_prev = None
        for k in range(1, K):
            gx = hip.linear_fwd(zm_k[-1].contiguous(), L.weight_ih_l0, L.bias_ih_l0)
            act = torch.empty(B, 4 * H, device=dev)
            c = torch.empty(B, H, device=dev)
            h = torch.empty(B, H, device=dev)
            hip.lstm_step_fwd(gx, h_prev, c_prev, L.weight_hh_l0, L.bias_hh_l0, act, c, h)
            lin = hip.linear_fwd(h, self.prior_linear.weight, self.prior_linear.bias)
            zm_k.append(hip.latent_prior_sample(lin, eps_m[k], tanh_mu=False))
            h_prev, c_prev = h, c
        zm = torch.stack(zm_k, 0)                                                     # [K,B,ldim]
        # --- masks: decode all K latents as one batch (the reference decodes slot by slot, modules/attention.py:60-61:
        # the same numbers unless a training-mode BatchNorm takes its statistics over the batch it is given -- then
        # slot by slot here too), K stick-breaking steps in one launch
        core = self.att_process.core
        if self.training and core.dec_norm == 'bn':
            logits = torch.stack([core.decode(z_) for z_ in zm_k], 0).view(K, B, 1, S, S).contiguous()
        else:
            logits = core.decode(zm.flatten(0, 1)).view(K, B, 1, S, S).contiguous()
        log_m, log_s = hip.sbp_scan_fwd(logits, None, True)
        log_m_k = list(log_m.unbind(0))
        log_s_k = [torch.zeros(B, 1, S, S, device=dev)] + list(log_s.unbind(0))
        # --- component appearances
        if self.two_stage:
            Lc = self.comp_vae.ldim
            if eps_c is None:
                eps_c = torch.randn(K, B, Lc, device=dev)
            eps_c = eps_c.to(dev).contiguous()
            assert eps_c.shape == (K, B, Lc)
            if self.comp_prior:
                pm_ = self.prior_mlp
                o = F.elu(hip.linear_fwd(zm.flatten(0, 1).contiguous(), pm_[0].weight, pm_[0].bias))
                o = F.elu(hip.linear_fwd(o, pm_[2].weight, pm_[2].bias))
                o = hip.linear_fwd(o, pm_[4].weight, pm_[4].bias)                    # [K*B, 2*Lc]
                zc = hip.latent_prior_sample(o, eps_c.view(K * B, Lc), tanh_mu=True)
            else:
                zc = eps_c.view(K * B, Lc)
            dm, z_dec = self.comp_vae.decoder_module, zc
        else:
            dm, z_dec = self.decoder, zm.flatten(0, 1)
        if self.two_stage and self.comp_symmetric:
            hd = gc_decoder_forward(dm[1], z_dec, SYM_STRIDES, self.training)
            dec = fn.Conv1x1Fn.apply(hd, dm[2].weight, dm[2].bias)
        else:
            dec = fn.BroadcastDecoderFn.apply(z_dec, self._canvas_coords(dev), 'elu', None, *dm.flat_params())
        # sigmoid (pixel_bound) + sum_k exp(log_m_k) * x_k on the mixture kernel (its likelihood output is unused)
        x0 = torch.zeros(B, 3, S, S, device=dev)
        _, generated_image, x_r = hip.mixture_w_fwd(x0, dec.contiguous(), log_m.contiguous(), K, self._std12[0],
                                                    self._std12[1], bool(self.pixel_bound))
        x_k = list(x_r.unbind(0))
        stats = AttrDict(x_k=x_k, log_m_k=log_m_k, log_s_k=log_s_k, mx_k=[x * m.exp() for x, m in zip(x_k, log_m_k)],
                         zm_k=zm_k)
        if self.two_stage:
            stats['zc_k'] = list(zc.view(K, B, -1).unbind(0))
        return generated_image, stats

    def get_features(self, image_batch):
        """genesis_config.py:427-436."""
        with torch.no_grad():
            _, _, _, att_stats, comp_stats = self.forward(image_batch)
        if self.two_stage:
            return torch.cat([*att_stats.z_k[:self.K_steps - 1], *comp_stats.z_k], dim=1)
        return torch.cat(list(att_stats.z_k), dim=1)
