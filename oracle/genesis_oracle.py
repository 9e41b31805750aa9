"""ORACLE (test infrastructure) -- GENESIS (v1), BASELINE config 3: models/genesis_config.py:145-271 (forward),
modules/attention.py:77-133 (LatentSBP), modules/component_vae.py (ComponentVAE with ELU, nout=3, pixel_bound),
models/genesis_config.py:288-343 (mask_latent_loss), :229-247 (component prior MLP).  Default flags: two_stage,
autoreg_prior, comp_prior, enc_norm = dec_norm = 'bn' (training-mode batch statistics over the K*B decoder batch);
also comp_prior=False (N(0,1) component prior, :248-254) and two_stage=False (components decoded from the ATTENTION
latents by a BroadcastDecoder, :183-191; no component KL)."""
import math

import torch
import torch.nn.functional as F

from . import monet_oracle as M
from . import sylvester_oracle as S
from . import v2_oracle as V

DEFAULT_CFG = dict(two_stage=True, autoreg_prior=True, comp_prior=True, attention_latents=64, enc_norm='bn',
                   dec_norm='bn', comp_enc_channels=32, comp_ldim=16, comp_dec_channels=32, comp_dec_layers=4,
                   comp_symmetric=False, pixel_bound=True, pixel_std1=0.7, pixel_std2=0.7, montecarlo_kl=True)


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    return cfg


def _lstm_step(p, prefix, inp, state):
    H = p[prefix + '.weight_hh_l0'].size(1)
    if state is None:
        state = (torch.zeros(inp.size(0), H, dtype=inp.dtype), torch.zeros(inp.size(0), H, dtype=inp.dtype))
    h, c = state
    gates = F.linear(inp, p[prefix + '.weight_ih_l0'], p[prefix + '.bias_ih_l0']) + \
        F.linear(h, p[prefix + '.weight_hh_l0'], p[prefix + '.bias_hh_l0'])
    i, f, g, o = gates.chunk(4, 1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, (h, c)


SYM_STRIDES = [1, 2, 1, 2, 1]


def _gc_encoder(p, x, prefix, norm):
    """sylvester.build_gc_encoder (VAE.py:18-24) under `prefix`."""
    h = x
    for l, s in enumerate(SYM_STRIDES):
        name = '%s.%d' % (prefix, l)
        h = S.gated(p, name, F.conv2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], s, 2), norm)
    name = '%s.%d' % (prefix, len(SYM_STRIDES))
    h = S.gated(p, name, F.conv2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], 1, 0), None)
    return h.reshape(h.size(0), -1)


def _gc_decoder(p, z, prefix, norm, training=True):
    """sylvester.build_gc_decoder (VAE.py:27-33) under `prefix`."""
    h = z.view(z.size(0), -1, 1, 1)
    name = prefix + '.0'
    h = S.gated(p, name, F.conv_transpose2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], 1, 0), None)
    for l, s in enumerate(SYM_STRIDES):
        name = '%s.%d' % (prefix, l + 1)
        h = S.gated(p, name, F.conv_transpose2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], s, 2, s - 1), norm,
                    training)
    return h


def _gc_shapes(sh, enc_prefix, dec_prefix, nin, cfc, zdim, kfc, enc_norm, dec_norm):
    f32, i64 = torch.float32, torch.int64

    def norms(name, c, norm):
        for hn in ('h_norm', 'g_norm'):
            if norm in ('bn', 'in'):
                sh['%s.%s.weight' % (name, hn)] = ((c,), f32)
                sh['%s.%s.bias' % (name, hn)] = ((c,), f32)
            if norm == 'bn':
                sh['%s.%s.running_mean' % (name, hn)] = ((c,), f32)
                sh['%s.%s.running_var' % (name, hn)] = ((c,), f32)
                sh['%s.%s.num_batches_tracked' % (name, hn)] = ((), i64)

    for l, (a, b) in enumerate(zip([nin, 32, 32, 64, 64], [32, 32, 64, 64, 64])):
        name = '%s.%d' % (enc_prefix, l)
        sh[name + '.conv.weight'] = ((2 * b, a, 5, 5), f32)
        sh[name + '.conv.bias'] = ((2 * b,), f32)
        norms(name, b, enc_norm)
    name = '%s.5' % enc_prefix
    sh[name + '.conv.weight'] = ((2 * cfc, 64, kfc, kfc), f32)
    sh[name + '.conv.bias'] = ((2 * cfc,), f32)
    name = dec_prefix + '.0'
    sh[name + '.conv.weight'] = ((zdim, 128, kfc, kfc), f32)
    sh[name + '.conv.bias'] = ((128,), f32)
    for l, (a, b) in enumerate(zip([64, 64, 32, 32, 32], [64, 32, 32, 32, 32])):
        name = '%s.%d' % (dec_prefix, l + 1)
        sh[name + '.conv.weight'] = ((a, 2 * b, 5, 5), f32)
        sh[name + '.conv.bias'] = ((2 * b,), f32)
        norms(name, b, dec_norm)


def latent_sbp(p, x, K, cfg, eps_m):
    """LatentSBP.forward (modules/attention.py:84-133) followed by the K+1 -> K mask correction of
    genesis_config.py:167-169.  eps_m: list of K [B, ldim] noises (one rsample per step)."""
    S_, L = cfg['img_size'], cfg['attention_latents']
    pre = 'att_process.core'
    h = S.encode(p, x, S_, pre, cfg['enc_norm'])
    mean, var = S.posterior(p, h, pre)
    mu_k, sigma_k, z_k = [mean], [var.sqrt()], [mean + var.sqrt() * eps_m[0]]
    state = None
    for step in range(1, K):
        out, state = _lstm_step(p, 'att_process.lstm', torch.cat([h, z_k[-1]], 1), state)
        lin = F.linear(out, p['att_process.linear.weight'], p['att_process.linear.bias'])
        mean_k, var_raw = lin.chunk(2, dim=1)
        sig = V.to_sigma(var_raw)                       # sqrt(to_var) == to_sigma
        mu_k.append(mean_k); sigma_k.append(sig); z_k.append(mean_k + sig * eps_m[step])
    out = S.decode(p, torch.cat(z_k, 0), S_, pre, cfg['dec_norm']).chunk(K, 0)
    log_s_k = [torch.zeros_like(x)[:, :1]]
    log_m_k = []
    for step in range(K):
        a = out[step][:, :1]
        log_m_k.append(log_s_k[step] + F.logsigmoid(a))
        log_s_k.append(log_s_k[step] + F.logsigmoid(-a))
    log_m_k[K - 1] = log_s_k[K - 1]                      # genesis_config.py:167-169
    return log_m_k, log_s_k, mu_k, sigma_k, z_k


def genesis_forward(p, x, cfg, eps_m=None, eps_c=None):
    """Genesis.forward, models/genesis_config.py:145-271 (two_stage, K > 1).  RNG order when noise is not
    injected: K normals [B, ldim] (LatentSBP), then one [K*B, comp_ldim] (ComponentVAE)."""
    K, S_, B = cfg['K_steps'], cfg['img_size'], x.size(0)
    L, Lc = cfg['attention_latents'], cfg['comp_ldim']
    if eps_m is None:
        eps_m = [torch.normal(torch.zeros(B, L), torch.ones(B, L)) for _ in range(K)]
    log_m_k, log_s_k, mu_k, sigma_k, z_k = latent_sbp(p, x, K, cfg, eps_m)
    std = cfg.get('pixel_std2', 0.7) * torch.ones(1, 1, 1, 1, K, dtype=x.dtype)
    std[0, 0, 0, 0, 0] = cfg.get('pixel_std1', 0.7)
    att_stats = dict(mu_k=mu_k, sigma_k=sigma_k, z_k=z_k)
    if not cfg.get('two_stage', True):
        # one stage (genesis_config.py:183-191): x_r_k from the attention latents through `decoder`
        dec = M.broadcast_decoder(p, torch.cat(z_k, 0), S_, cfg['comp_dec_layers'], act=F.elu, prefix='decoder.seq')
        if cfg.get('pixel_bound', True):
            dec = torch.sigmoid(dec)
        x_r_k = list(dec.chunk(K, 0))
        recon = (torch.stack(log_m_k, 4).exp() * torch.stack(x_r_k, 4)).sum(4)
        losses = {'err': V.x_loss(x, log_m_k, x_r_k, std)}
        losses['kl_m_k'] = V.mask_latent_loss(p, mu_k, sigma_k, z_k, cfg.get('autoreg_prior', True))
        stats = dict(recon=recon, log_m_k=log_m_k, log_s_k=log_s_k, x_r_k=x_r_k)
        return recon, losses, stats, att_stats, None
    # ComponentVAE (ELU)
    inp = torch.cat((torch.cat(log_m_k, 0), x.repeat(K, 1, 1, 1)), 1)
    sym = cfg.get('comp_symmetric', False)
    if sym:
        # comp_symmetric (genesis_config.py:104-123): gated-conv encoder / decoder as in the attention VAE, strides
        # [1, 2, 1, 2, 1], 'fc' kernel = the attention core's last_kernel_size
        enc = _gc_encoder(p, inp, 'comp_vae.encoder_module.0', cfg['enc_norm'])
    else:
        enc = M.comp_encoder(p, inp, act=F.elu)
    mu_c, sig_ps = enc.chunk(2, dim=1)
    sig_c = V.to_sigma(sig_ps)
    if eps_c is None:
        eps_c = torch.normal(torch.zeros(K * B, Lc), torch.ones(K * B, Lc))
    z_c = mu_c + sig_c * eps_c
    if sym:
        h = _gc_decoder(p, z_c, 'comp_vae.decoder_module.1', cfg['dec_norm'])
        dec = F.conv2d(h, p['comp_vae.decoder_module.2.weight'], p['comp_vae.decoder_module.2.bias'])
    else:
        dec = M.broadcast_decoder(p, z_c, S_, cfg['comp_dec_layers'], act=F.elu)
    if cfg.get('pixel_bound', True):
        dec = torch.sigmoid(dec)                         # comp_vae.pixel_bound (component_vae.py:89-93)
    x_r_k = list(dec.chunk(K, 0))
    recon = (torch.stack(log_m_k, 4).exp() * torch.stack(x_r_k, 4)).sum(4)
    losses = {'err': V.x_loss(x, log_m_k, x_r_k, std)}
    losses['kl_m_k'] = V.mask_latent_loss(p, mu_k, sigma_k, z_k, cfg.get('autoreg_prior', True))
    mu_ck, sig_ck, z_ck = mu_c.chunk(K, 0), sig_c.chunk(K, 0), z_c.chunk(K, 0)
    kl_l_k = []
    for k in range(K):
        if cfg.get('comp_prior', True):
            h = F.elu(F.linear(z_k[k], p['prior_mlp.0.weight'], p['prior_mlp.0.bias']))
            h = F.elu(F.linear(h, p['prior_mlp.2.weight'], p['prior_mlp.2.bias']))
            o = F.linear(h, p['prior_mlp.4.weight'], p['prior_mlp.4.bias'])
            pm, ps = o.chunk(2, dim=1)
            pm, ps = torch.tanh(pm), V.to_prior_sigma(ps)
        else:
            pm, ps = 0.0, 1.0
        kl_l_k.append((V.normal_log_prob(z_ck[k], mu_ck[k], sig_ck[k]) - V.normal_log_prob(z_ck[k], pm, ps)).sum(1))
    losses['kl_l_k'] = kl_l_k
    stats = dict(recon=recon, log_m_k=log_m_k, log_s_k=log_s_k, x_r_k=x_r_k)
    comp_stats = dict(mu_k=list(mu_ck), sigma_k=list(sig_ck), z_k=list(z_ck))
    return recon, losses, stats, att_stats, comp_stats


def genesis_sample(p, cfg, eps_m, eps_c=None, training=False):
    """Genesis.sample, models/genesis_config.py:345-425, on injected standard-normal draws: eps_m K x [B, ldim] (mask
    rollout; every torch Normal.sample is mean + std * eps), eps_c K x [B, comp_ldim] (two-stage model).  The mask
    rollout's mean is the RAW first half of prior_linear's output (:358 -- no tanh, unlike mask_latent_loss :309);
    masks by LatentSBP.masks_from_zm_k (modules/attention.py:53-75) + the K+1 -> K fix-up (:374-376).
    training=False: eval-mode BatchNorm, as every caller of the reference's sample() runs it.
    -> (generated_image, x_k, log_m_k, log_s_k, zm_k, zc_k or None)."""
    K, S_ = cfg['K_steps'], cfg['img_size']
    L = cfg['attention_latents']
    zm_k = [eps_m[0]]
    state = None
    for k in range(1, K):
        out, state = _lstm_step(p, 'prior_lstm', zm_k[-1], state)
        lin = F.linear(out, p['prior_linear.weight'], p['prior_linear.bias'])
        mu, sigma = lin[:, :L], V.to_prior_sigma(lin[:, L:])
        zm_k.append(mu + sigma * eps_m[k])
    B = zm_k[0].size(0)
    log_m_k, log_s_k = [], [torch.zeros(B, 1, S_, S_, dtype=zm_k[0].dtype)]
    for zm in zm_k:                                       # masks_from_zm_k decodes slot by slot (:60-61)
        a = S.decode(p, zm, S_, 'att_process.core', cfg['dec_norm'], training)[:, :1]
        log_m_k.append(log_s_k[-1] + F.logsigmoid(a))
        log_s_k.append(log_s_k[-1] + F.logsigmoid(-a))
    log_m_k[K - 1] = log_s_k[K - 1]                       # :374-376 (the appended (K+1)-th mask is dropped)
    zc_k = None
    if cfg.get('two_stage', True):
        zc_k = []
        for k, zm in enumerate(zm_k):
            if cfg.get('comp_prior', True):
                h = F.elu(F.linear(zm, p['prior_mlp.0.weight'], p['prior_mlp.0.bias']))
                h = F.elu(F.linear(h, p['prior_mlp.2.weight'], p['prior_mlp.2.bias']))
                o = F.linear(h, p['prior_mlp.4.weight'], p['prior_mlp.4.bias'])
                pm, ps = o.chunk(2, dim=1)
                zc_k.append(torch.tanh(pm) + V.to_prior_sigma(ps) * eps_c[k])
            else:
                zc_k.append(eps_c[k])
        zc = torch.cat(zc_k, 0)
        if cfg.get('comp_symmetric', False):
            h = _gc_decoder(p, zc, 'comp_vae.decoder_module.1', cfg['dec_norm'], training)
            dec = F.conv2d(h, p['comp_vae.decoder_module.2.weight'], p['comp_vae.decoder_module.2.bias'])
        else:
            dec = M.broadcast_decoder(p, zc, S_, cfg['comp_dec_layers'], act=F.elu)
    else:
        dec = M.broadcast_decoder(p, torch.cat(zm_k, 0), S_, cfg['comp_dec_layers'], act=F.elu, prefix='decoder.seq')
    if cfg.get('pixel_bound', True):
        dec = torch.sigmoid(dec)
    x_k = list(dec.chunk(K, 0))
    img = (torch.stack(log_m_k, 4).exp() * torch.stack(x_k, 4)).sum(4)
    return img, x_k, log_m_k, log_s_k, zm_k, zc_k


def aggregate_losses(losses):
    """train.py:226-242: kl_m from the kl_m_k list, kl_l from the kl_l_k list."""
    err = losses['err'].mean(0)
    kl_m = torch.stack(losses['kl_m_k'], dim=1).mean(0).sum()
    kl_l = torch.stack(losses['kl_l_k'], dim=1).mean(0).sum() if 'kl_l_k' in losses else torch.zeros(())
    return err, kl_l, kl_m


def param_shapes(cfg):
    """Ordered state_dict layout of Genesis (construction order genesis_config.py:92-143): root buffer `std`, then
    att_process.core (sylvester VAE) / lstm / linear, comp_vae, prior_lstm, prior_linear, prior_mlp."""
    f32 = torch.float32
    K, S_, L, Lc = cfg['K_steps'], cfg['img_size'], cfg['attention_latents'], cfg['comp_ldim']
    sh = {'std': ((1, 1, 1, 1, K), f32)}
    sh.update(S.param_shapes('att_process.core', L, 3, S_, 1, cfg['enc_norm'], cfg['dec_norm']))
    H = 2 * L
    sh['att_process.lstm.weight_ih_l0'] = ((4 * H, L + 256), f32)
    sh['att_process.lstm.weight_hh_l0'] = ((4 * H, H), f32)
    sh['att_process.lstm.bias_ih_l0'] = ((4 * H,), f32)
    sh['att_process.lstm.bias_hh_l0'] = ((4 * H,), f32)
    sh['att_process.linear.weight'] = ((2 * L, H), f32)
    sh['att_process.linear.bias'] = ((2 * L,), f32)
    mon = M.param_shapes(dict(cfg, filter_start=32))
    two_stage = cfg.get('two_stage', True)
    if not two_stage:
        # `decoder` = BroadcastDecoder(ldim -> 3), registered where comp_vae would be (genesis_config.py:124-129)
        c, nl = cfg['comp_dec_channels'], cfg['comp_dec_layers']
        for l in range(nl):
            sh['decoder.seq.%d.weight' % (1 + 2 * l)] = ((c, (L + 2) if l == 0 else c, 3, 3), f32)
            sh['decoder.seq.%d.bias' % (1 + 2 * l)] = ((c,), f32)
        sh['decoder.seq.%d.weight' % (1 + 2 * nl)] = ((3, c, 1, 1), f32)
        sh['decoder.seq.%d.bias' % (1 + 2 * nl)] = ((3,), f32)
    if two_stage and cfg.get('comp_symmetric', False):
        kfc, _ = S.vae_geometry(S_)
        _gc_shapes(sh, 'comp_vae.encoder_module.0', 'comp_vae.decoder_module.1', 4, 2 * Lc, Lc, kfc, cfg['enc_norm'],
                   cfg['dec_norm'])
        sh['comp_vae.decoder_module.2.weight'] = ((3, 32, 1, 1), f32)
        sh['comp_vae.decoder_module.2.bias'] = ((3,), f32)
        mon = {}
    for k, v in mon.items():
        if two_stage and k.startswith('comp_vae.'):
            if k.endswith('decoder_module.seq.%d.weight' % (1 + 2 * cfg['comp_dec_layers'])):
                v = ((3,) + v[0][1:], v[1])
            if k.endswith('decoder_module.seq.%d.bias' % (1 + 2 * cfg['comp_dec_layers'])):
                v = ((3,), v[1])
            sh[k] = v
    sh['prior_lstm.weight_ih_l0'] = ((1024, L), f32)
    sh['prior_lstm.weight_hh_l0'] = ((1024, 256), f32)
    sh['prior_lstm.bias_ih_l0'] = ((1024,), f32)
    sh['prior_lstm.bias_hh_l0'] = ((1024,), f32)
    sh['prior_linear.weight'] = ((2 * L, 256), f32)
    sh['prior_linear.bias'] = ((2 * L,), f32)
    if two_stage and cfg.get('comp_prior', True):
        for j, (o, i_) in zip((0, 2, 4), ((256, L), (256, 256), (2 * Lc, 256))):
            sh['prior_mlp.%d.weight' % j] = ((o, i_), f32)
            sh['prior_mlp.%d.bias' % j] = ((o,), f32)
    return sh


def template_state_dict(cfg):
    sd = {}
    for k, (s, dt) in param_shapes(cfg).items():
        sd[k] = torch.ones(s, dtype=dt) if k.endswith('running_var') else torch.zeros(s, dtype=dt)
    std = cfg.get('pixel_std2', 0.7) * torch.ones(1, 1, 1, 1, cfg['K_steps'])
    std[0, 0, 0, 0, 0] = cfg.get('pixel_std1', 0.7)
    sd['std'] = std
    return sd
