"""ORACLE -- CPU restatement of the MONet training path (BASELINE config 4).  TEST INFRASTRUCTURE: only
tests/, smoke() and bench.py's cpu_baseline leg may import this module.

Restates models/monet_config.py:74-128 (MONet.forward), modules/attention.py:31-51 (SimpleSBP),
modules/component_vae.py:45-93 (ComponentVAE), modules/encoders.py:22-40 (MONetCompEncoder),
modules/decoders.py:21-35 (BroadcastDecoder), models/monet_config.py:157-170 (kl_m_loss), utils/misc.py:238-255
(MC KL) with stock torch CPU ops, keyed by the reference's state_dict names.  Pinned against golden vectors
captured from the imported reference (tests/golden/make_golden_monet.py)."""
import math

import torch
import torch.nn.functional as F

from . import v2_oracle as V

DEFAULT_CFG = dict(filter_start=32, prior_mode='softmax', comp_enc_channels=32, comp_ldim=16, comp_dec_channels=32,
                   comp_dec_layers=4, montecarlo_kl=True, pixel_bound=True, pixel_std1=0.7, pixel_std2=0.7)


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    return cfg


def simple_sbp(p, x, K, nb):
    """SimpleSBP.forward, modules/attention.py:31-51: the UNet(IN) runs K-1 times on [x | log_s]."""
    log_s_k = [torch.zeros_like(x)[:, :1]]
    log_m_k = []
    for step in range(K - 1):
        core_out = V.unet_forward(p, torch.cat((x, log_s_k[step]), 1), nb, prefix='att_process.core', norm='in',
                                  final_conv=True)
        a = core_out[:, :1]
        log_m_k.append(log_s_k[step] + F.logsigmoid(a))
        log_s_k.append(log_s_k[step] + F.logsigmoid(-a))
    log_m_k.append(log_s_k[-1])
    return log_m_k, log_s_k


def comp_encoder(p, x, act=F.relu, prefix='comp_vae.encoder_module.module'):
    """MONetCompEncoder, modules/encoders.py:31-37: 4x conv3x3 s2 p1 + act, flatten, Linear + act, Linear."""
    h = x
    for i in (0, 2, 4, 6):
        h = act(F.conv2d(h, p['%s.%d.weight' % (prefix, i)], p['%s.%d.bias' % (prefix, i)], 2, 1))
    h = h.reshape(h.size(0), -1)
    h = act(F.linear(h, p[prefix + '.9.weight'], p[prefix + '.9.bias']))
    return F.linear(h, p[prefix + '.11.weight'], p[prefix + '.11.bias'])


def broadcast_decoder(p, z, img_size, num_layers, act=F.relu, prefix='comp_vae.decoder_module.seq'):
    """BroadcastDecoder, modules/decoders.py:25-35 (+ BroadcastLayer, modules/blocks.py:104-130): broadcast z
    to (img + 2L)^2, append row / column coordinates, L valid 3x3 convs + act, 1x1 conv."""
    N = z.size(0)
    d = img_size + 2 * num_layers
    h = z.view(N, -1, 1, 1).expand(-1, -1, d, d)
    h = torch.cat((h, V.pixel_coords(d).expand(N, -1, -1, -1).to(z.dtype)), 1)
    for l in range(num_layers):
        h = act(F.conv2d(h, p['%s.%d.weight' % (prefix, 1 + 2 * l)], p['%s.%d.bias' % (prefix, 1 + 2 * l)]))
    j = 1 + 2 * num_layers
    return F.conv2d(h, p['%s.%d.weight' % (prefix, j)], p['%s.%d.bias' % (prefix, j)])


def kl_m_loss(log_m_k, log_m_r_k):
    return V.kl_m_loss(log_m_k, log_m_r_k)


def monet_forward(p, x, cfg, eps=None):
    """MONet.forward, models/monet_config.py:74-128.  eps: [K*B, ldim] standard normal (the single rsample of
    component_vae.py:73); drawn from the default generator when None."""
    K, S = cfg['K_steps'], cfg['img_size']
    B = x.size(0)
    nb = int(math.log2(S) - 1)
    L = cfg['comp_ldim']
    log_m_k, log_s_k = simple_sbp(p, x, K, nb)
    # ComponentVAE over the K slots batched along dim 0 (slot-major), mask as FIRST channel
    inp = torch.cat((torch.cat(log_m_k, 0), x.repeat(K, 1, 1, 1)), 1)
    enc = comp_encoder(p, inp)
    mu, sigma_ps = enc.chunk(2, dim=1)
    sigma = V.to_sigma(sigma_ps)
    if eps is None:
        eps = torch.normal(torch.zeros(K * B, L), torch.ones(K * B, L))
    z = mu + sigma * eps
    dec = broadcast_decoder(p, z, S, cfg['comp_dec_layers'])          # comp_vae.pixel_bound = False (:68)
    dec_k = dec.chunk(K, 0)
    x_r_k = [d[:, :3] for d in dec_k]
    logits = [d[:, 3:] for d in dec_k]
    if cfg.get('pixel_bound', True):
        x_r_k = [torch.sigmoid(t) for t in x_r_k]
    recon = (torch.stack(log_m_k, 4).exp() * torch.stack(x_r_k, 4)).sum(4)
    if cfg.get('prior_mode', 'softmax') == 'softmax':
        log_m_r = F.log_softmax(torch.stack(logits, 4), 4)
        log_m_r_k = [log_m_r[..., k] for k in range(K)]
    else:
        # prior_mode == 'scope' (monet_config.py:141-153): stick-breaking over the K mask logits, last = remaining scope
        log_m_r_k = []
        log_s = torch.zeros_like(logits[0])
        for step, lg in enumerate(logits):
            if step == K - 1:
                log_m_r_k.append(log_s)
            else:
                log_m_r_k.append(log_s + F.logsigmoid(lg))
                log_s = log_s + F.logsigmoid(-lg)
    std = cfg.get('pixel_std2', 0.7) * torch.ones(1, 1, 1, 1, K, dtype=x.dtype)
    std[0, 0, 0, 0, 0] = cfg.get('pixel_std1', 0.7)
    losses = {'err': V.x_loss(x, log_m_k, x_r_k, std), 'kl_m': kl_m_loss(log_m_k, log_m_r_k)}
    mu_k, sigma_k, z_k = mu.chunk(K, 0), sigma.chunk(K, 0), z.chunk(K, 0)
    if cfg.get('montecarlo_kl', True):
        losses['kl_l_k'] = [(V.normal_log_prob(zz, m, s) - V.normal_log_prob(zz, 0.0, 1.0)).sum(1)
                            for zz, m, s in zip(z_k, mu_k, sigma_k)]
    else:
        # utils/misc.py:247: torch.distributions.kl_divergence(Normal(mu, sigma), Normal(0, 1)), summed over the latent
        # dimension like the Monte-Carlo estimate (monet_config.py:113)
        losses['kl_l_k'] = [(-torch.log(s) + 0.5 * (s * s + m * m) - 0.5).sum(1) for m, s in zip(mu_k, sigma_k)]
    stats = dict(recon=recon, log_m_k=log_m_k, log_s_k=log_s_k, x_r_k=x_r_k, log_m_r_k=log_m_r_k)
    comp_stats = dict(mu_k=list(mu_k), sigma_k=list(sigma_k), z_k=list(z_k))
    return recon, losses, stats, {}, comp_stats


def monet_sample(p, cfg, eps, K_steps=None):
    """MONet.sample, models/monet_config.py:172-198: z = the one standard-normal draw eps [B*K, ldim]; decode; masks by
    get_mask_recon_stack(log=False) (:135-155).  -> (gen_image, x_k, log_m_k)."""
    K = cfg['K_steps'] if K_steps is None else K_steps
    S = cfg['img_size']
    dec = broadcast_decoder(p, eps, S, cfg['comp_dec_layers'])
    x_r, logits = dec[:, :3], dec[:, 3:]
    if cfg.get('pixel_bound', True):
        x_r = torch.sigmoid(x_r)
    x_k, logit_k = list(x_r.chunk(K, 0)), list(logits.chunk(K, 0))
    if cfg.get('prior_mode', 'softmax') == 'softmax':
        m_r = F.softmax(torch.stack(logit_k, 4), 4)
    else:
        log_m, log_s = [], torch.zeros_like(logit_k[0])
        for step, lg in enumerate(logit_k):
            if step == K - 1:
                log_m.append(log_s)
            else:
                log_m.append(log_s + F.logsigmoid(lg))
                log_s = log_s + F.logsigmoid(-lg)
        m_r = torch.stack(log_m, 4).exp()
    img = (m_r * torch.stack(x_k, 4)).sum(4)
    log_m_k = [m_r[..., k].log() for k in range(K)]
    return img, x_k, log_m_k


def aggregate_losses(losses):
    """train.py:226-242."""
    err = losses['err'].mean(0)
    kl_l = torch.stack(losses['kl_l_k'], dim=1).mean(0).sum()
    kl_m = losses['kl_m'].mean(0)
    return err, kl_l, kl_m


def param_shapes(cfg):
    """Ordered name -> (shape, dtype) of MONet.state_dict() (construction order of monet_config.py:46-72)."""
    S, K = cfg['img_size'], cfg['K_steps']
    nb = int(math.log2(S) - 1)
    c = cfg['filter_start']
    if nb == 4:
        enc_in, enc_out = [4, c, 2 * c, 2 * c], [c, 2 * c, 2 * c, 2 * c]
        dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c], [2 * c, 2 * c, c, c]
    elif nb == 5:
        enc_in, enc_out = [4, c, c, 2 * c, 2 * c], [c, c, 2 * c, 2 * c, 2 * c]
        dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c, 2 * c], [2 * c, 2 * c, c, c, c]
    else:
        enc_in, enc_out = [4, c, c, c, 2 * c, 2 * c], [c, c, c, 2 * c, 2 * c, 2 * c]
        dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c, 2 * c, 2 * c], [2 * c, 2 * c, c, c, c, c]
    f32 = torch.float32
    sh = {'std': ((1, 1, 1, 1, K), f32)}   # registered buffer (monet_config.py:70-72): root-module entries come first
    pre = 'att_process.core'
    for i, (a, b) in enumerate(zip(enc_in, enc_out)):
        sh['%s.down.%d.0.weight' % (pre, i)] = ((b, a, 3, 3), f32)
        sh['%s.down.%d.1.weight' % (pre, i)] = ((b,), f32)
        sh['%s.down.%d.1.bias' % (pre, i)] = ((b,), f32)
    for i, (a, b) in enumerate(zip(dec_in, dec_out)):
        sh['%s.up.%d.0.weight' % (pre, i)] = ((b, a, 3, 3), f32)
        sh['%s.up.%d.1.weight' % (pre, i)] = ((b,), f32)
        sh['%s.up.%d.1.bias' % (pre, i)] = ((b,), f32)
    fs = S // 2 ** (nb - 1)
    flat = 2 * c * fs * fs
    for j, (o, i_) in zip((1, 3, 5), ((128, flat), (128, 128), (flat, 128))):
        sh['%s.mlp.%d.weight' % (pre, j)] = ((o, i_), f32)
        sh['%s.mlp.%d.bias' % (pre, j)] = ((o,), f32)
    sh[pre + '.final_conv.weight'] = ((1, c, 1, 1), f32)
    sh[pre + '.final_conv.bias'] = ((1,), f32)
    ce, L = cfg['comp_enc_channels'], cfg['comp_ldim']
    enc = 'comp_vae.encoder_module.module'
    for i, (a, b) in zip((0, 2, 4, 6), ((4, ce), (ce, ce), (ce, 2 * ce), (2 * ce, 2 * ce))):
        sh['%s.%d.weight' % (enc, i)] = ((b, a, 3, 3), f32)
        sh['%s.%d.bias' % (enc, i)] = ((b,), f32)
    nin = 2 * ce * (S // 16) ** 2
    nhid = max(256, 2 * L)
    sh[enc + '.9.weight'] = ((nhid, nin), f32)
    sh[enc + '.9.bias'] = ((nhid,), f32)
    sh[enc + '.11.weight'] = ((2 * L, nhid), f32)
    sh[enc + '.11.bias'] = ((2 * L,), f32)
    dec = 'comp_vae.decoder_module.seq'
    cd, nl = cfg['comp_dec_channels'], cfg['comp_dec_layers']
    for l in range(nl):
        sh['%s.%d.weight' % (dec, 1 + 2 * l)] = ((cd, L + 2 if l == 0 else cd, 3, 3), f32)
        sh['%s.%d.bias' % (dec, 1 + 2 * l)] = ((cd,), f32)
    sh['%s.%d.weight' % (dec, 1 + 2 * nl)] = ((4, cd, 1, 1), f32)
    sh['%s.%d.bias' % (dec, 1 + 2 * nl)] = ((4,), f32)
    return sh


def template_state_dict(cfg):
    sd = {}
    for name, (shape, dt) in param_shapes(cfg).items():
        sd[name] = torch.zeros(shape, dtype=dt)
    std = cfg.get('pixel_std2', 0.7) * torch.ones(1, 1, 1, 1, cfg['K_steps'])
    std[0, 0, 0, 0, 0] = cfg.get('pixel_std1', 0.7)
    sd['std'] = std
    return sd
