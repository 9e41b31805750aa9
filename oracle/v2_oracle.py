"""ORACLE -- CPU restatement of the GENESIS-V2 training hot path.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (genesis_amd/) never does.

It restates, with stock torch CPU fp32 ops, what the reference computes on the path
`GenesisV2.forward` -> loss aggregation -> GECO (reference file:line cited per
function).  It is written in *reference-equivalent form* (per-slot Python loops, the
K-fold recomputation of feat_head, per-image seed gather) so that timing it on a
host's CPU cores stands for "the reference CPU path" (BASELINE.md §4).

Pinning: tests/test_oracle_vs_golden.py checks this restatement against golden vectors
captured from the real reference imported in the build container
(tests/golden/make_golden.py); when /root/reference is present the same test file also
compares live against the import.  Third-party arithmetic = PyTorch ATen (reference
pins pytorch 1.3.1, environment.yml:67; here torch 2.10) -- unpinned beyond those
captured vectors.

The functional API takes a `state_dict`-shaped dict of tensors with the reference's
key names, so identical weights can be loaded into the
reference, this oracle and the HIP product.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- helpers
def conv_gn_relu(p, prefix, x, groups=8, norm='gn'):
    """ConvGNReLU: conv3x3 s1 p1 no-bias -> GroupNorm(8, eps=1e-5, affine) -> ReLU, modules/blocks.py:159-165;
    norm='in': ConvINReLU with InstanceNorm2d(affine=True), modules/blocks.py:151-157."""
    y = F.conv2d(x, p[prefix + '.0.weight'], None, 1, 1)
    if norm == 'in':
        y = F.instance_norm(y, weight=p[prefix + '.1.weight'], bias=p[prefix + '.1.bias'], eps=1e-5)
    else:
        y = F.group_norm(y, groups, p[prefix + '.1.weight'], p[prefix + '.1.bias'], 1e-5)
    return F.relu(y)


def pixel_coords(size, dtype=torch.float32):
    """Row coordinate grid then column grid, linspace(-1,1). modules/blocks.py:42-47
    (torch.meshgrid default 'ij' indexing: first output varies along rows)."""
    lin = torch.linspace(-1, 1, size, dtype=dtype)
    g_row = lin.view(size, 1).expand(size, size)
    g_col = lin.view(1, size).expand(size, size)
    return torch.stack((g_row, g_col), 0).unsqueeze(0).contiguous()  # [1,2,S,S]


def unet_forward(p, x, num_blocks, prefix='encoder', norm='gn', final_conv=False):
    """UNet.forward, modules/unet.py:69-90.  GENESIS-V2 replaces final_conv by Identity
    (models/genesisv2_config.py:70); MONet keeps the 1x1 final_conv and uses norm='in'."""
    B = x.size(0)
    skip = []
    act = x
    for i in range(num_blocks):
        act = conv_gn_relu(p, '%s.down.%d' % (prefix, i), act, norm=norm)
        skip.append(act)
        if i < num_blocks - 1:
            act = F.interpolate(act, scale_factor=0.5, mode='nearest')
    fs = act.size(-1)
    h = act.reshape(B, -1)
    for j in (1, 3, 5):
        h = F.relu(F.linear(h, p['%s.mlp.%d.weight' % (prefix, j)],
                            p['%s.mlp.%d.bias' % (prefix, j)]))
    x_up = h.view(B, -1, fs, fs)
    for i in range(num_blocks):
        feat = torch.cat([x_up, skip[-1 - i]], dim=1)
        x_up = conv_gn_relu(p, '%s.up.%d' % (prefix, i), feat, norm=norm)
        if i < num_blocks - 1:
            x_up = F.interpolate(x_up, scale_factor=2.0, mode='nearest')
    if final_conv:
        x_up = F.conv2d(x_up, p[prefix + '.final_conv.weight'], p[prefix + '.final_conv.bias'])
    return x_up


def semiconv(p, feat, img_size, semiconv_on=True, prefix='att_process.colour_head'):
    """SemiConv: gate * (conv1x1 + b) + uv, delta = gated last two channels.
    modules/blocks.py:167-178, ScalarGate :85-90."""
    if not semiconv_on:
        return F.conv2d(feat, p[prefix + '.weight'], p[prefix + '.bias']), None
    out = p[prefix + '.gate.gate'] * F.conv2d(
        feat, p[prefix + '.conv.weight'], p[prefix + '.conv.bias'])
    delta = out[:, -2:]
    nout = out.size(1)
    uv = torch.cat((torch.zeros(1, nout - 2, img_size, img_size),
                    pixel_coords(img_size)), dim=1).to(out.dtype)
    return out + uv, delta


def clamp_preserve_gradients(x, lo, hi):
    """Straight-through clamp, modules/blocks.py:18-20."""
    return x + (x.clamp(lo, hi) - x).detach()


def ic_sbp(colour, log_sigma, steps, rand_pixel, kernel='gaussian', seed_idx=None, dynamic_K=False):
    """InstanceColouringSBP.forward, modules/attention.py:162-226.

    colour [B,C,H,W]; log_sigma 0-dim (float64 in the reference's checkpoints);
    rand_pixel [B,1,H,W] drawn once (:177-178).  `seed_idx` (list of [B] long) lets a
    test inject the seed pixels.  Returns (log_m_k, log_s_k, seeds, seed_indices)."""
    B, C, H, W = colour.shape
    log_s_k = [torch.zeros(B, 1, H, W)]
    log_m_k, seeds, idxs = [], [], []
    flat = colour.flatten(2)
    for step in range(steps):
        scope = log_s_k[step].exp()  # bilinear interpolate to same size is identity
        if seed_idx is None:
            rand_max = (rand_pixel * scope).flatten(2).argmax(2).flatten()
        else:
            rand_max = seed_idx[step]
        idxs.append(rand_max)
        # reference: per-image python loop with CopySlices (:190-193); gradient flows
        # through the gathered seed back into `colour`.
        seed = torch.stack([flat[b, :, rand_max[b]] for b in range(B)], 0)
        seeds.append(seed)
        diff = colour - seed.view(B, C, 1, 1)
        if kernel == 'laplacian':
            dist = clamp_preserve_gradients((diff ** 2).sum(1), 1e-10, 1e10).sqrt()
            alpha = torch.exp(-dist / log_sigma.exp())
        elif kernel == 'gaussian':
            dist = (diff ** 2).sum(1)
            alpha = torch.exp(-dist / log_sigma.exp())
        elif kernel == 'epanechnikov':
            dist = (diff ** 2).sum(1)
            alpha = (1 - dist / log_sigma.exp()).relu()
        else:
            raise ValueError('No valid kernel.')
        alpha = alpha.unsqueeze(1)
        alpha = clamp_preserve_gradients(alpha, 0.01, 0.99)
        log_a = torch.log(alpha)
        log_neg_a = torch.log(1 - alpha)
        log_m = log_s_k[step] + log_a
        if dynamic_K and log_m.exp().sum() < 20:      # modules/attention.py:218-219 (one image per call, :168-169)
            assert B == 1
            break
        log_m_k.append(log_m)
        log_s_k.append(log_s_k[step] + log_neg_a)
    log_m_k.append(log_s_k[-1])
    return log_m_k, log_s_k, seeds, idxs


def to_sigma(x):
    """modules/blocks.py:22-23."""
    return F.softplus(x + 0.5) + 1e-8


def to_prior_sigma(x, bias=4.0, eps=1e-4):
    """modules/blocks.py:28-34."""
    return torch.sigmoid(x + bias) + eps


def normal_log_prob(x, mu, sigma):
    """torch.distributions.Normal.log_prob restated."""
    if not torch.is_tensor(sigma):
        log_sigma = math.log(sigma)
    else:
        log_sigma = sigma.log()
    var = sigma ** 2
    return -((x - mu) ** 2) / (2 * var) - log_sigma - math.log(math.sqrt(2 * math.pi))


def decoder(p, z, img_size, prefix='decoder_module'):
    """BroadcastLayer(img/16) + 4x[ConvT k5 s2 p2 op1 + GN(8) + ReLU] + conv1x1.
    models/genesisv2_config.py:89-99, modules/blocks.py:104-130."""
    B = z.size(0)
    d = img_size // 16
    h = z.view(B, -1, 1, 1).expand(-1, -1, d, d)
    coords = pixel_coords(d).expand(B, -1, -1, -1).to(z.dtype)
    h = torch.cat((h, coords), dim=1)
    for conv_i, gn_i in ((1, 2), (4, 5), (7, 8), (10, 11)):
        h = F.conv_transpose2d(h, p['%s.%d.weight' % (prefix, conv_i)],
                               p['%s.%d.bias' % (prefix, conv_i)], 2, 2, 1)
        h = F.relu(F.group_norm(h, 8, p['%s.%d.weight' % (prefix, gn_i)],
                                p['%s.%d.bias' % (prefix, gn_i)], 1e-5))
    return F.conv2d(h, p[prefix + '.13.weight'], p[prefix + '.13.bias'])


def decode_latents(p, z_k, img_size, pixel_bound=True, batched=False):
    """GenesisV2.decode_latents, models/genesisv2_config.py:205-225;
    MONet.get_mask_recon_stack('softmax', log=True), models/monet_config.py:135-139."""
    if batched:
        K = len(z_k)
        dec = decoder(p, torch.cat(z_k, 0), img_size).chunk(K, 0)
    else:
        dec = [decoder(p, z, img_size) for z in z_k]
    x_r_k = [d[:, :3] for d in dec]
    logits = [d[:, 3:] for d in dec]
    if pixel_bound:
        x_r_k = [torch.sigmoid(t) for t in x_r_k]
    log_m_r = F.log_softmax(torch.stack(logits, dim=4), dim=4)
    log_m_r_k = [log_m_r[..., k] for k in range(len(z_k))]
    recon = (torch.stack(log_m_r_k, 4).exp() * torch.stack(x_r_k, 4)).sum(4)
    return recon, x_r_k, log_m_r_k


def x_loss(x, log_m_k, x_r_k, std):
    """Genesis.x_loss, models/genesis_config.py:273-286 (no log-sum-exp trick)."""
    log_xr = normal_log_prob(x.unsqueeze(4), torch.stack(x_r_k, 4), std)
    log_mx = torch.stack(log_m_k, 4) + log_xr
    err_ppc = -torch.log(log_mx.exp().sum(4))
    return err_ppc.sum(dim=(1, 2, 3))


def mask_latent_loss(p, mu_k, sigma_k, z_k, autoreg_prior=True):
    """Genesis.mask_latent_loss, models/genesis_config.py:288-343, as called from
    models/genesisv2_config.py:178-181 (zm_k_k=None, ldj_k=None)."""
    K = len(z_k)
    B, D = z_k[0].shape
    prior = [(0.0, 1.0)]
    if autoreg_prior and K > 1:
        seq = torch.stack(z_k[:-1], 0)  # [K-1,B,D]
        H = p['prior_lstm.weight_hh_l0'].size(1)
        h = torch.zeros(B, H, dtype=seq.dtype)
        c = torch.zeros(B, H, dtype=seq.dtype)
        outs = []
        for t in range(K - 1):
            gates = F.linear(seq[t], p['prior_lstm.weight_ih_l0'], p['prior_lstm.bias_ih_l0']) \
                + F.linear(h, p['prior_lstm.weight_hh_l0'], p['prior_lstm.bias_hh_l0'])
            i, f, g, o = gates.chunk(4, 1)  # torch LSTM gate order: i, f, g, o
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        lin = F.linear(torch.stack(outs, 0), p['prior_linear.weight'], p['prior_linear.bias'])
        mu_raw, sig_raw = lin.chunk(2, dim=2)
        mu_p = torch.tanh(mu_raw)
        sig_p = to_prior_sigma(sig_raw)
        prior += [(mu_p[t], sig_p[t]) for t in range(K - 1)]
    else:
        prior = K * [(0.0, 1.0)]
    kl_k = []
    for k in range(K):
        log_q = normal_log_prob(z_k[k], mu_k[k], sigma_k[k]).sum(1)
        log_p = normal_log_prob(z_k[k], prior[k][0], prior[k][1]).sum(1)
        kl_k.append(log_q - log_p)
    return kl_k


def v2_sample(p, cfg, eps_k):
    """GenesisV2.sample, models/genesisv2_config.py:227-256, with the standard-normal draws injected: eps_k is the
    list of K [B,D] draws the reference makes in order (Normal(0,1).sample for the first slot, one p_z.sample() =
    mu + sigma * eps per later slot).  Returns (recon, x_r_k, log_m_r_k, z_k)."""
    K = len(eps_k)
    B, D = eps_k[0].shape
    if cfg['autoreg_prior']:
        z_k = [eps_k[0]]
        H = p['prior_lstm.weight_hh_l0'].size(1)
        h = torch.zeros(B, H, dtype=eps_k[0].dtype)
        c = torch.zeros(B, H, dtype=eps_k[0].dtype)
        for k in range(1, K):
            gates = F.linear(z_k[-1], p['prior_lstm.weight_ih_l0'], p['prior_lstm.bias_ih_l0']) \
                + F.linear(h, p['prior_lstm.weight_hh_l0'], p['prior_lstm.bias_hh_l0'])
            i, f, g, o = gates.chunk(4, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            lin = F.linear(h, p['prior_linear.weight'], p['prior_linear.bias'])
            mu_raw, sig_raw = lin.chunk(2, dim=1)
            z_k.append(torch.tanh(mu_raw) + to_prior_sigma(sig_raw) * eps_k[k])
    else:
        z_k = list(eps_k)
    recon, x_r_k, log_m_r_k = decode_latents(p, z_k, cfg['img_size'], cfg.get('pixel_bound', True))
    return recon, x_r_k, log_m_r_k, z_k


def kl_m_loss(log_m_k, log_m_r_k):
    """MONet.kl_m_loss, models/monet_config.py:157-170 (Categorical KL, probs floored at
    1e-5 then renormalised by Categorical)."""
    B = log_m_k[0].size(0)
    K = len(log_m_k)
    q = torch.stack(log_m_k, 4).exp().clamp_min(1e-5).view(-1, K)
    pr = torch.stack(log_m_r_k, 4).exp().clamp_min(1e-5).view(-1, K)
    q = q / q.sum(-1, keepdim=True)
    pr = pr / pr.sum(-1, keepdim=True)
    kl = (q * (q.log() - pr.log())).sum(-1)
    return kl.view(B, -1).sum(1)


# --------------------------------------------------------------------------- forward
def v2_forward(p, x, cfg, rand_pixel=None, eps_k=None, seed_idx=None,
               reference_form=True):
    """GenesisV2.forward, models/genesisv2_config.py:110-203.

    cfg: mapping with K_steps, img_size, feat_dim, kernel, semiconv, pixel_bound,
    autoreg_prior, pixel_std1, klm_loss, detach_mr_in_klm.
    RNG: if `rand_pixel`/`eps_k` are None they are drawn from torch's default CPU
    generator in the reference's order (one uniform [B,1,H,W], attention.py:177-178,
    then K standard normals [B,D], genesisv2_config.py:157).
    reference_form=True keeps the reference's K-fold feat_head recomputation and per-slot
    decoder calls (for CPU-baseline timing); False computes them once / batched
    (same values up to fp32 summation order).
    """
    K = cfg['K_steps']
    S = cfg['img_size']
    D = cfg['feat_dim']
    B = x.size(0)
    nb = int(math.log2(S) - 1)
    enc = F.relu(unet_forward(p, x, nb))
    seg = conv_gn_relu(p, 'seg_head', enc)
    colour, delta = semiconv(p, seg, S, cfg.get('semiconv', True))
    if rand_pixel is None:
        rand_pixel = torch.empty(B, 1, S, S).uniform_()
    dyn_batched = False
    if cfg.get('dynamic_K', False) and B > 1:
        # models/genesisv2_config.py:119-132: one image at a time, finished images padded with -1e10 masks
        dyn_batched = True
        log_m_k = [[] for _ in range(K)]
        for b in range(B):
            lm_b, _, _, _ = ic_sbp(colour[b:b + 1], p['att_process.log_sigma'], K - 1, rand_pixel[b:b + 1],
                                   cfg.get('kernel', 'gaussian'), None, True)
            for step in range(K):
                log_m_k[step].append(lm_b[step] if step < len(lm_b) else -1e10 * torch.ones(1, 1, S, S))
        log_m_k = [torch.cat(l, 0) for l in log_m_k]
        log_s_k, seeds, idxs = None, None, None
    else:
        log_m_k, log_s_k, seeds, idxs = ic_sbp(
            colour, p['att_process.log_sigma'], K - 1, rand_pixel,
            cfg.get('kernel', 'gaussian'), seed_idx, cfg.get('dynamic_K', False))

    def feat_head():
        f = conv_gn_relu(p, 'feat_head.0', enc)
        return F.conv2d(f, p['feat_head.1.weight'], p['feat_head.1.bias'])

    feat = None if reference_form else feat_head()
    mu_k, sigma_k, z_k = [], [], []
    for k, log_m in enumerate(log_m_k):
        mask = log_m.exp()
        f = feat_head() if reference_form else feat
        obj = (mask * f).sum((2, 3)) / (mask.sum((2, 3)) + 1e-5)
        h = F.layer_norm(obj, (2 * D,), p['z_head.0.weight'], p['z_head.0.bias'], 1e-5)
        h = F.relu(F.linear(h, p['z_head.1.weight'], p['z_head.1.bias']))
        h = F.linear(h, p['z_head.3.weight'], p['z_head.3.bias'])
        mu, sigma_ps = h.chunk(2, dim=1)
        sigma = to_sigma(sigma_ps)
        eps = torch.normal(torch.zeros(B, D), torch.ones(B, D)) if eps_k is None else eps_k[k]
        mu_k.append(mu)
        sigma_k.append(sigma)
        z_k.append(mu + sigma * eps)

    recon, x_r_k, log_m_r_k = decode_latents(
        p, z_k, S, cfg.get('pixel_bound', True), batched=not reference_form)
    losses = {'err': x_loss(x, log_m_r_k, x_r_k, cfg.get('pixel_std1', 0.7))}
    if cfg.get('klm_loss', False):
        lmr = [m.detach() for m in log_m_r_k] if cfg.get('detach_mr_in_klm', True) else log_m_r_k
        losses['kl_m'] = kl_m_loss(log_m_k, lmr)
    losses['kl_l_k'] = mask_latent_loss(p, mu_k, sigma_k, z_k, cfg.get('autoreg_prior', True))
    stats = dict(recon=recon, log_m_k=log_m_k, log_s_k=log_s_k, x_r_k=x_r_k,
                 log_m_r_k=log_m_r_k,
                 instance_seg=torch.argmax(torch.cat(log_m_k, 1), 1),
                 instance_seg_r=torch.argmax(torch.cat(log_m_r_k, 1), 1))
    att_stats = None if dyn_batched else dict(colour=colour, delta=delta, seeds=seeds, seed_idx=idxs)
    comp_stats = dict(mu_k=mu_k, sigma_k=sigma_k, z_k=z_k)
    return recon, losses, stats, att_stats, comp_stats


def aggregate_losses(losses):
    """train.py:226-242: err = mean_b, kl = sum_k mean_b."""
    err = losses['err'].mean(0)
    kl_l = torch.stack(losses['kl_l_k'], dim=1).mean(0).sum()
    kl_m = losses['kl_m'].mean(0) if 'kl_m' in losses else torch.zeros(())
    return err, kl_l, kl_m


class GECO(object):
    """utils/geco.py:17-51, restated."""

    def __init__(self, goal, step_size, alpha=0.99, beta_init=1.0, beta_min=1e-10,
                 speedup=None):
        self.err_ema = None
        self.goal = goal
        self.step_size = step_size
        self.alpha = alpha
        self.beta = torch.tensor(beta_init)
        self.beta_min = torch.tensor(beta_min)
        self.beta_max = torch.tensor(1e10)
        self.speedup = speedup

    def loss(self, err, kld):
        loss = err + self.beta * kld
        with torch.no_grad():
            if self.err_ema is None:
                self.err_ema = err.detach()
            else:
                self.err_ema = (1.0 - self.alpha) * err.detach() + self.alpha * self.err_ema
            constraint = self.goal - self.err_ema
            if self.speedup is not None and constraint.item() > 0:
                factor = torch.exp(self.speedup * self.step_size * constraint)
            else:
                factor = torch.exp(self.step_size * constraint)
            self.beta = (factor * self.beta).clamp(self.beta_min, self.beta_max)
        return loss


def make_geco(img_size, g_goal=0.5655, g_lr=1e-5, g_alpha=0.99, g_init=1.0,
              g_min=1e-10, g_speedup=10):
    """train.py:159-167."""
    return GECO(g_goal * 3 * img_size ** 2, g_lr * (64 ** 2 / img_size ** 2),
                g_alpha, g_init, g_min, g_speedup)


DEFAULT_CFG = dict(feat_dim=64, kernel='gaussian', semiconv=True, dynamic_K=False,
                   klm_loss=False, detach_mr_in_klm=True, pixel_bound=True,
                   autoreg_prior=True, pixel_std1=0.7, pixel_std2=0.7)


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    return cfg


def train_step(p, opt, geco, x, cfg, rand_pixel=None, eps_k=None, reference_form=True):
    """One training iteration: train.py:223-263.  `p` values must be leaf tensors with
    requires_grad; `opt` a torch.optim.Adam over them.  Returns (elbo, err, kl, beta)."""
    opt.zero_grad()
    _, losses, _, _, _ = v2_forward(p, x, cfg, rand_pixel, eps_k,
                                    reference_form=reference_form)
    err, kl_l, kl_m = aggregate_losses(losses)
    elbo = (err + kl_l + kl_m).detach()
    beta = geco.beta
    loss = geco.loss(err, kl_l + kl_m)
    loss.backward()
    opt.step()
    return float(elbo), float(err), float(kl_l + kl_m), float(beta)


def param_shapes(cfg):
    """Ordered name -> (shape, dtype) of GenesisV2.state_dict() (SURVEY.md Appendix B;
    construction order of models/genesisv2_config.py:51-105, modules/unet.py:23-67)."""
    S, D, K = cfg['img_size'], cfg['feat_dim'], cfg['K_steps']
    nb = int(math.log2(S) - 1)
    c = min(D, 64)
    if nb == 4:
        enc_in, enc_out = [3, c, 2 * c, 2 * c], [c, 2 * c, 2 * c, 2 * c]
        dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c], [2 * c, 2 * c, c, c]
    elif nb == 5:
        enc_in, enc_out = [3, c, c, 2 * c, 2 * c], [c, c, 2 * c, 2 * c, 2 * c]
        dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c, 2 * c], [2 * c, 2 * c, c, c, c]
    elif nb == 6:
        enc_in, enc_out = [3, c, c, c, 2 * c, 2 * c], [c, c, c, 2 * c, 2 * c, 2 * c]
        dec_in, dec_out = [4 * c, 4 * c, 4 * c, 2 * c, 2 * c, 2 * c], [2 * c, 2 * c, c, c, c, c]
    else:
        raise ValueError('img_size must be 32, 64 or 128')
    f32, f64 = torch.float32, torch.float64
    sh = {}
    for i, (a, b) in enumerate(zip(enc_in, enc_out)):
        sh['encoder.down.%d.0.weight' % i] = ((b, a, 3, 3), f32)
        sh['encoder.down.%d.1.weight' % i] = ((b,), f32)
        sh['encoder.down.%d.1.bias' % i] = ((b,), f32)
    for i, (a, b) in enumerate(zip(dec_in, dec_out)):
        sh['encoder.up.%d.0.weight' % i] = ((b, a, 3, 3), f32)
        sh['encoder.up.%d.1.weight' % i] = ((b,), f32)
        sh['encoder.up.%d.1.bias' % i] = ((b,), f32)
    fs = S // 2 ** (nb - 1)
    flat = 2 * c * fs * fs
    for j, (o, i_) in zip((1, 3, 5), ((128, flat), (128, 128), (flat, 128))):
        sh['encoder.mlp.%d.weight' % j] = ((o, i_), f32)
        sh['encoder.mlp.%d.bias' % j] = ((o,), f32)
    sh['att_process.log_sigma'] = ((), f64)
    if cfg.get('semiconv', True):
        sh['att_process.colour_head.conv.weight'] = ((8, D, 1, 1), f32)
        sh['att_process.colour_head.conv.bias'] = ((8,), f32)
        sh['att_process.colour_head.gate.gate'] = ((), f32)
    else:
        sh['att_process.colour_head.weight'] = ((8, D, 1, 1), f32)
        sh['att_process.colour_head.bias'] = ((8,), f32)
    sh['seg_head.0.weight'] = ((D, D, 3, 3), f32)
    sh['seg_head.1.weight'] = ((D,), f32)
    sh['seg_head.1.bias'] = ((D,), f32)
    sh['feat_head.0.0.weight'] = ((D, D, 3, 3), f32)
    sh['feat_head.0.1.weight'] = ((D,), f32)
    sh['feat_head.0.1.bias'] = ((D,), f32)
    sh['feat_head.1.weight'] = ((2 * D, D, 1, 1), f32)
    sh['feat_head.1.bias'] = ((2 * D,), f32)
    sh['z_head.0.weight'] = ((2 * D,), f32)
    sh['z_head.0.bias'] = ((2 * D,), f32)
    for j in (1, 3):
        sh['z_head.%d.weight' % j] = ((2 * D, 2 * D), f32)
        sh['z_head.%d.bias' % j] = ((2 * D,), f32)
    cm = min(D, 64)
    chans = [(D + 2, D), (D, D), (D, cm), (cm, cm)]
    for (ci, gi), (a, b) in zip(((1, 2), (4, 5), (7, 8), (10, 11)), chans):
        sh['decoder_module.%d.weight' % ci] = ((a, b, 5, 5), f32)
        sh['decoder_module.%d.bias' % ci] = ((b,), f32)
        sh['decoder_module.%d.weight' % gi] = ((b,), f32)
        sh['decoder_module.%d.bias' % gi] = ((b,), f32)
    sh['decoder_module.13.weight'] = ((4, cm, 1, 1), f32)
    sh['decoder_module.13.bias'] = ((4,), f32)
    if cfg.get('autoreg_prior', True) and K > 1:
        sh['prior_lstm.weight_ih_l0'] = ((16 * D, D), f32)
        sh['prior_lstm.weight_hh_l0'] = ((16 * D, 4 * D), f32)
        sh['prior_lstm.bias_ih_l0'] = ((16 * D,), f32)
        sh['prior_lstm.bias_hh_l0'] = ((16 * D,), f32)
        sh['prior_linear.weight'] = ((2 * D, 4 * D), f32)
        sh['prior_linear.bias'] = ((2 * D,), f32)
    return sh


def default_log_sigma(cfg):
    """modules/attention.py:145-155: 0-dim log of the kernel bandwidth; float64 for the gaussian /
    laplacian kernels (numpy scalars), float32 for epanechnikov (python float)."""
    import numpy as np
    K = cfg['K_steps']
    kern = cfg.get('kernel', 'gaussian')
    if kern == 'laplacian':
        s = 1.0 / (np.sqrt(K) * np.log(2))
    elif kern == 'gaussian':
        s = 1.0 / (K * np.log(2))
    elif kern == 'epanechnikov':
        s = 2.0 / K
        return torch.tensor(float(s)).log()  # python float -> float32 parameter (attention.py:149)
    else:
        raise ValueError('No valid kernel.')
    return torch.tensor(s).log()  # numpy float64 -> float64 parameter


def template_state_dict(cfg):
    sd = {}
    for name, (shape, dt) in param_shapes(cfg).items():
        sd[name] = torch.zeros(shape, dtype=dt)
    sd['att_process.log_sigma'] = default_log_sigma(cfg)
    return sd
