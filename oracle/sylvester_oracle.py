"""ORACLE (test infrastructure) -- CPU restatement of the gated-convolution VAE the reference vendors under
third_party/sylvester (VAE.py:18-168, layers.py:11-101), used by BaselineVAE (models/vae_config.py) and as the
attention core of GENESIS (models/genesis_config.py:92-99).  Functional, keyed by the reference's state_dict names
under a `prefix`.  BatchNorm runs in training mode (batch statistics), as in the reference's training step."""
import torch
import torch.nn.functional as F

from . import v2_oracle as V


def vae_geometry(img_size):
    """VAE.py:56-69: (last_kernel_size, strides)."""
    if img_size == 32:
        return 8, [1, 2, 1, 2, 1]
    if img_size == 64:
        return 16, [1, 2, 1, 2, 1]
    if img_size == 128:
        return 16, [2, 2, 2, 1, 1]
    if img_size == 256:
        return 16, [2, 2, 2, 2, 1]
    raise ValueError('Invalid input size.')


def _norm(p, name, x, norm, training=True):
    if norm == 'bn':
        return F.batch_norm(x, None if training else p[name + '.running_mean'],
                            None if training else p[name + '.running_var'], p[name + '.weight'], p[name + '.bias'],
                            training, 0.1, 1e-5)
    if norm == 'in':
        return F.instance_norm(x, weight=p[name + '.weight'], bias=p[name + '.bias'], eps=1e-5)
    return x


def gated(p, name, y, norm, training=True):
    """h * sigmoid(g) with optional norms on both halves (layers.py:40-54); training=False = eval-mode BatchNorm
    (running statistics), as the reference's sample() callers run it (train.py:425, scripts/compute_fid.py:104)."""
    h, g = y.chunk(2, dim=1)
    if norm in ('bn', 'in'):
        h = _norm(p, name + '.h_norm', h, norm, training)
        g = _norm(p, name + '.g_norm', g, norm, training)
    return h * torch.sigmoid(g)


def encode(p, x, img_size, prefix, enc_norm):
    """q_z_nn (VAE.py:18-24): 5 gated 5x5 convs (pad 2, strides per size) + gated 'fc' conv of kernel kfc."""
    kfc, strides = vae_geometry(img_size)
    h = x
    for l, s in enumerate(strides):
        name = '%s.q_z_nn.%d' % (prefix, l)
        h = gated(p, name, F.conv2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], s, 2), enc_norm)
    name = '%s.q_z_nn.%d' % (prefix, len(strides))
    h = gated(p, name, F.conv2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], 1, 0), None)
    return h.reshape(h.size(0), -1)                      # [N, 256]


def posterior(p, h, prefix):
    """q_z_mean / q_z_var (VAE.py:106-110): var = to_sigma(linear)**2 (modules/blocks.py:25-26)."""
    mean = F.linear(h, p[prefix + '.q_z_mean.weight'], p[prefix + '.q_z_mean.bias'])
    var = V.to_sigma(F.linear(h, p[prefix + '.q_z_var.0.weight'], p[prefix + '.q_z_var.0.bias'])) ** 2
    return mean, var


def decode(p, z, img_size, prefix, dec_norm, training=True):
    """p_x_nn + p_x_mean (VAE.py:27-33,112-124,143-152): gated deconv kz from 1x1, 5 gated 5x5 deconvs (pad 2,
    output_padding s-1) with the reversed strides, 1x1 conv."""
    kz, strides = vae_geometry(img_size)
    h = z.view(z.size(0), -1, 1, 1)
    name = prefix + '.p_x_nn.0'
    h = gated(p, name, F.conv_transpose2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], 1, 0), None)
    for l, s in enumerate(reversed(strides)):
        name = '%s.p_x_nn.%d' % (prefix, l + 1)
        h = gated(p, name, F.conv_transpose2d(h, p[name + '.conv.weight'], p[name + '.conv.bias'], s, 2, s - 1),
                  dec_norm, training)
    return F.conv2d(h, p[prefix + '.p_x_mean.weight'], p[prefix + '.p_x_mean.bias'])


def param_shapes(prefix, z_size, nin, img_size, nout, enc_norm, dec_norm):
    """Ordered name -> (shape, dtype) for sylvester.VAE under `prefix` (construction order VAE.py:73-83)."""
    f32, i64 = torch.float32, torch.int64
    kfc, strides = vae_geometry(img_size)
    sh = {}

    def norms(name, c, norm):
        for hn in ('h_norm', 'g_norm'):
            if norm == 'bn':
                sh['%s.%s.weight' % (name, hn)] = ((c,), f32)
                sh['%s.%s.bias' % (name, hn)] = ((c,), f32)
                sh['%s.%s.running_mean' % (name, hn)] = ((c,), f32)
                sh['%s.%s.running_var' % (name, hn)] = ((c,), f32)
                sh['%s.%s.num_batches_tracked' % (name, hn)] = ((), i64)
            elif norm == 'in':
                sh['%s.%s.weight' % (name, hn)] = ((c,), f32)
                sh['%s.%s.bias' % (name, hn)] = ((c,), f32)

    cin, cout = [nin, 32, 32, 64, 64], [32, 32, 64, 64, 64]
    for l, (a, b) in enumerate(zip(cin, cout)):
        name = '%s.q_z_nn.%d' % (prefix, l)
        sh[name + '.conv.weight'] = ((2 * b, a, 5, 5), f32)
        sh[name + '.conv.bias'] = ((2 * b,), f32)
        norms(name, b, enc_norm)
    name = '%s.q_z_nn.5' % prefix
    sh[name + '.conv.weight'] = ((512, 64, kfc, kfc), f32)
    sh[name + '.conv.bias'] = ((512,), f32)
    sh[prefix + '.q_z_mean.weight'] = ((z_size, 256), f32)
    sh[prefix + '.q_z_mean.bias'] = ((z_size,), f32)
    sh[prefix + '.q_z_var.0.weight'] = ((z_size, 256), f32)
    sh[prefix + '.q_z_var.0.bias'] = ((z_size,), f32)
    name = prefix + '.p_x_nn.0'
    sh[name + '.conv.weight'] = ((z_size, 128, kfc, kfc), f32)
    sh[name + '.conv.bias'] = ((128,), f32)
    cin, cout = [64, 64, 32, 32, 32], [64, 32, 32, 32, 32]
    for l, (a, b) in enumerate(zip(cin, cout)):
        name = '%s.p_x_nn.%d' % (prefix, l + 1)
        sh[name + '.conv.weight'] = ((a, 2 * b, 5, 5), f32)
        sh[name + '.conv.bias'] = ((2 * b,), f32)
        norms(name, b, dec_norm)
    sh[prefix + '.p_x_mean.weight'] = ((nout, 32, 1, 1), f32)
    sh[prefix + '.p_x_mean.bias'] = ((nout,), f32)
    return sh
