"""Deterministic, platform-stable test vectors shared by the golden-fixture generator,
the parity tests and smoke(): closed-form ("formula") weights for any state_dict layout
and summaries (sum / abs-sum / strided samples) of large tensors.  No model arithmetic."""
import numpy as np
import torch


def formula_tensor(index, shape, dtype=torch.float32, kind='weight'):
    n = int(np.prod(shape)) if len(shape) else 1
    t = np.sin(0.61803398875 * np.arange(n, dtype=np.float64) * (1.0 + 0.013 * index)
               + 0.7 * index + 0.25)
    if kind == 'matrix':
        fan_in = max(1, n // shape[0])
        t = t * (1.5 / np.sqrt(fan_in))
    elif kind == 'scale':
        t = 1.0 + 0.2 * t
    elif kind == 'bias':
        t = 0.1 * t
    return torch.from_numpy(t.reshape(shape)).to(dtype)


def formula_state_dict(template):
    """template: mapping name -> tensor (only shape/dtype are used).  Returns a new
    state_dict filled by closed-form functions of (position, name, shape)."""
    out = {}
    for i, (name, v) in enumerate(template.items()):
        shape = tuple(v.shape)
        if name == 'std' or name.endswith('num_batches_tracked'):
            out[name] = v.detach().clone()      # registered buffers that are not weights
        elif name.endswith('running_var'):
            out[name] = formula_tensor(i, shape, v.dtype, 'scale')
        elif name.endswith('running_mean'):
            out[name] = formula_tensor(i, shape, v.dtype, 'bias')
        elif name.endswith('log_sigma'):
            out[name] = (v.detach().clone().double() + 0.1).to(v.dtype)
        elif name.endswith('gate.gate'):
            out[name] = torch.tensor(0.35, dtype=v.dtype)
        elif len(shape) >= 2:
            if 'decoder_module' in name and len(shape) == 4 and shape[-1] == 5:
                # ConvTranspose2d weight (Cin,Cout,5,5): fan-in per output ~ Cin*25/4
                t = formula_tensor(i, shape, v.dtype, 'weight') * (1.5 / np.sqrt(shape[0] * 6.25))
                out[name] = t.to(v.dtype)
            else:
                out[name] = formula_tensor(i, shape, v.dtype, 'matrix')
        elif name.endswith('bias') or 'bias_' in name:
            out[name] = formula_tensor(i, shape, v.dtype, 'bias')
        else:
            out[name] = formula_tensor(i, shape, v.dtype, 'scale')
    return out


def draw_noise(seed, B, S, D, K):
    """Replays the reference's RNG consumption order after torch.manual_seed(seed):
    one uniform [B,1,S,S] (modules/attention.py:177-178) then K normals [B,D]
    (models/genesisv2_config.py:157), on the default CPU generator."""
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    rand_pixel = torch.empty(B, 1, S, S).uniform_()
    eps_k = [torch.normal(torch.zeros(B, D), torch.ones(B, D)) for _ in range(K)]
    torch.set_rng_state(state)
    return rand_pixel, eps_k


def make_input(seed, B, S):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, S, S, generator=g)


RECT_LEVELS = (0.0, 63.0 / 255.0, 127.0 / 255.0, 191.0 / 255.0, 1.0)


def make_rect_input(seed, B, S):
    """The STRUCTURED input set of SURVEY section 8(d): images with the value distribution of Multi-dSprites -- a flat background
    and one to four flat objects, every colour channel one of the five levels {0, 63, 127, 191, 255} / 255
    (scripts/generate_multid.py:32-34 the levels, :48-49 the flat background, :75 the / 255) -- with axis-aligned rectangles in
    place of the dSprites shapes (which are a dataset download).  Exact zeros, exact ones and large constant regions: the
    adversarial case for per-tensor fp16 operand scales and for ReLU decisions at pre-activation 0.  Closed form, seeded."""
    g = torch.Generator().manual_seed(seed)
    lv = torch.tensor(RECT_LEVELS, dtype=torch.float32)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))   # noqa: E731
    x = torch.empty(B, 3, S, S, dtype=torch.float32)
    for b in range(B):
        x[b] = lv[torch.randint(0, 5, (3,), generator=g)].view(3, 1, 1)
        for _ in range(ri(1, 4)):
            h, w = ri(S // 8, S // 2), ri(S // 8, S // 2)
            y0, x0 = ri(0, S - h), ri(0, S - w)
            x[b, :, y0:y0 + h, x0:x0 + w] = lv[torch.randint(0, 5, (3,), generator=g)].view(3, 1, 1)
    return x


def make_input_of(kind, seed, B, S):
    return make_rect_input(seed, B, S) if kind == 'rect' else make_input(seed, B, S)


MAX_SAMPLES = 2048


def summarize(t):
    """-> dict(sum, asum, stride, samples) in float64/float32 numpy."""
    t = t.detach().to('cpu')
    flat = t.reshape(-1)
    n = flat.numel()
    stride = max(1, n // MAX_SAMPLES)
    d = flat.double()
    return dict(sum=np.float64(d.sum().item()), asum=np.float64(d.abs().sum().item()),
                stride=np.int64(stride), n=np.int64(n),
                samples=flat[::stride].float().numpy().copy())


def pack_summary(prefix, t, out):
    s = summarize(t)
    for k, v in s.items():
        out['%s/%s' % (prefix, k)] = v


def check_summary(prefix, t, gold, rtol, atol, what=''):
    """Compares tensor `t` against a stored summary; raises AssertionError."""
    s = summarize(t)
    assert int(gold[prefix + '/n']) == int(s['n']), (prefix, 'numel', s['n'])
    ref = gold[prefix + '/samples']
    np.testing.assert_allclose(s['samples'], ref, rtol=rtol, atol=atol,
                               err_msg='%s %s samples' % (what, prefix))
    scale = float(gold[prefix + '/asum']) + 1e-30
    assert abs(float(s['sum']) - float(gold[prefix + '/sum'])) <= rtol * scale + atol * int(s['n']), \
        '%s %s sum %r vs %r' % (what, prefix, s['sum'], gold[prefix + '/sum'])
    assert abs(float(s['asum']) - scale) <= rtol * scale + atol * int(s['n']), \
        '%s %s asum' % (what, prefix)
