"""Gated-convolution VAE of third_party/sylvester (VAE.py:36-168, layers.py:11-101) on the HIP path: parameter
containers with the reference's key layout / init order, and the autograd Functions that run its layers through the
C ABI (direct (de)convolution kernels + the fused gated-norm kernel; the two 'fc' layers -- a kfc x kfc valid conv
from a kfc x kfc map and its transpose from a 1x1 map -- are plain GEMMs and go through the library)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops as hip
from . import functions as fn
from .functions import _gout, _ret, ctx_bound


def _pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def vae_geometry(img_size):
    """VAE.py:56-69 -> (last_kernel_size, strides)."""
    table = {32: (8, [1, 2, 1, 2, 1]), 64: (16, [1, 2, 1, 2, 1]), 128: (16, [2, 2, 2, 1, 1]), 256: (16, [2, 2, 2, 2, 1])}
    if img_size not in table:
        raise ValueError('Invalid input size.')
    return table[img_size]


# ------------------------------------------------------------------ parameter containers
class _Gated(nn.Module):
    def __init__(self, conv, cout, h_norm, g_norm):
        super().__init__()
        self.conv = conv
        self.norm = h_norm if h_norm in ('bn', 'in') else None
        mk = {'bn': lambda: nn.BatchNorm2d(cout), 'in': lambda: nn.InstanceNorm2d(cout, affine=True)}
        self.h_norm = mk[h_norm]() if h_norm in mk else None
        self.g_norm = mk[g_norm]() if g_norm in mk else None


def GatedConv2d(cin, cout, k, s, p, h_norm=None, g_norm=None):
    return _Gated(nn.Conv2d(cin, 2 * cout, k, s, p), cout, h_norm, g_norm)


def GatedConvTranspose2d(cin, cout, k, s, p, op=0, h_norm=None, g_norm=None):
    return _Gated(nn.ConvTranspose2d(cin, 2 * cout, k, s, p, op), cout, h_norm, g_norm)


class _ToVar(nn.Module):
    pass


class SylvesterVAE(nn.Module):
    """Parameter container + HIP forward for sylvester.VAE (construction order of VAE.py:73-83)."""

    def __init__(self, z_size, input_size, nout, enc_norm=None, dec_norm=None):
        super().__init__()
        self.z_size = z_size
        self.img_size = input_size[1]
        self.nout = nout if nout is not None else input_size[0]
        self.enc_norm, self.dec_norm = enc_norm, dec_norm
        self.last_kernel_size, strides = vae_geometry(self.img_size)
        self.strides = strides
        cin, cout = [input_size[0], 32, 32, 64, 64], [32, 32, 64, 64, 64]
        layers = [GatedConv2d(i, o, 5, s, 2, enc_norm, enc_norm) for i, o, s in zip(cin, cout, strides)]
        layers.append(GatedConv2d(cout[-1], 256, self.last_kernel_size, 1, 0))
        self.q_z_nn = nn.Sequential(*layers)
        self.q_z_mean = nn.Linear(256, z_size)
        self.q_z_var = nn.Sequential(nn.Linear(256, z_size), _ToVar())
        cin, cout = [64, 64, 32, 32, 32], [64, 32, 32, 32, 32]
        rs = list(reversed(strides))
        layers = [GatedConvTranspose2d(z_size, cin[0], self.last_kernel_size, 1, 0)]
        layers += [GatedConvTranspose2d(i, o, 5, s, 2, s - 1, dec_norm, dec_norm) for i, o, s in zip(cin, cout, rs)]
        self.p_x_nn = nn.Sequential(*layers)
        self.p_x_mean = nn.Conv2d(cout[-1], self.nout, 1, 1, 0)

    # -------------------------------------------------------------- HIP forward pieces
    def _gate(self, unit, y):
        return gate_unit(unit, y, self.training)

    def encode_features(self, x):
        """q_z_nn -> [N, 256]."""
        h = x
        for l, s in enumerate(self.strides):
            unit = self.q_z_nn[l]
            h = self._gate(unit, DirectConvFn.apply(h, unit.conv.weight, 'conv', s, 2, 0))
        unit = self.q_z_nn[len(self.strides)]
        y = fn.linear(h.flatten(1), unit.conv.weight).view(h.shape[0], -1, 1, 1)
        return self._gate(unit, y).flatten(1)

    def posterior_heads(self, h):
        """[N, 2 z]: (q_z_mean(h) | q_z_var's Linear(h)), i.e. (mu | pre-sigma) -- ToVar (to_sigma(.)**2, blocks.py:22-26) is
        applied by the consumer: the posterior kernel forms sigma = to_sigma(pre-sigma) = sqrt(var) directly
        (vae_config.py:66 takes var.sqrt())."""
        return fn.TwoHeadLinearFn.apply(h, self.q_z_mean.weight, self.q_z_mean.bias, self.q_z_var[0].weight, self.q_z_var[0].bias)

    def posterior(self, h):
        """(mean, var) of VAE.py:118-121 (the reference's encode()); the training path uses posterior_heads."""
        zh = self.posterior_heads(h)
        z = zh.shape[1] // 2
        _, mean, sigma, _ = fn.PosteriorFn.apply(zh.unsqueeze(1), zh.new_zeros(1, zh.shape[0], z))
        return mean[0], sigma[0] * sigma[0]

    def decode(self, z):
        unit = self.p_x_nn[0]
        k = self.last_kernel_size
        w = unit.conv.weight                                           # [z, 128, k, k]
        y = fn.MatmulNNFn.apply(z, w).view(z.shape[0], w.shape[1], k, k)
        h = self._gate(unit, y)
        for l, s in enumerate(reversed(self.strides)):
            unit = self.p_x_nn[l + 1]
            h = self._gate(unit, DirectConvFn.apply(h, unit.conv.weight, 'deconv', s, 2, s - 1))
        from .functions import Conv1x1Fn
        return Conv1x1Fn.apply(h, self.p_x_mean.weight, self.p_x_mean.bias)


def gate_unit(unit, y, training):
    """norm_h(h + b) * sigmoid(norm_g(g + b)) of one gated (de)conv unit (layers.py:40-101); BatchNorm running
    statistics are updated like nn.BatchNorm2d."""
    norm = unit.norm
    if norm == 'bn' and not training:
        # evaluation mode: running statistics (not the training hot path): plain pointwise ops
        h, g = (y + unit.conv.bias.view(1, -1, 1, 1)).chunk(2, 1)
        return unit.h_norm(h) * torch.sigmoid(unit.g_norm(g))
    args = (unit.h_norm.weight, unit.h_norm.bias, unit.g_norm.weight, unit.g_norm.bias) if norm else (None,) * 4
    if norm == 'bn' and not _SYNC['on']:
        # the running statistics are updated by the unit's own apply kernel (the library falls back to a launch of its own)
        assert unit.h_norm.momentum == 0.1 and unit.g_norm.momentum == 0.1
        hip.gated_bn_running_arm(y.shape[1] // 2, unit.h_norm, unit.g_norm, momentum=0.1)
        out, stats = GatedNormFn.apply(y, unit.conv.bias, norm, *args)
        return out
    out, stats = GatedNormFn.apply(y, unit.conv.bias, norm, *args)
    if norm == 'bn':
        # running = 0.9 running + 0.1 {mean, unbiased variance}, num_batches_tracked += 1 (momentum None is not used by
        # the reference's stacks)
        assert unit.h_norm.momentum == 0.1 and unit.g_norm.momentum == 0.1
        m = y.shape[0] * y.shape[2] * y.shape[3]
        if _SYNC['on']:
            m *= _world()      # (equal shards: the count the statistics were taken over)
        hip.bn_running_update(stats, out.shape[1], m, unit.h_norm, unit.g_norm, eps=unit.h_norm.eps)
    return out


def gc_encoder_forward(units, x, strides, training):
    """sylvester.build_gc_encoder (VAE.py:18-24) -> [N, cfc]: gated 5x5 convs (pad 2) + the gated 'fc' conv."""
    h = x
    for unit, s in zip(units, strides):
        h = gate_unit(unit, DirectConvFn.apply(h, unit.conv.weight, 'conv', s, 2, 0), training)
    unit = units[len(strides)]
    y = fn.linear(h.flatten(1), unit.conv.weight).view(h.shape[0], -1, 1, 1)
    return gate_unit(unit, y, training).flatten(1)


def gc_decoder_forward(units, z, strides, training):
    """sylvester.build_gc_decoder (VAE.py:27-33): gated deconv kz from 1x1, then gated 5x5 deconvs (pad 2, out_pad s-1)."""
    unit = units[0]
    w = unit.conv.weight                                           # [z, 2c, k, k]
    k = w.shape[2]
    h = gate_unit(unit, fn.MatmulNNFn.apply(z, w).view(z.shape[0], w.shape[1], k, k), training)
    for l, s in enumerate(strides):
        unit = units[l + 1]
        h = gate_unit(unit, DirectConvFn.apply(h, unit.conv.weight, 'deconv', s, 2, s - 1), training)
    return h


# ------------------------------------------------------------------ autograd Functions
@ctx_bound
class DirectConvFn(torch.autograd.Function):
    """Bias-free Conv2d / ConvTranspose2d through the generic direct kernels.  A ConvTranspose2d (weight
    [Cin, Cout, k, k]) is the data-gradient of the Conv2d with the same weight tensor, and vice versa."""

    @staticmethod
    def forward(ctx, x, w, kind, stride, pad, out_pad):
        x = x.contiguous()
        k = w.shape[2]
        H, W = x.shape[2], x.shape[3]
        # the stride-2 5x5 layers of the gated stacks (VAE.py:18-33: k 5, pad 2, output_padding 1) are exactly the
        # GENESIS-V2 decoder's transposed conv and its data gradient: they run on those MFMA kernels (gx_deconv5x5s2_*,
        # weight gradients in the step's stream-K launch) instead of the generic implicit-GEMM kernels
        fast = k == 5 and stride == 2 and pad == 2 and _pow2(H) and _pow2(W) and min(H, W) >= 4 and \
            (out_pad == 1 if kind == 'deconv' else True)
        # the stride-1 5x5 layers: the tap-conv MFMA kernel with a 25-tap table (gx_conv5x5s1)
        Cout = w.shape[0] if kind == 'conv' else w.shape[1]
        s1 = k == 5 and stride == 1 and pad == 2 and out_pad == 0 and hip.conv5x5s1_supported(x.shape[0], x.shape[1], Cout, H, W) \
            and hip.conv5x5s1_supported(x.shape[0], Cout, x.shape[1], H, W)
        if kind == 'conv':
            y = hip.deconv5x5s2_dgrad(x, w) if fast else (hip.conv5x5s1(x, w, Cout, False) if s1 else
                                                          hip.conv2d_direct_fwd(x, w, None, None, stride, pad))
        else:
            Ho, Wo = (H - 1) * stride - 2 * pad + k + out_pad, (W - 1) * stride - 2 * pad + k + out_pad
            y = hip.deconv5x5s2_fwd(x, w, None) if fast else (hip.conv5x5s1(x, w, Cout, True) if s1 else
                                                              hip.conv2d_direct_dgrad(x, w, Ho, Wo, stride, pad))
        ctx.save_for_backward(x)
        ctx.w = w
        ctx.cfg = (kind, stride, pad, k, fast, s1)
        ctx.am_x = getattr(x, '_gx_amax', None)      # (partial maxima of the input, left by the gated unit that produced it)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        w = ctx.w
        kind, stride, pad, k, fast, s1 = ctx.cfg
        am_g, am_x = getattr(g, '_gx_amax', None), ctx.am_x       # (the gated unit's backward left its dy's partial maxima on it)
        g = g.contiguous()
        ow = _gout(w)
        need_dx = ctx.needs_input_grad[0]
        # (weight gradients: both operands' maxima known -> three fp16 piece products instead of six bf16 ones; the data
        #  gradients below find g's maxima through the link the gated unit's backward armed)
        if fast and kind == 'conv':
            # conv s2 = the transposed conv's data gradient with (x, dy) in each other's roles
            dw = hip.deconv5x5s2_wgrad(g, x, out=ow, amax=(am_x, am_g))
            dx = hip.deconv5x5s2_fwd(g, w, None) if need_dx else None
        elif fast:
            dw = hip.deconv5x5s2_wgrad(x, g, out=ow, amax=(am_g, am_x))
            dx = hip.deconv5x5s2_dgrad(g, w) if need_dx else None
        elif kind == 'conv':
            if k == 5 and stride == 1 and pad == 2 and hip.conv5x5_wgrad_supported(x.shape[0], w.shape[0], w.shape[1], *x.shape[2:]):
                dw = hip.conv5x5_wgrad(g, x, out=ow, amax=(am_g, am_x))          # row-ring tiles, in the step's stream-K launch
            else:
                dw = hip.conv2d_direct_wgrad(x, g, k, stride, pad, out=ow)
            dx = (hip.conv5x5s1(g, w, x.shape[1], True) if s1 else
                  hip.conv2d_direct_dgrad(g, w, x.shape[2], x.shape[3], stride, pad)) if need_dx else None
        else:
            if k == 5 and stride == 1 and pad == 2 and hip.conv5x5_wgrad_supported(x.shape[0], w.shape[0], w.shape[1], *x.shape[2:]):
                dw = hip.conv5x5_wgrad(x, g, out=ow, amax=(am_x, am_g))
            else:
                dw = hip.conv2d_direct_wgrad(g, x, k, stride, pad, out=ow)
            dx = (hip.conv5x5s1(g, w, x.shape[1], False) if s1 else
                  hip.conv2d_direct_fwd(g, w, None, None, stride, pad)) if need_dx else None
        return dx, _ret(ow, dw), None, None, None, None


# Cross-replica BatchNorm (the reference's single-device batch statistics, genesis_config.py:39-40 ->
# layers.py:26-27, at the GLOBAL batch when the batch is sharded over ranks): `sync_bn(group)` -- TrainStep calls it when
# GENESIS_SYNC_BN=1 and the process group has more than one rank -- makes every training-mode BatchNorm of the gated stacks
# take its statistics (forward) and its two gradient sums (backward) over all ranks: two small all-reduces per gated unit and
# step (4C doubles, 4C floats).  Off (the default): per-replica statistics, what nn.DataParallel gives the reference
# (train.py:153-155).
_SYNC = {'group': None, 'on': False}


def sync_bn(group=None, on=True):
    _SYNC['group'], _SYNC['on'] = group, bool(on)


def _world():
    import torch.distributed as dist
    return dist.get_world_size(_SYNC['group'])


def _all_reduce_sum(t, info=None):
    """t (a device tensor of a few hundred bytes) <- its sum over the ranks, the same bits on every rank.  A backend that carries
    host memory (gloo: the two-ranks-on-one-GPU tests): staged explicitly -- device -> host, all_gather, the ranks' parts added in
    RANK ORDER by every rank, host -> device.  GENESIS_SYNC_BN_DEBUG=1 sends the call site's identity along and checks that every
    rank is at the same one."""
    import os
    import torch.distributed as dist
    grp = _SYNC['group']
    fake = os.environ.get('GENESIS_SYNC_BN_FAKE')
    if fake:                  # (diagnosis: the partner is a copy of this rank; 2: ... behind a host-blocking barrier)
        if fake == '2':
            torch.cuda.synchronize()
            dist.barrier(group=grp)
        t.mul_(dist.get_world_size(grp))
        return
    if t.is_cuda and dist.get_backend(grp) != 'nccl':
        h = t.detach().cpu().contiguous()
        debug = os.environ.get('GENESIS_SYNC_BN_DEBUG') == '1'
        if debug:
            _SYNC['seq'] = _SYNC.get('seq', 0) + 1
            tag = torch.tensor([float(_SYNC['seq'])] + [float(v) for v in (info or ())], dtype=h.dtype)
            h = torch.cat([h.flatten(), tag])
        parts = [torch.empty_like(h) for _ in range(dist.get_world_size(grp))]
        dist.all_gather(parts, h, group=grp)
        if debug:
            n = tag.numel()
            for r, q in enumerate(parts):
                if not torch.equal(q[-n:], tag):
                    print('SYNC MISMATCH: rank %d is at %s, rank %d at %s' % (dist.get_rank(grp), tag.tolist(), r, q[-n:].tolist()), flush=True)
            parts = [q[:-n] for q in parts]
        acc = parts[0].clone()
        for q in parts[1:]:
            acc += q
        if _SYNC.get('log') is not None:
            _SYNC['log'].append((tuple(info or ()), [float(q.double().sum()) for q in parts], float(acc.double().abs().sum())))
        t.copy_(acc.view(t.shape))
        return
    dist.all_reduce(t, group=grp)


@ctx_bound
class GatedNormFn(torch.autograd.Function):
    """returns (out [N,C,H,W], stats); stats ({mean, rstd} per unit) is non-differentiable."""

    @staticmethod
    def forward(ctx, y, bias, norm, gh, bh, gg, bg):
        y = y.contiguous()
        ctx.m_global = None
        if norm == 'bn' and _SYNC['on']:
            out, stats, ctx.m_global = hip.gated_bn_sync_fwd(y, bias, gh, bh, gg, bg, lambda t: _all_reduce_sum(t, (1,) + tuple(y.shape)), _world())
        else:
            out, stats = hip.gated_norm_fwd(y, bias, norm, gh, bh, gg, bg)
            out._gx_amax = hip.take_amax()
        ctx.save_for_backward(y, stats)
        ctx.params = (bias, gh, bh, gg, bg)
        ctx.norm = norm
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)      # (no zero-filled gradient tensor for `stats` in every backward)
        return out, stats

    @staticmethod
    def backward(ctx, g, _unused):
        y, stats = ctx.saved_tensors
        bias, gh, bh, gg, bg = ctx.params
        outs = tuple(_gout(p) if p is not None else None for p in (gh, bh, gg, bg, bias))
        if ctx.m_global is not None:
            dy, dgh, dbh, dgg, dbg, dbias = hip.gated_bn_sync_bwd(y, bias, gh, bh, gg, bg, stats, g.contiguous(), ctx.m_global,
                                                                  lambda t: _all_reduce_sum(t, (2,) + tuple(y.shape)), out=outs)
        else:
            dy, dgh, dbh, dgg, dbg, dbias = hip.gated_norm_bwd(y, bias, ctx.norm, gh, bh, gg, bg, stats, g.contiguous(), out=outs)
            dy._gx_amax = hip.take_amax()
        return (dy, _ret(outs[4], dbias), None, _ret(outs[0], dgh), _ret(outs[1], dbh), _ret(outs[2], dgg),
                _ret(outs[3], dbg))
