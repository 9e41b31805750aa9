"""Raw (non-autograd) Python entry points of the HIP kernels: torch tensors in, torch tensors out,
every call goes through the C ABI in include/genesis_hip.h on torch's current HIP stream.

PyTorch is used for device memory (caching allocator) and streams only.  There is no CPU or eager
fallback: non-HIP tensors raise."""
import ctypes
import os

import torch

from . import _lib
from ._lib import GenesisHipError

F32 = torch.float32
KERNELS = {'gaussian': 0, 'laplacian': 1, 'epanechnikov': 2}


def _chk(t, name, dtype=F32):
    if t is None:
        return
    if not t.is_cuda:
        raise GenesisHipError('%s: the HIP hot path needs device tensors (got %s); there is no CPU fallback'
                              % (name, t.device))
    if t.dtype != dtype:
        raise GenesisHipError('%s: expected %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise GenesisHipError('%s: tensor must be contiguous' % name)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """The current HIP stream of the current device as a C pointer (every launch asks: the raw handle, not a torch.cuda.Stream
    object built for the occasion)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# GENESIS_POISON=1 (diagnosis; tools/diag_poison.py, tests/test_poison_gpu.py): every tensor allocated with torch.empty / empty_like
# from here on -- outputs, workspaces, scratch -- starts as NaN, so a kernel that reads memory nobody wrote shows up in the results
# instead of depending on what the allocator happened to hand out.
if os.environ.get('GENESIS_POISON') == '1' and not getattr(torch, '_gx_poisoned', False):
    _real_empty, _real_empty_like = torch.empty, torch.empty_like

    def _poison(t):
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float('nan'))
            elif t.dtype == torch.uint8:
                t.fill_(0xFF)          # (read as floats: NaN)
        return t

    torch.empty = lambda *a, **k: _poison(_real_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_real_empty_like(*a, **k))
    torch._gx_poisoned = True


# ---- the operands' largest magnitudes for the fp16-piece weight gradients (gx_wgq_operand_amax, include/genesis_hip.h) ----------
# The stream-K weight-gradient launch at the end of a backward pass forms fp32 products from three fp16 piece products where it
# knows max |x| and max |dy| of a layer -- from the GroupNorm kernels that WROTE those tensors (gx_amax_tap: one partial maximum per
# workgroup), never from a pass of its own.  The wrappers below tap every GroupNorm forward / backward launch whose output feeds a
# conv (take_amax() hands the caller an `Amax`), and conv3x3_wgrad / deconv5x5s2_wgrad pass the handles of their two operands on.
# Partial maxima live in a ring arena (one allocation per device, far larger than what one iteration produces: they are consumed by
# the flush of the same iteration).  GENESIS_WGQ_F16X3=0: nothing is tapped, the weight gradients stay on six bf16 piece products.
WGQ_F16 = os.environ.get('GENESIS_WGQ_F16X3', '1') != '0'
_TAP_CAP = 16384         # (the 128 x 128 decoder head's chunked backward: K B x 8 groups x 4 chunks = 11 264 workgroups)
_ARENA_FLOATS = 1 << 22
_ARENA = {}                      # device index -> [tensor, base pointer, position (floats)]
import threading as _threading
_LAST = _threading.local()       # .amax: what the last tapped launch OF THIS THREAD left (two loops in two threads must not take each other's)


class Amax(object):
    """Partial maxima of a tensor: `n` floats at device address `ptr` (inside the arena)."""
    __slots__ = ('ptr', 'n')

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n


def _amax_scratch(device, n):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    a = _ARENA.get(idx)
    if a is None:
        t = torch.zeros(_ARENA_FLOATS, dtype=F32, device=device)
        a = _ARENA[idx] = [t, t.data_ptr(), 0]
    n = (n + 3) & ~3
    if a[2] + n > _ARENA_FLOATS:
        a[2] = 0
    off = a[2]
    a[2] = off + n
    return a[1] + 4 * off, a, off


def take_amax():
    """The partial maxima the LAST tapped GroupNorm launch left for the tensor it stored (None: not tapped / not served)."""
    h = getattr(_LAST, 'amax', None)
    _LAST.amax = None
    return h


def _tap_begin(device, hw, numel):
    # (hw: the plane size of a GroupNorm launch; the gated units' kernels tap at any size and pass 1 << 30)
    if not WGQ_F16 or hw < 256:      # (the register-resident GroupNorm kernels -- the only producers -- start at 16 x 16 planes)
        _LAST.amax = None
        return None
    ptr, a, off = _amax_scratch(device, _TAP_CAP)
    _lib.call('gx_amax_tap', ctypes.c_void_p(ptr), _TAP_CAP, int(numel))
    return ptr, a, off


def _tap_end(tap):
    if tap is None:
        return
    ptr, a, off = tap
    n = _lib.query('gx_amax_tap_result')
    if a[2] == off + _TAP_CAP:
        a[2] = off + ((n + 3) & ~3)          # give the unused part of the request back
    _LAST.amax = Amax(ptr, n) if n > 0 else None


def amax_values(h, device=None):
    """The partial maxima behind an Amax handle as a tensor (a view of the arena: read it before the ring wraps) -- tests."""
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    a = _ARENA[idx]
    off = (h.ptr - a[1]) // 4
    assert 0 <= off and off + h.n <= _ARENA_FLOATS, 'not an arena handle'
    return a[0][off:off + h.n]


def amax_of(t):
    """Partial maxima of any tensor by a pass of its own (gx_amax_parts: 256 floats) -- tests, and operands no producer taps."""
    _chk(t, 'amax_of.t')
    ptr, _, _ = _amax_scratch(t.device, 256)
    _lib.call('gx_amax_parts', _p(t), t.numel(), ctypes.c_void_p(ptr), _stream())
    return Amax(ptr, 256)


class _operand_amax(object):
    """with _operand_amax(device, amax): ... one weight-gradient call that takes the hint.  amax = (dy's Amax | [Amax, Amax],
    x's Amax | [Amax, Amax]) or None; an operand whose maxima are not all known disables the hint."""

    def __init__(self, device, amax):
        self.on = False
        if not WGQ_F16 or amax is None:
            return
        ops = []
        for h in amax:
            hs = list(h) if isinstance(h, (list, tuple)) else [h]
            if not hs or len(hs) > 2 or any(x is None for x in hs):
                return
            ops.append(hs + [None] * (2 - len(hs)))
        out, _, _ = _amax_scratch(device, 4)
        args = []
        for hs in ops:
            for x in hs:
                args += [ctypes.c_void_p(x.ptr) if x is not None else None, x.n if x is not None else 0]
        self.args = args + [ctypes.c_void_p(out)]
        self.on = True

    def __enter__(self):
        if self.on:
            _lib.call('gx_wgq_operand_amax', *self.args)

    def __exit__(self, *exc):
        if self.on:
            _lib.call('gx_wgq_operand_amax', None, 0, None, 0, None, 0, None, 0, None)      # (a call that queued nothing leaves it armed)
        return False


WINO_F16 = os.environ.get('GENESIS_WINO_F16X3', '1') != '0'
_AMAX_DEBUG = os.environ.get('GENESIS_AMAX_DEBUG') == '1'      # prints the partial-maxima counts every conv3x3 call was handed


class _input_amax(object):
    """with _input_amax(handles): ... one conv3x3 forward / data-gradient call whose INPUT tensor's partial maxima are `handles`
    (an Amax, or a list of up to two: a concat buffer, the pair data gradient's two tensors) -- gx_conv_input_amax: a layer that
    takes the Winograd kernel then runs on two fp16 pieces per operand.  Any unknown part disables the hint."""

    def __init__(self, handles):
        self.on = False
        if _AMAX_DEBUG and handles is None:
            import traceback
            print('conv input maxima: unknown  <- %s' % ' / '.join('%s:%d' % (f.name, f.lineno) for f in traceback.extract_stack()[-4:-1]), flush=True)
        if not WINO_F16 or handles is None:
            return
        hs = list(handles) if isinstance(handles, (list, tuple)) else [handles]
        if _AMAX_DEBUG:
            import traceback
            print('conv input maxima: %s  <- %s' % ([None if h is None else h.n for h in hs],
                                                     ' / '.join('%s:%d' % (f.name, f.lineno) for f in traceback.extract_stack()[-4:-1])), flush=True)
        if not hs or len(hs) > 2 or any(h is None for h in hs):
            return
        # (how many partial maxima a kernel takes is the library's decision: the Winograd kernel up to 1536 -- its workgroups reduce
        #  them themselves --, the <= 32-output-channel kernel any number, folded by one small launch)
        hs = hs + [None] * (2 - len(hs))
        self.args = [ctypes.c_void_p(hs[0].ptr), hs[0].n, ctypes.c_void_p(hs[1].ptr) if hs[1] is not None else None,
                     hs[1].n if hs[1] is not None else 0]
        self.on = True

    def __enter__(self):
        if self.on:
            _lib.call('gx_conv_input_amax', *self.args)
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib.call('gx_conv_input_amax', None, 0, None, 0)       # (a call that took another kernel leaves it armed)
        return False


# ---- deferred parameter-gradient reductions (gx_defer_*): TrainStep turns this on for one backward pass; calls that
# write a parameter gradient into a caller-provided buffer then queue their final reduce, and defer_flush() finishes
# all of them in one launch per kind.  Workspaces of queued calls are kept alive here until the flush.
class _DeferState(object):
    __slots__ = ('on', 'keep')

    def __init__(self):
        self.on = False          # TrainStep turns this on for one backward pass
        self.keep = []           # workspaces / operands of queued calls, alive until the flush


_DEFER = {}                      # library context id -> _DeferState (one per training loop)


def defer_state():
    return _DEFER.setdefault(_lib.current_ctx(), _DeferState())


class _deferring(object):
    def __init__(self, active, *keep):
        self.st = defer_state()
        self.active = bool(active) and self.st.on
        self.keep = keep

    def __enter__(self):
        if self.active:
            _lib.call('gx_defer_enable', 1)
        return self

    def __exit__(self, *exc):
        if self.active:
            _lib.call('gx_defer_enable', 0)
            self.st.keep.extend(self.keep)
        return False


def defer_flush():
    if _lib.query('gx_defer_pending'):
        _lib.call('gx_defer_flush', _stream())
    del defer_state().keep[:]


def defer_discard():
    """Drops queued reductions without running them (an exception interrupted the backward pass)."""
    _lib.call('gx_defer_enable', -1)
    del defer_state().keep[:]


# ------------------------------------------------------------------ conv3x3
def conv3x3_fwd(x, w, amax_in=None):
    _chk(x, 'conv3x3_fwd.x'); _chk(w, 'conv3x3_fwd.w')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert w.shape == (Cout, Cin, 3, 3), (w.shape, x.shape)
    y = torch.empty(N, Cout, H, W, dtype=F32, device=x.device)
    nb = _lib.query('gx_conv3x3_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    with _input_amax(amax_in):
        _lib.call('gx_conv3x3_fwd', _p(x), _p(w), _p(y), N, Cin, Cout, H, W, _p(ws), nb, _stream())
    return y


def conv3x3_dgrad(dy, w, amax_in=None):
    _chk(dy, 'conv3x3_dgrad.dy'); _chk(w, 'conv3x3_dgrad.w')
    N, Cout, H, W = dy.shape
    Cin = w.shape[1]
    assert w.shape == (Cout, Cin, 3, 3)
    dx = torch.empty(N, Cin, H, W, dtype=F32, device=dy.device)
    nb = _lib.query('gx_conv3x3_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, dy.device)
    with _input_amax(amax_in):
        _lib.call('gx_conv3x3_dgrad', _p(dy), _p(w), _p(dx), N, Cin, Cout, H, W, _p(ws), nb, _stream())
    return dx


def conv3x3_dgrad_parts(dy, w, amax_in=None):
    """conv3x3_dgrad without the split-K reduce launch: a tensor, or (the small layers, whose channel reduction is split to
    fill the chip) a Parts object for gn_relu_bwd's gradient sources (gx_conv3x3_dgrad_parts)."""
    if not FUSE_SPLITK_INTO_GN:
        return conv3x3_dgrad(dy, w, amax_in)
    _chk(dy, 'conv3x3_dgrad.dy'); _chk(w, 'conv3x3_dgrad.w')
    N, Cout, H, W = dy.shape
    Cin = w.shape[1]
    assert w.shape == (Cout, Cin, 3, 3)
    dx = torch.empty(N, Cin, H, W, dtype=F32, device=dy.device)
    nb = _lib.query('gx_conv3x3_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, dy.device)
    parts, nsplit, stride = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_size_t()
    with _input_amax(amax_in):
        _lib.call('gx_conv3x3_dgrad_parts', _p(dy), _p(w), _p(dx), N, Cin, Cout, H, W, _p(ws), nb, ctypes.byref(parts),
                  ctypes.byref(nsplit), ctypes.byref(stride), _stream())
    if nsplit.value == 1:
        return dx
    return Parts(parts.value, nsplit.value, stride.value, (N, Cin, H, W), (ws, dx))


def conv3x3_wgrad(x, dy, out=None, amax=None):
    """amax = (Amax of dy, Amax of x [or a pair: a concat buffer written by two producers]): see _operand_amax."""
    _chk(x, 'conv3x3_wgrad.x'); _chk(dy, 'conv3x3_wgrad.dy')
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    assert dy.shape == (N, Cout, H, W)
    dw = out if out is not None else torch.empty(Cout, Cin, 3, 3, dtype=F32, device=x.device)
    assert dw.shape == (Cout, Cin, 3, 3) and dw.is_contiguous()
    nb = _lib.query('gx_conv3x3_wgrad_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    with _deferring(out is not None, ws, x, dy), _operand_amax(x.device, amax):    # (a queued job reads x / dy at the flush: kept alive until then)
        _lib.call('gx_conv3x3_wgrad', _p(x), _p(dy), _p(dw), N, Cin, Cout, H, W, _p(ws), nb, _stream())
    return dw


def conv3x3_wgrad_quad_supported(N, C, H, W):
    return bool(_lib.query('gx_conv3x3_wgrad_quad_supported', N, C, H, W))


def conv3x3_wgrad_quad(x, dy, out=None, dbias_out=None):
    """conv3x3 weight gradient of a 32 -> 32 layer, four images per workgroup tile (gx_conv3x3_wgrad_quad); with dbias_out [C]
    also the layer's bias gradient sum_{n,hw} dy from the same read of dy (gx_conv3x3_wgrad_quad_bias)."""
    _chk(x, 'conv3x3_wgrad_quad.x'); _chk(dy, 'conv3x3_wgrad_quad.dy')
    N, C, H, W = x.shape
    assert tuple(dy.shape) == (N, C, H, W)
    dw = out if out is not None else torch.empty(C, C, 3, 3, dtype=F32, device=x.device)
    assert tuple(dw.shape) == (C, C, 3, 3) and dw.is_contiguous()
    if dbias_out is not None:
        _chk(dbias_out, 'conv3x3_wgrad_quad.dbias'); assert dbias_out.numel() == C
        nb = _lib.query('gx_conv3x3_wgrad_quad_bias_ws_bytes', N, C, H, W)
        ws = _ws(nb, x.device)
        _lib.call('gx_conv3x3_wgrad_quad_bias', _p(x), _p(dy), _p(dw), _p(dbias_out), N, C, H, W, _p(ws), nb, _stream())
        return dw
    nb = _lib.query('gx_conv3x3_wgrad_quad_ws_bytes', N, C, H, W)
    ws = _ws(nb, x.device)
    _lib.call('gx_conv3x3_wgrad_quad', _p(x), _p(dy), _p(dw), N, C, H, W, _p(ws), nb, _stream())
    return dw


# ------------------------------------------------------------------ ConvTranspose2d k5 s2 p2 op1
def deconv5x5s2_fwd(x, w, bias):
    _chk(x, 'deconv.x'); _chk(w, 'deconv.w'); _chk(bias, 'deconv.bias')
    N, Cin, H, W = x.shape
    Cout = w.shape[1]
    assert w.shape == (Cin, Cout, 5, 5)
    y = torch.empty(N, Cout, 2 * H, 2 * W, dtype=F32, device=x.device)
    nb = _lib.query('gx_deconv5x5s2_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    _lib.call('gx_deconv5x5s2_fwd', _p(x), _p(w), _p(bias), _p(y), N, Cin, Cout, H, W, _p(ws), nb, _stream())
    return y


AMAX_LINK = os.environ.get('GENESIS_AMAX_LINK', '1') != '0'
AMAX_LINK_REG = os.environ.get('GENESIS_AMAX_LINK_REG', '1') != '0'     # 0: only the decoder head's gradient is handed over


def amax_link(device, numel, capacity=16384):
    """gx_kq_amax_link: arms the one-shot hand-over of a tensor's partial maxima from the kernel that writes it (the decoder head's
    GroupNorm backward) to the fp16 x 3 conv that reads it next (gx_deconv5x5s2_dgrad) -- no second pass over a 235 MB gradient.
    numel: the elements of that tensor (a producer launch that covers only part of it leaves the link alone).
    Returns the scratch buffer (keep it alive until the consumer has been enqueued)."""
    if not AMAX_LINK:
        return None
    buf = torch.empty(capacity, dtype=F32, device=device)
    # the consumer is enqueued by a LATER call: the scratch must not go back to the allocator (and be handed to the tensors of the
    # calls in between) before that -- the last few links' buffers are kept alive here, whatever the caller does with its reference
    keep = getattr(_LAST, 'links', None)
    if keep is None:
        keep = _LAST.links = []          # (per thread, like the link itself)
    keep.append(buf)
    del keep[:-4]
    _lib.call('gx_kq_amax_link', _p(buf), capacity, int(numel))
    return buf


def deconv5x5s2_dgrad(dy, w, cin_out=None):
    _chk(dy, 'deconv_dgrad.dy'); _chk(w, 'deconv_dgrad.w')
    N, Cout, H2, W2 = dy.shape
    Cin = w.shape[0]
    assert w.shape == (Cin, Cout, 5, 5)
    H, W = H2 // 2, W2 // 2
    cin_out = Cin if cin_out is None else cin_out
    dx = torch.empty(N, cin_out, H, W, dtype=F32, device=dy.device)
    nb = _lib.query('gx_deconv5x5s2_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, dy.device)
    _lib.call('gx_deconv5x5s2_dgrad', _p(dy), _p(w), _p(dx), N, Cin, cin_out, Cout, H, W, _p(ws), nb, _stream())
    return dx


def deconv5x5s2_wgrad(x, dy, out=None, amax=None):
    _chk(x, 'deconv_wgrad.x'); _chk(dy, 'deconv_wgrad.dy')
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    assert dy.shape == (N, Cout, 2 * H, 2 * W)
    dw = out if out is not None else torch.empty(Cin, Cout, 5, 5, dtype=F32, device=x.device)
    assert dw.shape == (Cin, Cout, 5, 5) and dw.is_contiguous()
    nb = _lib.query('gx_deconv5x5s2_wgrad_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    with _deferring(out is not None, ws, x, dy), _operand_amax(x.device, amax):
        _lib.call('gx_deconv5x5s2_wgrad', _p(x), _p(dy), _p(dw), N, Cin, Cout, H, W, _p(ws), nb, _stream())
    return dw


# ------------------------------------------------------------------ GroupNorm + ReLU (+ resample / concat placement)
def _view_args(v, name):
    """v = (buffer [N,ctot,Hd,Wd], c0, mode) or None."""
    if v is None:
        return [None, 0, 0, 0]
    buf, c0, mode = v
    _chk(buf, name)
    return [_p(buf), int(buf.shape[1]), int(c0), int(mode)]


class Parts(object):
    """A data gradient that is still `nsplit` split-K partial slabs (conv3x3_dgrad_parts): only gn_relu_bwd reads it (it sums
    the slabs on load, in slab order); keeps the workspace that holds the slabs alive."""
    __slots__ = ('ptr', 'nsplit', 'stride', 'shape', 'keep')

    def __init__(self, ptr, nsplit, stride, shape, keep):
        self.ptr, self.nsplit, self.stride, self.shape, self.keep = ptr, nsplit, stride, shape, keep


def _view_args_p(v, name):
    """v = (tensor | Parts, c0, mode) or None -> (ptr, ctot, c0, mode, nsplit, split stride)."""
    if v is None:
        return [None, 0, 0, 0, 1, 0]
    buf, c0, mode = v
    if isinstance(buf, Parts):
        return [ctypes.c_void_p(buf.ptr), int(buf.shape[1]), int(c0), int(mode), int(buf.nsplit), int(buf.stride)]
    _chk(buf, name)
    return [_p(buf), int(buf.shape[1]), int(c0), int(mode), 1, 0]


# Test diagnostic (tests/test_fullbatch_gpu.py): a list here makes every ReLU-producing forward wrapper append
# (data_ptr of the layer's affine / bias parameter, device count of active outputs, number of outputs) -- the kernels' own
# ReLU pattern per layer, to be compared with the reference's nn.ReLU modules.  None (default): nothing is recorded.
RELU_PROBE = None


def _probe_gn(y, gamma, beta, mean, rstd, groups):
    if RELU_PROBE is not None:
        cnt = torch.zeros(1, dtype=torch.int64, device=y.device)
        N, C, H, W = y.shape
        _lib.call('gx_gn_relu_active_count', _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), N, C, H, W, groups, _p(cnt), _stream())
        RELU_PROBE.append((gamma.data_ptr(), cnt, y.numel()))


def _probe_act(y, bias, act):
    if RELU_PROBE is not None and act == 'relu' and bias is not None:
        RELU_PROBE.append((bias.data_ptr(), torch.count_nonzero(y > 0).reshape(1), y.numel()))


def gn_relu_fwd(y, gamma, beta, groups, eps, dst0, dst1=None):
    """dst0 None: group statistics only (the consumer normalises on load, conv1x1_gn_fwd)."""
    _chk(y, 'gn.y'); _chk(gamma, 'gn.gamma'); _chk(beta, 'gn.beta')
    N, C, H, W = y.shape
    mean = torch.empty(N * groups, dtype=F32, device=y.device)
    rstd = torch.empty(N * groups, dtype=F32, device=y.device)
    tap = _tap_begin(y.device, H * W, y.numel()) if dst0 is not None else None
    _lib.call('gx_gn_relu_fwd', _p(y), _p(gamma), _p(beta), N, C, H, W, groups, float(eps),
              *(_view_args(dst0, 'gn.dst0') + _view_args(dst1, 'gn.dst1')), _p(mean), _p(rstd), _stream())
    _tap_end(tap)
    _probe_gn(y, gamma, beta, mean, rstd, groups)
    return mean, rstd


# ------------------------------------------------------------------ conv -> GroupNorm+ReLU without the split-K reduce pass
# GroupNorm sums the split-K slabs itself (in slab order: bit-identical to the stand-alone reduce), 10 fewer launches
# per step.  Measured (B=32, K=7): 5470 vs 5490 img/s when first tried, 5988 vs 5969 with the later norm kernels: on by
# default, GENESIS_FUSE_SPLITK_GN=0 restores the stand-alone reduce.
FUSE_SPLITK_INTO_GN = os.environ.get('GENESIS_FUSE_SPLITK_GN', '1') == '1'


def _conv_gn(kind, x, w, bias, gamma, beta, groups, eps, dst0, dst1, link_out=False, amax_in=None):
    # link_out: arm gx_kq_amax_link for the normalised output (dst0) -- its next reader is an fp16 x 3 conv
    if not FUSE_SPLITK_INTO_GN:
        y = conv3x3_fwd(x, w, amax_in) if kind == 'conv3x3' else deconv5x5s2_fwd(x, w, bias)
        mean, rstd = gn_relu_fwd(y, gamma, beta, groups, eps, dst0, dst1)
        return y, mean, rstd
    N, Cin, H, W = x.shape
    if kind == 'conv3x3':
        Cout, Ho, Wo = w.shape[0], H, W
        nb = _lib.query('gx_conv3x3_ws_bytes', N, Cin, Cout, H, W)
    else:
        Cout, Ho, Wo = w.shape[1], 2 * H, 2 * W
        nb = _lib.query('gx_deconv5x5s2_ws_bytes', N, Cin, Cout, H, W)
    y = torch.empty(N, Cout, Ho, Wo, dtype=F32, device=x.device)
    ws = _ws(nb, x.device)
    parts, nsplit, stride = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_size_t()
    with _input_amax(amax_in if kind == 'conv3x3' else None):
        _lib.call('gx_conv3x3_fwd_parts' if kind == 'conv3x3' else 'gx_deconv5x5s2_fwd_parts', _p(x), _p(w), _p(y), N, Cin,
                  Cout, H, W, _p(ws), nb, ctypes.byref(parts), ctypes.byref(nsplit), ctypes.byref(stride), _stream())
    mean = torch.empty(N * groups, dtype=F32, device=x.device)
    rstd = torch.empty(N * groups, dtype=F32, device=x.device)
    need_sum = nsplit.value > 1 or bias is not None
    link = amax_link(x.device, N * Cout * Ho * Wo) if link_out and AMAX_LINK_REG else None      # noqa: F841  (alive until the launch below is enqueued)
    tap = _tap_begin(x.device, Ho * Wo, N * Cout * Ho * Wo)
    _lib.call('gx_gn_relu_fwd_parts', parts, nsplit.value, stride.value, _p(bias), _p(y) if need_sum else None,
              _p(gamma), _p(beta), N, Cout, Ho, Wo, groups, float(eps),
              *(_view_args(dst0, 'gn.dst0') + _view_args(dst1, 'gn.dst1')), _p(mean), _p(rstd), _stream())
    _tap_end(tap)
    _probe_gn(y, gamma, beta, mean, rstd, groups)
    return y, mean, rstd       # (ws, holding the partial slabs, is released only now)


def conv3x3_gn_relu_fwd(x, w, gamma, beta, groups, eps, dst0, dst1=None, amax_in=None):
    """conv3x3 (no bias) -> GroupNorm+ReLU into the destination views (modules/blocks.py:159-165); returns the
    pre-norm conv output y (saved for backward) and the group statistics."""
    _chk(x, 'conv_gn.x'); _chk(w, 'conv_gn.w'); _chk(gamma, 'conv_gn.gamma'); _chk(beta, 'conv_gn.beta')
    assert w.shape[1:] == (x.shape[1], 3, 3), (w.shape, x.shape)
    return _conv_gn('conv3x3', x, w, None, gamma, beta, groups, eps, dst0, dst1, amax_in=amax_in)


def deconv5x5s2_gn_relu_fwd(x, w, bias, gamma, beta, groups, eps, dst0, dst1=None, link_out=False):
    """ConvTranspose2d(k5,s2,p2,op1) + bias -> GroupNorm+ReLU (models/genesisv2_config.py:90-98)."""
    _chk(x, 'deconv_gn.x'); _chk(w, 'deconv_gn.w'); _chk(bias, 'deconv_gn.bias')
    _chk(gamma, 'deconv_gn.gamma'); _chk(beta, 'deconv_gn.beta')
    assert w.shape[0] == x.shape[1] and w.shape[2:] == (5, 5)
    return _conv_gn('deconv', x, w, bias, gamma, beta, groups, eps, dst0, dst1, link_out)


def deconv5x5s2_gn_stats_fwd(x, w, bias, gamma, beta, groups, eps):
    """ConvTranspose2d(k5,s2,p2,op1) + bias and the GroupNorm statistics of its output (the normalised tensor itself
    is never written: conv1x1_gn_fwd consumes y).  The statistics come out of the conv epilogue when the shape is
    eligible, else from a statistics-only pass over y."""
    _chk(x, 'deconv_stats.x'); _chk(w, 'deconv_stats.w'); _chk(bias, 'deconv_stats.bias')
    N, Cin, H, W = x.shape
    Cout = w.shape[1]
    assert w.shape[0] == Cin and w.shape[2:] == (5, 5)
    y = torch.empty(N, Cout, 2 * H, 2 * W, dtype=F32, device=x.device)
    mean = torch.empty(N * groups, dtype=F32, device=x.device)
    rstd = torch.empty(N * groups, dtype=F32, device=x.device)
    nb = _lib.query('gx_deconv5x5s2_gn_stats_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    fused = ctypes.c_int(0)
    _lib.call('gx_deconv5x5s2_gn_stats_fwd', _p(x), _p(w), _p(bias), _p(y), N, Cin, Cout, H, W, groups, float(eps),
              _p(mean), _p(rstd), ctypes.byref(fused), _p(ws), nb, _stream())
    if not fused.value:
        mean, rstd = gn_relu_fwd(y, gamma, beta, groups, eps, None)
    else:
        _probe_gn(y, gamma, beta, mean, rstd, groups)
    return y, mean, rstd


def gn_relu_bwd(y, gamma, beta, mean, rstd, groups, g0, g1=None, want_dbias=False, out=None):
    """out = (dgamma, dbeta, dbias) preallocated [C] buffers (entries may be None) to write into."""
    _chk(y, 'gn_bwd.y')
    N, C, H, W = y.shape
    dy = torch.empty_like(y)
    o = out or (None, None, None)
    dgamma = o[0] if o[0] is not None else torch.empty(C, dtype=F32, device=y.device)
    dbeta = o[1] if o[1] is not None else torch.empty(C, dtype=F32, device=y.device)
    dbias = (o[2] if o[2] is not None else torch.empty(C, dtype=F32, device=y.device)) if want_dbias else None
    nb = _lib.query('gx_gn_relu_bwd_ws_bytes', N, C)
    ws = _ws(nb, y.device)
    direct = o[0] is not None and o[1] is not None and (not want_dbias or o[2] is not None)
    tap = _tap_begin(y.device, H * W, y.numel())
    with _deferring(direct, ws):
        _lib.call('gx_gn_relu_bwd_parts', _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), N, C, H, W, groups,
                  *(_view_args_p(g0, 'gn_bwd.g0') + _view_args_p(g1, 'gn_bwd.g1')), _p(dy), _p(dgamma), _p(dbeta),
                  _p(dbias), _p(ws), nb, _stream())
    _tap_end(tap)
    return dy, dgamma, dbeta, dbias


# ------------------------------------------------------------------ IC-SBP
def icsbp_fwd(colour, log_sigma, rand_pixel, K, kernel='gaussian', seed_idx=None, min_mass=0.0):
    """min_mass > 0: dynamic_K (modules/attention.py:218-219); then also returns nsteps [B] int32."""
    _chk(colour, 'icsbp.colour'); _chk(rand_pixel, 'icsbp.rand_pixel')
    _chk(log_sigma, 'icsbp.log_sigma', torch.float64)
    _chk(seed_idx, 'icsbp.seed_idx', torch.int64)
    B, C, H, W = colour.shape
    dev = colour.device
    log_m = torch.empty(K, B, 1, H, W, dtype=F32, device=dev)
    log_s = torch.empty(K, B, 1, H, W, dtype=F32, device=dev)
    seeds = torch.empty(max(K - 1, 0), B, C, dtype=F32, device=dev)
    idx = torch.empty(max(K - 1, 0), B, dtype=torch.int64, device=dev)
    if min_mass > 0.0:
        nsteps = torch.empty(B, dtype=torch.int32, device=dev)
        _lib.call('gx_icsbp_fwd_dyn', _p(colour), _p(log_sigma), _p(rand_pixel), _p(seed_idx), B, C, H, W, K,
                  KERNELS[kernel], float(min_mass), _p(log_m), _p(log_s), _p(seeds), _p(idx), _p(nsteps), _stream())
        return log_m, log_s, seeds, idx, nsteps
    _lib.call('gx_icsbp_fwd', _p(colour), _p(log_sigma), _p(rand_pixel), _p(seed_idx), B, C, H, W, K,
              KERNELS[kernel], _p(log_m), _p(log_s), _p(seeds), _p(idx), _stream())
    return log_m, log_s, seeds, idx


def icsbp_bwd(colour, log_sigma, seeds, idx, g_log_m, kernel='gaussian', out_dls=None, nsteps=None):
    _chk(g_log_m, 'icsbp_bwd.g_log_m')
    B, C, H, W = colour.shape
    K = g_log_m.shape[0]
    dcolour = torch.empty_like(colour)
    dls = out_dls if out_dls is not None else torch.empty((), dtype=torch.float64, device=colour.device)
    _chk(dls, 'icsbp_bwd.dls', torch.float64)
    nb = _lib.query('gx_icsbp_bwd_ws_bytes', B, H, W, K)
    ws = _ws(nb, colour.device)
    if nsteps is not None:
        _chk(nsteps, 'icsbp_bwd.nsteps', torch.int32)
        _lib.call('gx_icsbp_bwd_dyn', _p(colour), _p(log_sigma), _p(seeds), _p(idx), _p(g_log_m), _p(nsteps), B, C, H, W,
                  K, KERNELS[kernel], _p(dcolour), _p(dls), _p(ws), nb, _stream())
    else:
        _lib.call('gx_icsbp_bwd', _p(colour), _p(log_sigma), _p(seeds), _p(idx), _p(g_log_m), B, C, H, W, K,
                  KERNELS[kernel], _p(dcolour), _p(dls), _p(ws), nb, _stream())
    return dcolour, dls


# ------------------------------------------------------------------ masked pooling
def maskpool_fwd(f, log_m):
    _chk(f, 'maskpool.f'); _chk(log_m, 'maskpool.log_m')
    B, C, H, W = f.shape
    K = log_m.shape[0]
    S = torch.empty(B, K, C, dtype=F32, device=f.device)
    msum = torch.empty(B, K, dtype=F32, device=f.device)
    _lib.call('gx_maskpool_fwd', _p(f), _p(log_m), B, C, H, W, K, _p(S), _p(msum), _stream())
    return S, msum


def maskpool_bwd(f, log_m, gS, gmsum):
    _chk(gS, 'maskpool_bwd.gS'); _chk(gmsum, 'maskpool_bwd.gmsum')
    B, C, H, W = f.shape
    K = log_m.shape[0]
    df = torch.empty_like(f)
    dlog_m = torch.empty_like(log_m)
    _lib.call('gx_maskpool_bwd', _p(f), _p(log_m), _p(gS), _p(gmsum), B, C, H, W, K, _p(df), _p(dlog_m), _stream())
    return df, dlog_m


# ------------------------------------------------------------------ mixture likelihood
def mixture_fwd(x, dec, K, pixel_std, pixel_bound=True):
    _chk(x, 'mixture.x'); _chk(dec, 'mixture.dec')
    B, _, H, W = x.shape
    assert dec.shape == (K * B, 4, H, W), dec.shape
    dev = x.device
    recon = torch.empty(B, 3, H, W, dtype=F32, device=dev)
    x_r = torch.empty(K, B, 3, H, W, dtype=F32, device=dev)
    log_m_r = torch.empty(K, B, 1, H, W, dtype=F32, device=dev)
    err = torch.empty(B, dtype=F32, device=dev)
    nb = _lib.query('gx_mixture_ws_bytes', B, H, W)
    ws = _ws(nb, dev)
    _lib.call('gx_mixture_fwd', _p(x), _p(dec), B, H, W, K, float(pixel_std), int(bool(pixel_bound)), _p(recon),
              _p(x_r), _p(log_m_r), _p(err), _p(ws), nb, _stream())
    return err, recon, x_r, log_m_r


def mixture_bwd(x, dec, g_err, K, pixel_std, pixel_bound=True):
    _chk(g_err, 'mixture_bwd.g_err')
    B, _, H, W = x.shape
    ddec = torch.empty_like(dec)
    _lib.call('gx_mixture_bwd', _p(x), _p(dec), _p(g_err), B, H, W, K, float(pixel_std), int(bool(pixel_bound)),
              _p(ddec), _stream())
    return ddec


# ------------------------------------------------------------------ small 1x1 conv
def conv1x1_fwd(x, w, bias, gate=None, addend=None):
    _chk(x, 'conv1x1.x'); _chk(w, 'conv1x1.w'); _chk(bias, 'conv1x1.bias')
    _chk(gate, 'conv1x1.gate'); _chk(addend, 'conv1x1.addend')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = torch.empty(N, Cout, H, W, dtype=F32, device=x.device)
    _lib.call('gx_conv1x1_fwd', _p(x), _p(w), _p(bias), _p(gate), _p(addend), N, Cin, Cout, H, W, _p(y), _stream())
    return y


def conv1x1_bwd(x, dy, w, bias, gate=None, out=None, accumulate=False):
    """out = (dw, db, dgate) preallocated destinations (entries may be None), e.g. the parameters' .grad buffers;
    accumulate: dw / db are ADDED to them (plain conv only)."""
    assert not accumulate or (out is not None and out[0] is not None and gate is None)
    _chk(dy, 'conv1x1_bwd.dy')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    dev = x.device
    o = out or (None, None, None)
    dx = torch.empty_like(x)
    dw = o[0] if o[0] is not None else torch.empty(Cout, Cin, dtype=F32, device=dev)
    db = (o[1] if o[1] is not None else torch.empty(Cout, dtype=F32, device=dev)) if bias is not None else None
    dgate = (o[2] if o[2] is not None else torch.empty((), dtype=F32, device=dev)) if gate is not None else None
    _chk(dw, 'conv1x1_bwd.dw'); _chk(db, 'conv1x1_bwd.db'); _chk(dgate, 'conv1x1_bwd.dgate')
    assert dw.numel() == Cout * Cin
    nb = _lib.query('gx_conv1x1_bwd_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, dev)
    _lib.call('gx_conv1x1_bwd_ex', _p(x), _p(dy), _p(w), _p(bias), _p(gate), N, Cin, Cout, H, W, _p(dx), _p(dw),
              _p(db), _p(dgate), int(bool(accumulate)), _p(ws), nb, _stream())
    return dx, (dw.view(w.shape) if o[0] is None else dw), db, dgate


# ------------------------------------------------------------------ ComponentVAE / MONet path
ACTS = {None: 0, 'none': 0, 'relu': 1, 'elu': 2}


def conv1x1_bwd_act(x, dy, w, bias, act, out=None, dbx_out=None, want_dbx=True, tap=False):
    """conv1x1_bwd of a conv whose input x is a bias + activation layer's output: returns (dxa, dw, db, dbx) with
    dxa = dx * act'(x) and dbx[c] = sum_{n,hw} dxa, that layer's bias gradient (gx_conv1x1_bwd_act)."""
    _chk(dy, 'conv1x1_bwd_act.dy'); _chk(x, 'conv1x1_bwd_act.x')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    dev = x.device
    o = out or (None, None)
    dxa = torch.empty_like(x)
    dw = o[0] if o[0] is not None else torch.empty(Cout, Cin, dtype=F32, device=dev)
    db = (o[1] if o[1] is not None else torch.empty(Cout, dtype=F32, device=dev)) if bias is not None else None
    dbx = None
    if want_dbx:
        dbx = dbx_out if dbx_out is not None else torch.empty(Cin, dtype=F32, device=dev)
    _chk(dw, 'conv1x1_bwd_act.dw'); _chk(db, 'conv1x1_bwd_act.db'); _chk(dbx, 'conv1x1_bwd_act.dbx')
    assert dw.numel() == Cout * Cin and (dbx is None or dbx.numel() == Cin)
    nb = _lib.query('gx_conv1x1_bwd_act_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, dev)
    t = _tap_begin(dev, 1 << 30, dxa.numel()) if tap else None      # (tap: dxa's partial maxima, take_amax() afterwards)
    _lib.call('gx_conv1x1_bwd_act', _p(x), _p(dy), _p(w), _p(bias), N, Cin, Cout, H, W, ACTS[act], _p(dxa), _p(dw), _p(db),
              _p(dbx), _p(ws), nb, _stream())
    if tap:
        _tap_end(t)
    return dxa, (dw.view(w.shape) if o[0] is None else dw), db, dbx


def broadcast_concat(z, coords):
    """z [N,D], coords [1,2,d,d] -> [N, D+2, d, d] (BroadcastLayer + PixelCoords, modules/blocks.py:104-130)."""
    _chk(z, 'broadcast_concat.z'); _chk(coords, 'broadcast_concat.coords')
    N, D = z.shape
    d = coords.shape[-1]
    assert coords.numel() == 2 * d * d
    out = torch.empty(N, D + 2, d, d, dtype=F32, device=z.device)
    _lib.call('gx_broadcast_concat', _p(z), _p(coords), _p(out), N, D, d, _stream())
    return out


def bcast_conv3x3_fwd(z, w, bias, rowc, colc, act, tap=False):
    """act(conv3x3([z broadcast | row coord | col coord], w) + bias) on the d x d canvas without materialising it."""
    for t, n in ((z, 'z'), (w, 'w'), (bias, 'bias'), (rowc, 'rowc'), (colc, 'colc')):
        _chk(t, 'bcast_conv3x3_fwd.' + n)
    N, L = z.shape
    Co, d = w.shape[0], rowc.numel()
    assert w.shape == (Co, L + 2, 3, 3) and colc.numel() == d
    out = torch.empty(N, Co, d, d, dtype=F32, device=z.device)
    t = _tap_begin(z.device, 1 << 30, out.numel()) if tap else None      # (tap: out's partial maxima, take_amax() afterwards)
    _lib.call('gx_bcast_conv3x3_fwd', _p(z), _p(w), _p(bias), _p(rowc), _p(colc), ACTS[act], _p(out), N, L, Co, d,
              _stream())
    if tap:
        _tap_end(t)
    return out


def bcast_conv3x3_bwd(y, g, z, w, rowc, colc, act, out=(None, None)):
    """-> (dz [N,L], dw [Co,L+2,3,3], db [Co]); out = preallocated (dw, db) buffers (entries may be None)."""
    for t, n in ((y, 'y'), (g, 'g'), (z, 'z'), (w, 'w'), (rowc, 'rowc'), (colc, 'colc')):
        _chk(t, 'bcast_conv3x3_bwd.' + n)
    N, L = z.shape
    Co, d = w.shape[0], rowc.numel()
    assert y.shape == (N, Co, d, d) and g.shape == y.shape
    dz = torch.empty(N, L, dtype=F32, device=z.device)
    dw = out[0] if out[0] is not None else torch.empty_like(w)
    db = out[1] if out[1] is not None else torch.empty(Co, dtype=F32, device=z.device)
    assert dw.is_contiguous() and dw.shape == w.shape
    nb = _lib.query('gx_bcast_conv3x3_bwd_ws_bytes', N, Co)
    ws = _ws(nb, z.device)
    _lib.call('gx_bcast_conv3x3_bwd', _p(y), _p(g), _p(z), _p(w), _p(rowc), _p(colc), ACTS[act], N, L, Co, d, _p(dz),
              _p(dw), _p(db), _p(ws), nb, _stream())
    return dz, dw, db


def sbp_scan_fwd(logits, log_s0=None, last_scope=False):
    """logits [T, ...] -> (log_m [T, ...], log_s [T, ...]): T stick-breaking steps in one launch."""
    _chk(logits, 'sbp_scan_fwd.logits'); _chk(log_s0, 'sbp_scan_fwd.log_s0')
    T = logits.shape[0]
    P = logits[0].numel()
    log_m, log_s = torch.empty_like(logits), torch.empty_like(logits)
    _lib.call('gx_sbp_scan_fwd', _p(logits), _p(log_s0), T, P, int(bool(last_scope)), _p(log_m), _p(log_s), _stream())
    return log_m, log_s


def sbp_scan_bwd(logits, g_log_m, g_log_s, last_scope=False, want_g_s0=False):
    _chk(logits, 'sbp_scan_bwd.logits'); _chk(g_log_m, 'sbp_scan_bwd.g_log_m'); _chk(g_log_s, 'sbp_scan_bwd.g_log_s')
    T = logits.shape[0]
    P = logits[0].numel()
    g_logits = torch.empty_like(logits)
    g_s0 = torch.empty_like(logits[0]) if want_g_s0 else None
    _lib.call('gx_sbp_scan_bwd', _p(logits), _p(g_log_m), _p(g_log_s), T, P, int(bool(last_scope)), _p(g_logits),
              _p(g_s0), _stream())
    return g_logits, g_s0


def categorical_kl_fwd(log_m, log_m_r):
    """log_m, log_m_r [K,B,1,H,W] (slot-major) -> kl [B] (MONet.kl_m_loss)."""
    _chk(log_m, 'categorical_kl.log_m'); _chk(log_m_r, 'categorical_kl.log_m_r')
    K, B = log_m.shape[0], log_m.shape[1]
    HW = log_m[0, 0].numel()
    assert log_m_r.shape == log_m.shape
    kl = torch.empty(B, dtype=F32, device=log_m.device)
    _lib.call('gx_categorical_kl_fwd', _p(log_m), _p(log_m_r), K, B, HW, _p(kl), _stream())
    return kl


def categorical_kl_bwd(log_m, log_m_r, g_kl, want_r=False):
    _chk(g_kl, 'categorical_kl_bwd.g_kl')
    K, B = log_m.shape[0], log_m.shape[1]
    HW = log_m[0, 0].numel()
    g_m = torch.empty_like(log_m)
    g_r = torch.empty_like(log_m_r) if want_r else None
    _lib.call('gx_categorical_kl_bwd', _p(log_m), _p(log_m_r), _p(g_kl), K, B, HW, _p(g_m), _p(g_r), _stream())
    return g_m, g_r


def logsoftmax_k_bwd(log_m_r, g, C):
    """g [K,B,1,H,W] on log_m_r = log_softmax_K(dec[:, C-1]) -> g_dec [K*B, C, H, W]."""
    _chk(log_m_r, 'logsoftmax_k_bwd.log_m_r'); _chk(g, 'logsoftmax_k_bwd.g')
    K, B = log_m_r.shape[0], log_m_r.shape[1]
    H, W = log_m_r.shape[-2:]
    g_dec = torch.empty(K * B, C, H, W, dtype=F32, device=g.device)
    _lib.call('gx_logsoftmax_k_bwd', _p(log_m_r), _p(g), K, B, H * W, C, _p(g_dec), _stream())
    return g_dec


def conv3x3_bias_act_fwd(x, w, bias, act, amax_in=None, tap=False):
    """act(conv3x3 s1 p1 (x, w) + bias) on any HxW grid (W*H % 4 == 0).  amax_in: the input's partial maxima (an Amax from the
    kernel that wrote x); tap: ask the launch for its OUTPUT's partial maxima (take_amax() afterwards; served by the <= 32-channel
    bf16-pipe kernel, which is also the one that would otherwise make a pass over its input)."""
    _chk(x, 'conv3x3_bias_act.x'); _chk(w, 'conv3x3_bias_act.w'); _chk(bias, 'conv3x3_bias_act.bias')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert w.shape == (Cout, Cin, 3, 3)
    y = torch.empty(N, Cout, H, W, dtype=F32, device=x.device)
    nb = _lib.query('gx_conv3x3_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    t = _tap_begin(x.device, 1 << 30, y.numel()) if tap else None
    with _input_amax(amax_in):
        _lib.call('gx_conv3x3_bias_act_fwd', _p(x), _p(w), _p(bias), ACTS[act], _p(y), N, Cin, Cout, H, W, _p(ws), nb,
                  _stream())
    if tap:
        _tap_end(t)
    return y


def bias_act_bwd(out, g, act, want_dbias=True, dbias_out=None):
    """dy = g * act'(out) (derivative expressed through the output), dbias[c] = sum_{n,hw} dy."""
    _chk(out, 'bias_act_bwd.out'); _chk(g, 'bias_act_bwd.g')
    N, C, H, W = out.shape
    dy = torch.empty_like(out)
    dbias = None
    if want_dbias:
        dbias = dbias_out if dbias_out is not None else torch.empty(C, dtype=F32, device=out.device)
    nb = _lib.query('gx_bias_act_bwd_ws_bytes', N, C)
    ws = _ws(nb, out.device)
    _lib.call('gx_bias_act_bwd', _p(out), _p(g), N, C, H, W, ACTS[act], _p(dy), _p(dbias), _p(ws), nb, _stream())
    return dy, dbias


def conv3x3_dgrad_act_supported(N, Cin, Cout, H, W):
    return bool(_lib.query('gx_conv3x3_dgrad_act_supported', N, Cin, Cout, H, W))


def conv3x3_dgrad_act(dy, w, xout, act, dbias_out=None, want_dbias=True, amax_in=None, tap=False):
    """dxa = conv3x3_dgrad(dy, w) * act'(xout), dbias[c] = sum_{n,hw} dxa: conv3x3_dgrad + bias_act_bwd of the layer that
    produced xout, the activation's backward in the conv kernel's epilogue (gx_conv3x3_dgrad_act)."""
    _chk(dy, 'conv3x3_dgrad_act.dy'); _chk(w, 'conv3x3_dgrad_act.w'); _chk(xout, 'conv3x3_dgrad_act.xout')
    N, Cout, H, W = dy.shape
    Cin = w.shape[1]
    assert w.shape == (Cout, Cin, 3, 3) and xout.shape == (N, Cin, H, W)
    dxa = torch.empty(N, Cin, H, W, dtype=F32, device=dy.device)
    dbias = None
    if want_dbias:
        dbias = dbias_out if dbias_out is not None else torch.empty(Cin, dtype=F32, device=dy.device)
    nb = _lib.query('gx_conv3x3_dgrad_act_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, dy.device)
    t = _tap_begin(dy.device, 1 << 30, dxa.numel()) if tap else None      # (amax_in / tap: as conv3x3_bias_act_fwd)
    with _input_amax(amax_in):
        _lib.call('gx_conv3x3_dgrad_act', _p(dy), _p(w), _p(xout), ACTS[act], _p(dxa), _p(dbias), N, Cin, Cout, H, W, _p(ws), nb,
                  _stream())
    if tap:
        _tap_end(t)
    return dxa, dbias


def conv2d_direct_fwd(x, w, bias, act, stride, pad):
    _chk(x, 'conv2d_direct.x'); _chk(w, 'conv2d_direct.w'); _chk(bias, 'conv2d_direct.bias')
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(N, Cout, Ho, Wo, dtype=F32, device=x.device)
    nb = _lib.query('gx_conv2d_direct_fwd_ws_bytes', N, Cin, Cout, H, W, k, stride, pad)
    ws = _ws(nb, x.device) if nb else None          # (under-filled grids split the contraction: partial slabs)
    _lib.call('gx_conv2d_direct_fwd_ws', _p(x), _p(w), _p(bias), ACTS[act], _p(y), N, Cin, Cout, H, W, k, stride, pad,
              _p(ws), nb, _stream())
    _probe_act(y, bias, act)
    return y


def conv2d_direct_dgrad(dy, w, H, W, stride, pad):
    _chk(dy, 'conv2d_direct_dgrad.dy'); _chk(w, 'conv2d_direct_dgrad.w')
    N = dy.shape[0]
    Cout, Cin, k, _ = w.shape
    dx = torch.empty(N, Cin, H, W, dtype=F32, device=dy.device)
    _lib.call('gx_conv2d_direct_dgrad', _p(dy), _p(w), _p(dx), N, Cin, Cout, H, W, k, stride, pad, _stream())
    return dx


def mask_image_stack(mask, x):
    """mask [K,B,1,H,W], x [B,C,H,W] -> [K*B, 1 + C, H, W] = [mask_k | x] slot-major (gx_mask_image_stack)."""
    _chk(mask, 'mask_image_stack.mask'); _chk(x, 'mask_image_stack.x')
    K, B = mask.shape[:2]
    C, H, W = x.shape[1:]
    if tuple(mask.shape) != (K, B, 1, H, W) or x.shape[0] != B:
        raise GenesisHipError('mask_image_stack: mask %s / x %s' % (tuple(mask.shape), tuple(x.shape)))
    out = torch.empty(K * B, 1 + C, H, W, dtype=F32, device=x.device)
    _lib.call('gx_mask_image_stack', _p(mask), _p(x), _p(out), K, B, C, H, W, _stream())
    return out


def conv3x3s2_dgrad_lead(dy, w, H, W, cin_n):
    """The compact gradient [N,cin_n,H,W] of the first cin_n input channels of a conv3x3 stride 2 pad 1 (gx_conv3x3s2_dgrad_small_ex)."""
    _chk(dy, 'conv3x3s2_dgrad.dy'); _chk(w, 'conv3x3s2_dgrad.w')
    N = dy.shape[0]
    Cout, Cin = w.shape[0], w.shape[1]
    dx = torch.empty(N, cin_n, H, W, dtype=F32, device=dy.device)
    _lib.call('gx_conv3x3s2_dgrad_small_ex', _p(dy), _p(w), _p(dx), N, Cin, Cout, H, W, int(cin_n), int(cin_n), _stream())
    return dx


def conv3x3s2_dgrad_small(dy, w, H, W, cin_n=None):
    """dx [N,Cin,H,W] of a conv3x3 stride 2 pad 1 from dy [N,Cout,H/2,W/2]; only the first cin_n channels are computed, the
    others are zero (gx_conv3x3s2_dgrad_small)."""
    _chk(dy, 'conv3x3s2_dgrad.dy'); _chk(w, 'conv3x3s2_dgrad.w')
    N = dy.shape[0]
    Cout, Cin = w.shape[0], w.shape[1]
    cin_n = Cin if cin_n is None else int(cin_n)
    dx = (torch.empty if cin_n == Cin else torch.zeros)(N, Cin, H, W, dtype=F32, device=dy.device)
    _lib.call('gx_conv3x3s2_dgrad_small', _p(dy), _p(w), _p(dx), N, Cin, Cout, H, W, cin_n, _stream())
    return dx


def conv3x3s2_wgrad_small(x, dy, out=None):
    """dw [Cout,Cin,3,3] of a conv3x3 stride 2 pad 1 (even H, W) on the vector ALUs (gx_conv3x3s2_wgrad_small)."""
    _chk(x, 'conv3x3s2_wgrad.x'); _chk(dy, 'conv3x3s2_wgrad.dy')
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    assert tuple(dy.shape) == (N, Cout, H // 2, W // 2), (x.shape, dy.shape)
    dw = out if out is not None else torch.empty(Cout, Cin, 3, 3, dtype=F32, device=x.device)
    nb = _lib.query('gx_conv3x3s2_wgrad_small_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    _lib.call('gx_conv3x3s2_wgrad_small', _p(x), _p(dy), _p(dw), N, Cin, Cout, H, W, _p(ws), nb, _stream())
    return dw


def conv2d_direct_wgrad(x, dy, k, stride, pad, out=None):
    _chk(x, 'conv2d_direct_wgrad.x'); _chk(dy, 'conv2d_direct_wgrad.dy')
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    dw = out if out is not None else torch.empty(Cout, Cin, k, k, dtype=F32, device=x.device)
    nb = _lib.query('gx_conv2d_direct_wgrad_ws_bytes', N, Cin, Cout, H, W, k, stride, pad)
    ws = _ws(nb, x.device)
    _lib.call('gx_conv2d_direct_wgrad', _p(x), _p(dy), _p(dw), N, Cin, Cout, H, W, k, stride, pad, _p(ws), nb,
              _stream())
    return dw


def conv5x5s1_supported(N, K, M, H, W):
    return bool(_lib.query('gx_conv5x5s1_supported', N, K, M, H, W))


def conv5x5s1(inp, w, M, flip):
    """5x5 stride-1 pad-2 conv on the tap-conv MFMA kernel: flip False = cross-correlation with w [M][K][5][5], flip True =
    convolution with w [K][M][5][5] (gx_conv5x5s1)."""
    _chk(inp, 'conv5x5s1.in'); _chk(w, 'conv5x5s1.w')
    N, K, H, W = inp.shape
    assert tuple(w.shape) == ((K, M, 5, 5) if flip else (M, K, 5, 5)), (w.shape, K, M, flip)
    out = torch.empty(N, M, H, W, dtype=F32, device=inp.device)
    nb = _lib.query('gx_conv5x5s1_ws_bytes', N, K, M, H, W)
    ws = _ws(nb, inp.device)
    _lib.call('gx_conv5x5s1', _p(inp), _p(w), _p(out), N, K, M, H, W, int(bool(flip)), _p(ws), nb, _stream())
    return out


def conv5x5_wgrad_supported(N, CA, CB, H, W):
    return bool(_lib.query('gx_conv5x5_wgrad_supported', N, CA, CB, H, W))


def conv5x5_wgrad(a, b, out=None, amax=None):
    """dw [CA][CB][5][5] = sum_{n,p} a[n][CA][p] * b[n][CB][p + (kh - 2, kw - 2)] (5x5, stride 1, pad 2): Conv2d with
    (a, b) = (dy, x), stride-1 ConvTranspose2d with (a, b) = (x, dy); bf16-pipe row-ring tiles (gx_conv5x5_wgrad)."""
    _chk(a, 'conv5x5_wgrad.a'); _chk(b, 'conv5x5_wgrad.b')
    N, CA, H, W = a.shape
    CB = b.shape[1]
    assert b.shape == (N, CB, H, W)
    dw = out if out is not None else torch.empty(CA, CB, 5, 5, dtype=F32, device=a.device)
    assert dw.shape == (CA, CB, 5, 5) and dw.is_contiguous()
    nb = _lib.query('gx_conv5x5_wgrad_ws_bytes', N, CA, CB, H, W)
    ws = _ws(nb, a.device)
    with _deferring(out is not None, ws, a, b), _operand_amax(a.device, amax):       # amax = (a's Amax, b's Amax)
        _lib.call('gx_conv5x5_wgrad', _p(a), _p(b), _p(dw), N, CA, CB, H, W, _p(ws), nb, _stream())
    return dw


def mixture_w_fwd(x, dec, log_w, K, std1, std2, pixel_bound=True):
    """Mixture likelihood with external mixing log-weights log_w [K,B,1,H,W] (MONet)."""
    _chk(x, 'mixture_w.x'); _chk(dec, 'mixture_w.dec'); _chk(log_w, 'mixture_w.log_w')
    B, _, H, W = x.shape
    DC = dec.shape[1]
    assert dec.shape == (K * B, DC, H, W) and DC in (3, 4) and log_w.shape == (K, B, 1, H, W)
    dev = x.device
    recon = torch.empty(B, 3, H, W, dtype=F32, device=dev)
    x_r = torch.empty(K, B, 3, H, W, dtype=F32, device=dev)
    err = torch.empty(B, dtype=F32, device=dev)
    nb = _lib.query('gx_mixture_ws_bytes', B, H, W)
    ws = _ws(nb, dev)
    _lib.call('gx_mixture_w_fwd', _p(x), _p(dec), _p(log_w), DC, B, H, W, K, float(std1), float(std2),
              int(bool(pixel_bound)), _p(recon), _p(x_r), _p(err), _p(ws), nb, _stream())
    return err, recon, x_r


def mixture_w_bwd(x, dec, log_w, g_err, K, std1, std2, pixel_bound=True):
    _chk(g_err, 'mixture_w_bwd.g_err')
    B, _, H, W = x.shape
    ddec = torch.empty_like(dec)
    dlog_w = torch.empty_like(log_w)
    _lib.call('gx_mixture_w_bwd', _p(x), _p(dec), _p(log_w), _p(g_err), int(dec.shape[1]), B, H, W, K, float(std1), float(std2),
              int(bool(pixel_bound)), _p(ddec), _p(dlog_w), _stream())
    return ddec, dlog_w


# ------------------------------------------------------------------ sylvester gated unit
NORMS = {None: 0, 'none': 0, 'bn': 1, 'in': 2}


GATED_AMAX = os.environ.get('GENESIS_GATED_AMAX', '1') != '0'       # 0: the convs behind the gated units make their own amax passes


def gated_norm_fwd(y, bias, norm, gh, bh, gg, bg, eps=1e-5):
    """out = norm_h(h + b_h) * sigmoid(norm_g(g + b_g)) for y = [h | g] (third_party/sylvester/layers.py:40-54)."""
    _chk(y, 'gated.y'); _chk(bias, 'gated.bias')
    N, C2, H, W = y.shape
    C = C2 // 2
    out = torch.empty(N, C, H, W, dtype=F32, device=y.device)
    stats = torch.empty(max(_lib.query('gx_gated_stats_floats', NORMS[norm], N, C), 2), dtype=F32, device=y.device)
    # the unit's output feeds a 5 x 5 conv on fp16 pieces (gx_conv5x5s1 / gx_deconv5x5s2_*) and that layer's weight gradient: its
    # partial maxima come out of the apply kernel (link: the next conv; tap: take_amax() for the weight gradient) -- no amax pass
    tap = _tap_begin(y.device, 1 << 30, out.numel()) if GATED_AMAX else None
    link = amax_link(y.device, out.numel()) if GATED_AMAX else None       # noqa: F841
    _lib.call('gx_gated_norm_fwd', _p(y), _p(bias), NORMS[norm], _p(gh), _p(bh), _p(gg), _p(bg), N, C, H, W,
              float(eps), _p(out), _p(stats), _stream())
    _tap_end(tap)
    return out, stats


def philox_noise(uniform_shape, normal_shape, seed, step=None, device='cuda'):
    """(u ~ U[0,1) of uniform_shape, z ~ N(0,1) of normal_shape) as a function of (seed, the int64 device scalar `step`,
    position): one launch, replayable inside a HIP graph (gx_philox_noise).  Either shape may be None."""
    u = torch.empty(uniform_shape, dtype=F32, device=device) if uniform_shape is not None else None
    z = torch.empty(normal_shape, dtype=F32, device=device) if normal_shape is not None else None
    if step is not None:
        assert step.dtype == torch.int64 and step.is_cuda and step.numel() == 1
    _lib.call('gx_philox_noise', _p(u), u.numel() if u is not None else 0, _p(z), z.numel() if z is not None else 0,
              int(seed) & 0xFFFFFFFFFFFFFFFF, step.data_ptr() if step is not None else None, _stream())
    return u, z


def gated_bn_running_arm(C, h_norm, g_norm, momentum=0.1):
    """The next gated_norm_fwd of this thread (norm 'bn') also updates these two BatchNorm2d's running statistics, inside its
    apply kernel (gx_gated_bn_running) -- instead of a bn_running_update launch after it."""
    bufs = (h_norm.running_mean, h_norm.running_var, g_norm.running_mean, g_norm.running_var)
    for b in bufs:
        _chk(b, 'gated_bn_running.buffer')
        assert b.numel() == C
    for n in (h_norm.num_batches_tracked, g_norm.num_batches_tracked):
        assert n.dtype == torch.int64 and n.is_cuda
    _lib.call('gx_gated_bn_running', *[_p(b) for b in bufs], h_norm.num_batches_tracked.data_ptr(),
              g_norm.num_batches_tracked.data_ptr(), float(momentum))


def bn_running_update(stats, C, m, h_norm, g_norm, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d's running_mean / running_var / num_batches_tracked of a gated unit's two norms from the {mean, rstd}
    pairs of gated_norm_fwd, in one launch (gx_bn_running_update)."""
    bufs = (h_norm.running_mean, h_norm.running_var, g_norm.running_mean, g_norm.running_var)
    for b in bufs:
        _chk(b, 'bn_running_update.buffer')
        assert b.numel() == C
    for n in (h_norm.num_batches_tracked, g_norm.num_batches_tracked):
        assert n.dtype == torch.int64 and n.is_cuda
    _lib.call('gx_bn_running_update', _p(stats), C, float(m), float(eps), float(momentum), *[_p(b) for b in bufs],
              h_norm.num_batches_tracked.data_ptr(), g_norm.num_batches_tracked.data_ptr(), _stream())


def gated_norm_bwd(y, bias, norm, gh, bh, gg, bg, stats, dout, out=None):
    """out: optional destinations (dgh, dbh, dgg, dbg, dbias) -- parameter gradient buffers written in place of fresh
    tensors (any of them may be None)."""
    _chk(dout, 'gated_bwd.dout')
    N, C2, H, W = y.shape
    C = C2 // 2
    dev = y.device
    dy = torch.empty_like(y)
    has = NORMS[norm] != 0
    o = out if out is not None else (None,) * 5
    mk = lambda dst, n, want: (dst if dst is not None else torch.empty(n, dtype=F32, device=dev)) if want else None  # noqa: E731
    dgh, dbh, dgg, dbg = mk(o[0], C, has), mk(o[1], C, has), mk(o[2], C, has), mk(o[3], C, has)
    dbias = mk(o[4], C2, bias is not None)
    for t in (dgh, dbh, dgg, dbg, dbias):
        _chk(t, 'gated_bwd.out')
    nb = _lib.query('gx_gated_norm_bwd_ws_bytes', NORMS[norm], N, C)
    ws = _ws(nb, dev)
    tap = _tap_begin(dev, 1 << 30, dy.numel()) if GATED_AMAX else None       # (dy feeds the conv's data gradient and its weight gradient)
    link = amax_link(dev, dy.numel()) if GATED_AMAX else None       # noqa: F841
    _lib.call('gx_gated_norm_bwd', _p(y), _p(bias), NORMS[norm], _p(gh), _p(bh), _p(gg), _p(bg), _p(stats), _p(dout),
              N, C, H, W, _p(dy), _p(dgh), _p(dbh), _p(dgg), _p(dbg), _p(dbias), _p(ws), nb, _stream())
    _tap_end(tap)
    return dy, dgh, dbh, dgg, dbg, dbias


def gated_bn_sync_fwd(y, bias, gh, bh, gg, bg, all_reduce_sum, world, eps=1e-5):
    """The gated unit with BatchNorm statistics over the batches of ALL ranks (cross-replica BatchNorm):
    local fp64 {sum, sum of squares} per channel -> `all_reduce_sum(tensor)` (in place, over the `world` ranks, which hold equal
    shards) -> {mean, rstd} from the global sums and the global count, the unit applied.  Enqueue-only (no host read).
    Returns (out, stats, m_global)."""
    _chk(y, 'gated.y'); _chk(bias, 'gated.bias')
    N, C2, H, W = y.shape
    C = C2 // 2
    dev = y.device
    pack = torch.empty(4 * C, dtype=torch.float64, device=dev)           # [2C][2] sums
    nb = _lib.query('gx_gated_bn_sums_ws_bytes', N, C)
    ws = _ws(nb, dev)
    _lib.call('gx_gated_bn_local_sums', _p(y), _p(bias), N, C, H, W, _p(pack), _p(ws), nb, _stream())
    all_reduce_sum(pack)
    m = float(N * H * W) * world
    out = torch.empty(N, C, H, W, dtype=F32, device=dev)
    stats = torch.empty(4 * C, dtype=F32, device=dev)
    _lib.call('gx_gated_bn_apply', _p(y), _p(bias), _p(pack), m, _p(gh), _p(bh), _p(gg), _p(bg), N, C, H, W, float(eps),
              _p(out), _p(stats), _stream())
    return out, stats, m


def gated_bn_sync_bwd(y, bias, gh, bh, gg, bg, stats, dout, m, all_reduce_sum, out=None):
    """Backward of gated_bn_sync_fwd: the affine / bias gradients from the rank's OWN {S1, S2} (the step's gradient all-reduce
    adds the ranks), dy from the sums added over the ranks and the global count m."""
    _chk(dout, 'gated_bwd.dout')
    N, C2, H, W = y.shape
    C = C2 // 2
    dev = y.device
    o = out if out is not None else (None,) * 5
    mk = lambda dst, n, want: (dst if dst is not None else torch.empty(n, dtype=F32, device=dev)) if want else None  # noqa: E731
    dgh, dbh, dgg, dbg = mk(o[0], C, True), mk(o[1], C, True), mk(o[2], C, True), mk(o[3], C, True)
    dbias = mk(o[4], C2, bias is not None)
    sums = torch.empty(4 * C, dtype=F32, device=dev)
    nb = _lib.query('gx_gated_bn_sums_ws_bytes', N, C)
    ws = _ws(nb, dev)
    _lib.call('gx_gated_bn_bwd_local_sums', _p(y), _p(bias), _p(gh), _p(bh), _p(gg), _p(bg), _p(stats), _p(dout), N, C, H, W,
              _p(sums), _p(dgh), _p(dbh), _p(dgg), _p(dbg), _p(dbias), _p(ws), nb, _stream())
    all_reduce_sum(sums)
    dy = torch.empty_like(y)
    _lib.call('gx_gated_bn_bwd_apply', _p(y), _p(bias), _p(gh), _p(bh), _p(gg), _p(bg), _p(stats), _p(dout), _p(sums), float(m),
              N, C, H, W, _p(dy), _stream())
    return dy, dgh, dbh, dgg, dbg, dbias


# ---------------------------------------------------------------------------------------------- slot latents
def latent_posterior_fwd(zh, eps):
    """zh [B,K,2D] (z_head output), eps [K,B,D] -> z, mu, sigma [K,B,D], log_q [K,B]
    (models/genesisv2_config.py:154-160, models/genesis_config.py:329)."""
    _chk(zh, 'latent.zh'); _chk(eps, 'latent.eps')
    B, K, D2 = zh.shape
    D = D2 // 2
    if tuple(eps.shape) != (K, B, D):
        raise GenesisHipError('latent_posterior_fwd: eps must be [K,B,D] = %s, got %s' % ((K, B, D), tuple(eps.shape)))
    z = torch.empty(K, B, D, dtype=F32, device=zh.device)
    mu, sigma = torch.empty_like(z), torch.empty_like(z)
    log_q = torch.empty(K, B, dtype=F32, device=zh.device)
    _lib.call('gx_latent_posterior_fwd', _p(zh), _p(eps), B, K, D, _p(z), _p(mu), _p(sigma), _p(log_q), _stream())
    return z, mu, sigma, log_q


def latent_posterior_bwd(zh, eps, gz, gmu, gsigma, glogq):
    for t, n in ((gz, 'gz'), (gmu, 'gmu'), (gsigma, 'gsigma'), (glogq, 'glogq')):
        _chk(t, 'latent_bwd.' + n)
    B, K, D2 = zh.shape
    dzh = torch.empty_like(zh)
    _lib.call('gx_latent_posterior_bwd', _p(zh), _p(eps), _p(gz), _p(gmu), _p(gsigma), _p(glogq), B, K, D2 // 2,
              _p(dzh), _stream())
    return dzh


def latent_posterior_step_fwd(zh, eps, z, mu, sigma, log_q, z2=None):
    """One slot of a recurrent posterior into preallocated rows: zh [B,2D] (mean | pre-sigma), eps [B,D] -> z, mu, sigma [B,D],
    log_q [B] (contiguous views of the per-sequence buffers); z2: a row-strided [B,D] view that receives a second copy of z
    (the z columns of the next LSTM step's input rows: LatentSBP's torch.cat((h, z)), modules/attention.py:104)."""
    for t, n in ((zh, 'zh'), (eps, 'eps'), (z, 'z'), (mu, 'mu'), (sigma, 'sigma'), (log_q, 'log_q')):
        _chk(t, 'latent_step.' + n)
    B, D2 = zh.shape
    ld2 = _rows(z2, 'latent_step.z2') if z2 is not None else 0
    _lib.call('gx_latent_posterior_fwd_ex', _p(zh), _p(eps), B, 1, D2 // 2, _p(z), _p(mu), _p(sigma), _p(log_q), _p(z2), ld2,
              _stream())


def latent_posterior_step_bwd(zh, eps, gz, gmu, gsigma, glogq, gz2, dzh):
    """Backward of one slot: gz / gmu / gsigma [B,D], glogq [B] (each may be None), gz2: a row-strided [B,D] view ADDED to gz
    (the LSTM input gradient's z columns) -> dzh [B,2D] (preallocated)."""
    for t, n in ((zh, 'zh'), (eps, 'eps'), (gz, 'gz'), (gmu, 'gmu'), (gsigma, 'gsigma'), (glogq, 'glogq'), (dzh, 'dzh')):
        _chk(t, 'latent_step_bwd.' + n)
    B, D2 = zh.shape
    ld2 = _rows(gz2, 'latent_step_bwd.gz2') if gz2 is not None else 0
    _lib.call('gx_latent_posterior_bwd_ex', _p(zh), _p(eps), _p(gz), _p(gmu), _p(gsigma), _p(glogq), _p(gz2), ld2, B, 1,
              D2 // 2, _p(dzh), _stream())


def latent_prior_logp_fwd(z, lin, log_q=None, all_slots=False):
    """z [K,B,D], lin [K-1,B,2D] or None -> log_p [K,B] (models/genesis_config.py:297-330); with log_q [K,B] the
    per-slot KL sample log_q - log_p (:329-331).  all_slots: lin [K,B,2D], every slot has a conditional prior."""
    _chk(z, 'prior.z'); _chk(lin, 'prior.lin'); _chk(log_q, 'prior.log_q')
    K, B, D = z.shape
    if lin is not None and tuple(lin.shape) != (K if all_slots else K - 1, B, 2 * D):
        raise GenesisHipError('latent_prior_logp_fwd: lin must be [%s,B,2D]' % ('K' if all_slots else 'K-1'))
    out = torch.empty(K, B, dtype=F32, device=z.device)
    _lib.call('gx_latent_prior_logp_fwd_ex', _p(z), _p(lin), _p(log_q), B, K, D, int(bool(all_slots)), _p(out), _stream())
    return out


def latent_prior_sample(lin, eps, tanh_mu=True):
    """lin [B,2D] (prior_linear output), eps [B,D] -> z [B,D] ~ N(tanh(lin[:D]), to_prior_sigma(lin[D:]));
    tanh_mu=False keeps the raw mean (Genesis.sample's mask rollout, models/genesis_config.py:358)."""
    _chk(lin, 'prior_sample.lin'); _chk(eps, 'prior_sample.eps')
    B, D = eps.shape
    assert lin.shape == (B, 2 * D)
    z = torch.empty(B, D, dtype=F32, device=eps.device)
    _lib.call('gx_latent_prior_sample_ex', _p(lin), _p(eps), B, D, int(bool(tanh_mu)), _p(z), _stream())
    return z


def latent_prior_logp_bwd(z, lin, g_out, kl_mode=False, all_slots=False):
    _chk(g_out, 'prior_bwd.g_out')
    K, B, D = z.shape
    dz = torch.empty_like(z)
    dlin = torch.empty_like(lin) if lin is not None else None
    _lib.call('gx_latent_prior_logp_bwd_ex', _p(z), _p(lin), _p(g_out), int(kl_mode), B, K, D, int(bool(all_slots)), _p(dz),
              _p(dlin), _stream())
    return dz, dlin


def elbo_fwd(err, kl, beta, tail=None):
    """err [B], kl [R,B] or None, beta: 1-element device tensor -> out[5] = (loss, elbo, err_mean, kl_mean, beta)
    (train.py:226-242).  tail: 2-element slice of the gradient bucket receiving (err_mean, kl_mean)."""
    _chk(err, 'elbo.err'); _chk(kl, 'elbo.kl'); _chk(beta, 'elbo.beta'); _chk(tail, 'elbo.tail')
    B = err.numel()
    R = 0 if kl is None else kl.numel() // B
    out = torch.empty(5, dtype=F32, device=err.device)
    loss = torch.empty(1, dtype=F32, device=err.device)
    _lib.call('gx_elbo_fwd', _p(err), _p(kl), _p(beta), B, R, _p(out), _p(tail), _p(loss), _stream())
    return loss, out


def elbo_fwd_grads(err, kl, beta, tail=None):
    """elbo_fwd plus (d_err [B], d_kl [R,B] | None) for a unit upstream gradient, one launch (gx_elbo_fwd_grads)."""
    _chk(err, 'elbo.err'); _chk(kl, 'elbo.kl'); _chk(beta, 'elbo.beta'); _chk(tail, 'elbo.tail')
    B = err.numel()
    R = 0 if kl is None else kl.numel() // B
    out = torch.empty(5, dtype=F32, device=err.device)
    d_err = torch.empty_like(err)
    d_kl = torch.empty_like(kl) if kl is not None else None
    _lib.call('gx_elbo_fwd_grads', _p(err), _p(kl), _p(beta), B, R, _p(out), _p(tail), None, _p(d_err), _p(d_kl), _stream())
    return out, d_err, d_kl


def elbo_bwd(g_loss, beta, B, R):
    _chk(g_loss, 'elbo_bwd.g')
    d_err = torch.empty(B, dtype=F32, device=g_loss.device)
    d_kl = torch.empty(R, B, dtype=F32, device=g_loss.device) if R else None
    _lib.call('gx_elbo_bwd', _p(g_loss), _p(beta), B, R, _p(d_err), _p(d_kl), _stream())
    return d_err, d_kl


def pooled_head_fwd(lin, msum, fbias, gamma, beta, eps):
    """(lin + msum fbias) / (msum + 1e-5) -> LayerNorm (models/genesisv2_config.py:146-154, :76)."""
    for t, n in ((lin, 'lin'), (msum, 'msum'), (fbias, 'fbias'), (gamma, 'gamma'), (beta, 'beta')):
        _chk(t, 'pooled_head.' + n)
    R, C = lin.shape
    y = torch.empty_like(lin)
    stats = torch.empty(R, 2, dtype=F32, device=lin.device)
    _lib.call('gx_pooled_head_fwd', _p(lin), _p(msum), _p(fbias), _p(gamma), _p(beta), float(eps), R, C, _p(y),
              _p(stats), _stream())
    return y, stats


def pooled_head_bwd(lin, msum, fbias, gamma, stats, g, out=(None, None, None)):
    """Returns dlin [R,C], dmsum [R], dfbias, dgamma, dbeta [C]; out = (dfbias, dgamma, dbeta) destination buffers."""
    _chk(g, 'pooled_head_bwd.g')
    R, C = lin.shape
    dev = lin.device
    dlin = torch.empty_like(lin)
    dmsum = torch.empty(R, dtype=F32, device=dev)
    dfb, dga, dbe = [o if o is not None else torch.empty(C, dtype=F32, device=dev) for o in out]
    nb = _lib.query('gx_pooled_head_bwd_ws_bytes', R, C)
    ws = _ws(nb, dev)
    _lib.call('gx_pooled_head_bwd', _p(lin), _p(msum), _p(fbias), _p(gamma), _p(stats), _p(g), R, C, _p(dlin),
              _p(dmsum), _p(dfb), _p(dga), _p(dbe), _p(ws), ws.numel(), _stream())
    return dlin, dmsum, dfb, dga, dbe


# ---------------------------------------------------------------------------------------------- dense layers
def _rows(t, name):
    """[M, n] tensor whose rows are contiguous (row stride >= n): returns its row stride."""
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1] or t.dtype != F32 or not t.is_cuda:
        raise GenesisHipError('%s: needs a float32 device matrix with contiguous rows, got %s strides %s'
                              % (name, tuple(t.shape), tuple(t.stride())))
    return t.stride(0)


def bcast_deconv_pack(w, b, coords):
    """First decoder layer on a broadcast input as a matrix product: w [D+2,Cout,5,5], b [Cout], coords [1,2,d,d] ->
    (wz [Cout (2d)^2, D], bias [Cout (2d)^2]); out = linear_fwd(z, wz, bias).view(N, Cout, 2d, 2d)."""
    _chk(w, 'bcast_deconv.w'); _chk(b, 'bcast_deconv.b'); _chk(coords, 'bcast_deconv.coords')
    D, Cout, d = w.shape[0] - 2, w.shape[1], coords.shape[-1]
    if tuple(w.shape[2:]) != (5, 5) or coords.numel() != 2 * d * d or D <= 0:
        raise GenesisHipError('bcast_deconv_pack: w %s / coords %s' % (tuple(w.shape), tuple(coords.shape)))
    rows = Cout * 4 * d * d
    wz = torch.empty(rows, D, dtype=F32, device=w.device)
    bias = torch.empty(rows, dtype=F32, device=w.device)
    _lib.call('gx_bcast_deconv5x5s2_pack', _p(w), _p(b), _p(coords), D, Cout, d, _p(wz), _p(bias), _stream())
    return wz, bias


def bcast_deconv_unpack(dwz, dbias, coords, Cout, out_dw=None, out_db=None, want_db=False):
    """Gradients of (wz, bias) of bcast_deconv_pack -> (dw [D+2,Cout,5,5], db [Cout] or None)."""
    _chk(dwz, 'bcast_deconv.dwz'); _chk(dbias, 'bcast_deconv.dbias'); _chk(coords, 'bcast_deconv.coords')
    _chk(out_dw, 'bcast_deconv.out_dw'); _chk(out_db, 'bcast_deconv.out_db')
    D, d = dwz.shape[1], coords.shape[-1]
    dw = out_dw if out_dw is not None else torch.empty(D + 2, Cout, 5, 5, dtype=F32, device=dwz.device)
    db = out_db if out_db is not None else (torch.empty(Cout, dtype=F32, device=dwz.device) if want_db else None)
    _lib.call('gx_bcast_deconv5x5s2_unpack', _p(dwz), _p(dbias), _p(coords), D, Cout, d, _p(dw), _p(db), _stream())
    return dw, db


def linear_fwd(x, w, b=None, act=None, out=None):
    """act(x [M,K] @ w[N,K]^T + b) (nn.Linear (+ReLU)).  x and out may be row-strided views (columns of a larger
    buffer); out: write the result there instead of a fresh [M,N] tensor."""
    _chk(w, 'linear.w'); _chk(b, 'linear.b')
    ldx = _rows(x, 'linear.x')
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise GenesisHipError('linear_fwd: x is [%d,%d] but w is %s' % (M, K, tuple(w.shape)))
    y = torch.empty(M, N, dtype=F32, device=x.device) if out is None else out
    ldy = _rows(y, 'linear.out')
    if tuple(y.shape) != (M, N):
        raise GenesisHipError('linear_fwd: out is %s, expected (%d, %d)' % (tuple(y.shape), M, N))
    _lib.call('gx_linear_fwd_ld', _p(x), ldx, _p(w), _p(b), ACTS[act], _p(y), ldy, M, N, K, _stream())
    _probe_act(y, b, act)
    return y


def linear_bwd(x, w, y, g, act=None, need_dx=True, need_dw=True, need_db=True, out_dw=None, out_db=None,
               out_db2=None, accumulate_dx=None, accumulate_dw=False, out_dx=None):
    """Returns (dx, dw, db); out_dw / out_db: write the parameter gradients into these buffers (out_db2: a second copy
    of db).  x, and (y, g) with one common row stride, may be row-strided views.  accumulate_dx: a [M,K] (row-strided)
    tensor that dx is ADDED to (returned as dx); out_dx: a [M,K] (row-strided) tensor dx is WRITTEN to.  accumulate_dw: dw / db
    are ADDED to out_dw / out_db."""
    assert not accumulate_dw or out_dw is not None
    M, K = x.shape
    N = w.shape[0]
    dev = x.device
    ldx, ldg = _rows(x, 'linear_bwd.x'), _rows(g, 'linear_bwd.g')
    if y is not None and _rows(y, 'linear_bwd.y') != ldg:
        raise GenesisHipError('linear_bwd: y and g must share a row stride')
    if accumulate_dx is not None:
        dx = accumulate_dx
    elif out_dx is not None:
        dx = out_dx
    else:
        dx = torch.empty(M, K, dtype=F32, device=dev) if need_dx else None
    if dx is not None and tuple(dx.shape) != (M, K):
        raise GenesisHipError('linear_bwd: dx is %s, expected (%d, %d)' % (tuple(dx.shape), M, K))
    lddx = _rows(dx, 'linear_bwd.dx') if dx is not None else K
    dw = (out_dw if out_dw is not None else torch.empty(N, K, dtype=F32, device=dev)) if need_dw else None
    db = (out_db if out_db is not None else torch.empty(N, dtype=F32, device=dev)) if (need_db and need_dw) else None
    _lib.call('gx_linear_bwd_ex', _p(x), ldx, _p(w), _p(y), _p(g), ldg, ACTS[act], _p(dx), lddx,
              int(accumulate_dx is not None) | (2 if accumulate_dw else 0), _p(dw), _p(db),
              _p(out_db2 if db is not None else None), M, N, K, _stream())
    return dx, dw, db


def matmul_nn_fwd(x, w):
    """x [M,K] @ w [K,N] (the weight in [in, out] layout: a ConvTranspose2d weight [z, 2c, k, k] flattened)."""
    _chk(x, 'matmul_nn.x'); _chk(w, 'matmul_nn.w')
    M, K = x.shape
    if w.shape[0] != K:
        raise GenesisHipError('matmul_nn_fwd: x is [%d,%d] but w is %s' % (M, K, tuple(w.shape)))
    y = torch.empty(M, w.shape[1], dtype=F32, device=x.device)
    _lib.call('gx_matmul_nn_fwd', _p(x), _p(w), _p(y), M, w.shape[1], K, _stream())
    return y


def matmul_nn_bwd(x, w, g, need_dx=True, out_dw=None, need_dw=True):
    """-> (dx [M,K] = g w^T, dw [K,N] = x^T g); out_dw: write dw there."""
    _chk(x, 'matmul_nn_bwd.x'); _chk(w, 'matmul_nn_bwd.w'); _chk(g, 'matmul_nn_bwd.g')
    M, K = x.shape
    N = w.shape[1]
    dx = torch.empty(M, K, dtype=F32, device=x.device) if need_dx else None
    dw = (out_dw if out_dw is not None else torch.empty(K, N, dtype=F32, device=x.device)) if need_dw else None
    nb = _lib.query('gx_matmul_nn_bwd_ws_bytes', M, N, K) if need_dx else 0
    ws = _ws(nb, x.device) if need_dx else None
    _lib.call('gx_matmul_nn_bwd', _p(x), _p(w), _p(g), _p(dx), _p(dw), M, N, K, _p(ws), nb, _stream())
    return dx, dw


def logsoftmax_k_fwd(dec, K):
    """dec [K*B, C, H, W] slot-major -> log_softmax over the K slots of the last channel, [K,B,1,H,W]."""
    _chk(dec, 'logsoftmax_k_fwd.dec')
    KB, C, H, W = dec.shape
    B = KB // K
    out = torch.empty(K, B, 1, H, W, dtype=F32, device=dec.device)
    _lib.call('gx_logsoftmax_k_fwd', _p(dec), K, B, H * W, C, _p(out), _stream())
    return out


def lstm_step_fwd(gx, h_prev, c_prev, w_hh, b_hh, act, c, h):
    """One LSTM cell step into preallocated act [B,4H], c, h [B,H] (views of the per-sequence buffers)."""
    B, H4 = gx.shape
    for t, n in ((gx, 'gx'), (h_prev, 'h_prev'), (c_prev, 'c_prev'), (w_hh, 'w_hh'), (b_hh, 'b_hh'), (act, 'act'),
                 (c, 'c'), (h, 'h')):
        _chk(t, 'lstm_step_fwd.' + n)
    _lib.call('gx_lstm_step_fwd', _p(gx), _p(h_prev), _p(c_prev), _p(w_hh), _p(b_hh), B, H4 // 4, _p(act), _p(c),
              _p(h), _stream())


def lstm_step_bwd(g_h, dgates_next, w_hh, act, c, c_prev, dc_next, dgates, dc_prev):
    B, H4 = act.shape
    for t, n in ((g_h, 'g_h'), (dgates_next, 'dgates_next'), (act, 'act'), (c, 'c'), (c_prev, 'c_prev'),
                 (dc_next, 'dc_next'), (dgates, 'dgates'), (dc_prev, 'dc_prev')):
        _chk(t, 'lstm_step_bwd.' + n)
    _lib.call('gx_lstm_step_bwd', _p(g_h), _p(dgates_next), _p(w_hh), _p(act), _p(c), _p(c_prev), _p(dc_next), B,
              H4 // 4, _p(dgates), _p(dc_prev), _stream())


# The whole-sequence LSTM launches (gx_lstm_seq_*) are OFF by default: measured on the metric step (T = 6, B = 32, H = 256) they take
# 42 us forward + 65 us backward against 35 + 38 us for the twelve step launches inside the replayed graph -- a grid-wide barrier
# across the eight XCDs (store acknowledgement, atomic at the memory side, poll, loads past the L2) costs about what a launch boundary
# inside a HIP graph costs (finding 46).  GENESIS_LSTM_SEQ=1 switches them on (tests: bit-identical).
LSTM_SEQ = os.environ.get('GENESIS_LSTM_SEQ', '0') == '1'
_LSTM_BAR = {}


def _lstm_bar(device, which):
    """The grid-barrier counters of the whole-sequence LSTM launches: zero once, left zero by every launch; one set per
    (device, stream, direction) -- launches sharing a set must not overlap."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, which)
    t = _LSTM_BAR.get(key)
    if t is None:
        t = _LSTM_BAR[key] = torch.zeros(_lib.query('gx_lstm_seq_ws_bytes') // 4, dtype=torch.int32, device=device)
    return t


def lstm_seq_steps(B, H):
    """Longest sequence gx_lstm_seq_fwd / _bwd take in one launch for these sizes (0: unroll with lstm_step_*)."""
    return lstm_seq_capacity(B, H) if LSTM_SEQ else 0


def lstm_seq_capacity(B, H):
    return _lib.query('gx_lstm_seq_max_steps', B, H)


def lstm_seq_fwd(gx3, w_hh, b_hh, act, c, h):
    """All T steps of a zero-state LSTM in one launch: gx3 [T,B,4H] -> act [T,B,4H], c, h [T,B,H] (bit-identical to T
    lstm_step_fwd calls)."""
    T, B, H4 = gx3.shape
    for t, n in ((gx3, 'gx'), (w_hh, 'w_hh'), (b_hh, 'b_hh'), (act, 'act'), (c, 'c'), (h, 'h')):
        _chk(t, 'lstm_seq_fwd.' + n)
    _lib.call('gx_lstm_seq_fwd', _p(gx3), _p(w_hh), _p(b_hh), T, B, H4 // 4, _p(act), _p(c), _p(h),
              _p(_lstm_bar(gx3.device, 0)), _stream())


def lstm_seq_bwd(g_h, w_hh, act, c, dgates, dc2):
    """Its backward in one launch: g_h [T,B,H] -> dgates [T,B,4H]; dc2 [2,B,H] scratch."""
    T, B, H4 = act.shape
    for t, n in ((g_h, 'g_h'), (w_hh, 'w_hh'), (act, 'act'), (c, 'c'), (dgates, 'dgates'), (dc2, 'dc2')):
        _chk(t, 'lstm_seq_bwd.' + n)
    _lib.call('gx_lstm_seq_bwd', _p(g_h), _p(w_hh), _p(act), _p(c), T, B, H4 // 4, _p(dgates), _p(dc2),
              _p(_lstm_bar(act.device, 1)), _stream())


# ------------------------------------------------------------------ 1x1 conv on a never-materialised GroupNorm+ReLU
def conv1x1_gn_fwd(y_pre, mean, rstd, gamma, beta, groups, w, bias, gate=None, addend=None):
    """gate * conv1x1(relu(gn(y_pre))) + addend: the normalised activation is formed on load."""
    _chk(y_pre, 'conv1x1_gn.y'); _chk(mean, 'conv1x1_gn.mean'); _chk(rstd, 'conv1x1_gn.rstd')
    _chk(gamma, 'conv1x1_gn.gamma'); _chk(beta, 'conv1x1_gn.beta'); _chk(w, 'conv1x1_gn.w'); _chk(bias, 'conv1x1_gn.bias')
    _chk(gate, 'conv1x1_gn.gate'); _chk(addend, 'conv1x1_gn.addend')
    N, Cin, H, W = y_pre.shape
    Cout = w.shape[0]
    out = torch.empty(N, Cout, H, W, dtype=F32, device=y_pre.device)
    _lib.call('gx_conv1x1_gn_fwd', _p(y_pre), _p(mean), _p(rstd), _p(gamma), _p(beta), groups, _p(w), _p(bias),
              _p(gate), _p(addend), N, Cin, Cout, H, W, _p(out), _stream())
    return out


def conv1x1_gn_wgrad(y_pre, mean, rstd, gamma, beta, groups, g_out, w=None, bias=None, gate=None, out=None):
    """(dw [Cout,Cin], db [Cout], dgate) of gate * conv1x1(relu(gn(y_pre))); out = (dw, db, dgate) destinations or
    None.  w, bias are only read for the gate gradient."""
    _chk(y_pre, 'conv1x1_gn_wgrad.y'); _chk(g_out, 'conv1x1_gn_wgrad.g')
    N, Cin, H, W = y_pre.shape
    Cout = g_out.shape[1]
    o = out or (None, None, None)
    dw = o[0] if o[0] is not None else torch.empty(Cout, Cin, dtype=F32, device=y_pre.device)
    db = o[1] if o[1] is not None else torch.empty(Cout, dtype=F32, device=y_pre.device)
    dgate = (o[2] if o[2] is not None else torch.empty((), dtype=F32, device=y_pre.device)) if gate is not None else None
    _chk(dw, 'conv1x1_gn_wgrad.dw'); _chk(db, 'conv1x1_gn_wgrad.db'); _chk(dgate, 'conv1x1_gn_wgrad.dgate')
    assert dw.numel() == Cout * Cin and db.numel() == Cout
    nb = _lib.query('gx_conv1x1_gn_wgrad_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, y_pre.device)
    _lib.call('gx_conv1x1_gn_wgrad', _p(y_pre), _p(mean), _p(rstd), _p(gamma), _p(beta), groups, _p(g_out), _p(w),
              _p(bias), _p(gate), N, Cin, Cout, H, W, _p(dw), _p(db), _p(dgate), _p(ws), nb, _stream())
    return dw, db, dgate


def conv1x1_gn_bwd_fused(y, gamma, beta, mean, rstd, groups, g_out, w, bias, gate=None, want_dbias=False,
                         out_gn=None, out_conv=None):
    """Whole backward of gate * conv1x1(relu(gn(y))) behind ONE pass over y: dy, the norm's (dgamma, dbeta, dbias) and
    the conv's (dw, db, dgate).  out_gn = (dgamma, dbeta, dbias), out_conv = (dw, db, dgate) destinations or None.
    Returns None when the shape is not supported by the fused kernel (use gn_relu_bwd_proj + conv1x1_gn_wgrad)."""
    N, C, H, W = y.shape
    Cout = int(g_out.shape[1])
    if not _lib.query('gx_gn_relu_bwd_proj_fuses_wgrad', C, H, W, groups, Cout):
        return None
    wpart = torch.empty(N, Cout, C, dtype=F32, device=y.device)
    bpart = torch.empty(N, Cout, dtype=F32, device=y.device)
    dy, dgamma, dbeta, dbias = gn_relu_bwd_proj(y, gamma, beta, mean, rstd, groups, g_out, w, want_dbias, out_gn, gate,
                                                parts=(wpart, bpart))
    o = out_conv or (None, None, None)
    dw = o[0] if o[0] is not None else torch.empty(Cout, C, dtype=F32, device=y.device)
    db = o[1] if o[1] is not None else torch.empty(Cout, dtype=F32, device=y.device)
    dgate = (o[2] if o[2] is not None else torch.empty((), dtype=F32, device=y.device)) if gate is not None else None
    _chk(dw, 'conv1x1_gn_bwd.dw'); _chk(db, 'conv1x1_gn_bwd.db'); _chk(dgate, 'conv1x1_gn_bwd.dgate')
    assert dw.numel() == Cout * C and db.numel() == Cout
    nb = _lib.query('gx_conv1x1_gn_wgrad_finish_ws_bytes', C, Cout)
    ws = _ws(nb, y.device)
    _lib.call('gx_conv1x1_gn_wgrad_finish', _p(wpart), _p(bpart), N, C, Cout, _p(w), _p(bias), _p(gate), _p(dw), _p(db),
              _p(dgate), _p(ws), nb, _stream())
    return dy, (dgamma, dbeta, dbias), (dw, db, dgate)


def gn_relu_bwd_proj(y, gamma, beta, mean, rstd, groups, g_out, w, want_dbias=False, out=None, gate=None,
                     parts=(None, None)):
    """gn_relu_bwd whose incoming gradient is the data gradient of a following 1x1 conv (weight w [Cout,C], output
    gradient g_out [N,Cout,H,W]), formed on load."""
    _chk(y, 'gn_bwd_proj.y'); _chk(g_out, 'gn_bwd_proj.g'); _chk(w, 'gn_bwd_proj.w')
    N, C, H, W = y.shape
    dy = torch.empty_like(y)
    o = out or (None, None, None)
    dgamma = o[0] if o[0] is not None else torch.empty(C, dtype=F32, device=y.device)
    dbeta = o[1] if o[1] is not None else torch.empty(C, dtype=F32, device=y.device)
    dbias = (o[2] if o[2] is not None else torch.empty(C, dtype=F32, device=y.device)) if want_dbias else None
    nb = _lib.query('gx_gn_relu_bwd_proj_ws_bytes', N, C, H, W, groups, int(g_out.shape[1]))
    ws = _ws(nb, y.device)
    direct = o[0] is not None and o[1] is not None and (not want_dbias or o[2] is not None)
    tap = _tap_begin(y.device, H * W, y.numel())
    with _deferring(direct, ws):
        _lib.call('gx_gn_relu_bwd_proj', _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), N, C, H, W, groups,
                  _p(g_out), int(g_out.shape[1]), _p(w), _p(gate), _p(dy), _p(dgamma), _p(dbeta), _p(dbias),
                  _p(parts[0]), _p(parts[1]), _p(ws), nb, _stream())
    _tap_end(tap)
    return dy, dgamma, dbeta, dbias


def conv3x3_pair_supported(x, w1, w2):
    N, Cin, H, W = x.shape
    return bool(_lib.query('gx_conv3x3_pair_supported', N, Cin, w1.shape[0], w2.shape[0], H, W))


class PairWs(object):
    """The packed weights of both directions of a layer pair + the form they were packed in (fp16 pieces or bf16 pieces)."""
    __slots__ = ('t', 'f16')

    def __init__(self, t, f16):
        self.t, self.f16 = t, f16


def conv3x3_pair_fwd(x, w1, w2, amax_in=None):
    """(conv3x3(x, w1), conv3x3(x, w2), ws) in one launch; ws = the packed weights of both directions, handed to
    conv3x3_pair_dgrad of the same iteration."""
    _chk(x, 'pair.x'); _chk(w1, 'pair.w1'); _chk(w2, 'pair.w2')
    N, Cin, H, W = x.shape
    Co1, Co2 = w1.shape[0], w2.shape[0]
    y1 = torch.empty(N, Co1, H, W, dtype=F32, device=x.device)
    y2 = torch.empty(N, Co2, H, W, dtype=F32, device=x.device)
    nb = _lib.query('gx_conv3x3_pair_ws_bytes', N, Cin, Co1, Co2, H, W)
    ws = _ws(nb, x.device)
    hint = _input_amax(amax_in)
    with hint:
        _lib.call('gx_conv3x3_pair_fwd', _p(x), _p(w1), _p(w2), _p(y1), _p(y2), N, Cin, Co1, Co2, H, W, _p(ws), nb, _stream())
    return y1, y2, PairWs(ws, hint.on)


def conv3x3_pair_dgrad(dy1, dy2, w1, w2, ws=None, amax_in=None):
    """dgrad(dy1, w1) + dgrad(dy2, w2) in one launch (ws: the forward's packed weights -- a PairWs; None: pack here).
    amax_in: [Amax of dy1, Amax of dy2].  The forward's packing is reused only in the form THIS call runs in."""
    _chk(dy1, 'pair.dy1'); _chk(dy2, 'pair.dy2'); _chk(w1, 'pair.w1'); _chk(w2, 'pair.w2')
    N, Co1, H, W = dy1.shape
    Co2, Cin = dy2.shape[1], w1.shape[1]
    dx = torch.empty(N, Cin, H, W, dtype=F32, device=dy1.device)
    nb = _lib.query('gx_conv3x3_pair_ws_bytes', N, Cin, Co1, Co2, H, W)
    hint = _input_amax(amax_in)
    if isinstance(ws, PairWs):
        ws = ws.t if ws.f16 == hint.on else None
    pack = ws is None
    if pack:
        ws = _ws(nb, dy1.device)
    with hint:
        _lib.call('gx_conv3x3_pair_dgrad', _p(dy1), _p(dy2), _p(w1), _p(w2), _p(dx), N, Cin, Co1, Co2, H, W, int(pack),
                  _p(ws), nb, _stream())
    return dx


def conv3x3_wino(x, w, mode=0, amax_in=None):
    """Winograd F(2x2,3x3) conv3x3: mode 0 forward (x [N,Cin,H,W]), mode 1 data gradient (x = dy [N,Cout,H,W])."""
    _chk(x, 'wino.x'); _chk(w, 'wino.w')
    N, _, H, W = x.shape
    Cout, Cin = w.shape[0], w.shape[1]
    y = torch.empty(N, Cout if mode == 0 else Cin, H, W, dtype=F32, device=x.device)
    nb = _lib.query('gx_conv3x3_wino_ws_bytes', N, Cin, Cout, H, W)
    ws = _ws(nb, x.device)
    with _input_amax(amax_in):
        _lib.call('gx_conv3x3_wino', _p(x), _p(w), _p(y), N, Cin, Cout, H, W, mode, _p(ws), nb, _stream())
    return y

