"""Block-level torch.autograd.Function wrappers: each block's forward AND backward are sequences of
hand-written HIP kernels (genesis_amd/hip_ops.py -> C ABI), with buffers planned by the block
(concat buffers, resampled copies) instead of torch.cat / F.interpolate passes.

Blocks (reference op groups):
  UNetEncoderFn  modules/unet.py:69-90 (+ blocks.py:159-165)        ops 1-3
  ConvGNReLUFn   modules/blocks.py:159-165 (seg_head / feat_head[0])  op 1
  ICSBPFn        modules/blocks.py:167-178 + modules/attention.py:162-226   ops 4-5
  MaskPoolFn     models/genesisv2_config.py:146-152                   op 6
  DecoderFn      models/genesisv2_config.py:89-99 (+ blocks.py:104-130)   op 8
  MixtureFn      models/genesisv2_config.py:212-223, genesis_config.py:273-286   op 9
"""
import os

import torch
import torch.nn.functional as F

from . import _lib
from . import autostep as _auto
from . import hip_ops as hip

GROUPS = 8
EPS = 1e-5

# When True (set by TrainStep around its iteration: the flat gradient bucket is zeroed every step and every
# parameter is used once), weight-gradient kernels write straight into `param.grad` and the Functions return
# None for those parameters -- no AccumulateGrad `grad += new` launch per parameter.
class _StepState(object):
    """Per-training-loop switches and bookkeeping, keyed by the library context (gx_ctx_*) the loop runs in: two
    TrainSteps in one process never see each other's flags, queues or keep-alive lists.  Autograd runs a HIP node's
    backward on its device thread: every Function below records the context of its forward and makes it current again
    at the start of its backward (ctx_bound), so forward and backward of one step always share one state."""
    __slots__ = ('direct_param_grads', 'direct_written', 'async_wgrad', 'side_prior', 'side_stream', 'keep_alive', 'early_flush')

    def __init__(self):
        self.direct_param_grads = False
        self.direct_written = set()    # ids of parameters whose .grad was already written directly this iteration
        self.async_wgrad = False
        self.side_prior = False
        self.side_stream = None
        self.keep_alive = []
        self.early_flush = None        # callable: the decoder's backward is complete (TrainStep, GENESIS_WGQ_EARLY_FLUSH=1)


_STEPS = {}
# GENESIS_DEBUG_SYNC=1: a device synchronisation at the top of every autograd Function's forward and backward -- the step's results
# must not depend on whether the host runs ahead of the device (tools/diag_two_proc.py)
_DEBUG_SYNC = os.environ.get('GENESIS_DEBUG_SYNC') == '1'


def step_state():
    return _STEPS.setdefault(_lib.current_ctx(), _StepState())


def ctx_bound(cls):
    """Class decorator for the autograd Functions: backward runs in the library context of its forward."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._gx_ctx = _lib.current_ctx()
        if _DEBUG_SYNC:
            torch.cuda.synchronize()
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        prev = _lib.current_ctx()
        _lib.make_current(ctx._gx_ctx)
        if _DEBUG_SYNC:
            torch.cuda.synchronize()
        try:
            if _auto._STATE.models and not _auto._STATE.in_pass and ctx._gx_ctx == 0:
                _auto.begin_backward()          # the unchanged train.py loop: this backward pass on the step machinery
            return bwd(ctx, *grads)
        finally:
            if prev != ctx._gx_ctx:
                try:
                    _lib.make_current(prev)
                except Exception:       # the previous context of this (autograd) thread was destroyed meanwhile
                    pass
    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls


def drop_ctx_state(ctx_id):
    """TrainStep.close(): forget the per-context Python state of a library context that is being destroyed (its id is
    recycled by gx_ctx_create; a new loop must not inherit keep-alive tensors, a side stream or flags)."""
    _STEPS.pop(ctx_id, None)
    hip._DEFER.pop(ctx_id, None)


def begin_direct_grads():
    """TrainStep calls this right after zeroing the bucket."""
    step_state().direct_written.clear()


def _gout_acc(p):
    """(destination, accumulate) for kernels that can ADD into the gradient buffer themselves (the dense layers): p.grad and
    False for the parameter's first gradient of the iteration, p.grad and True for every further use (MONet's recurrent UNet
    runs its weights K-1 times); (None, False) when direct writes are off."""
    st = step_state()
    if st.direct_param_grads and isinstance(p, torch.nn.Parameter) and p.grad is not None and p.grad.is_contiguous() \
            and not st.async_wgrad:
        first = id(p) not in st.direct_written
        st.direct_written.add(id(p))
        return p.grad, not first
    return _gout(p), False


def _gout(p, acc=False):
    """Destination for a parameter gradient: p.grad if direct writes are on and it is a usable buffer.  Only the
    FIRST gradient of a parameter in an iteration is written directly (it overwrites the zeroed bucket); a parameter
    that is used again (MONet's recurrent UNet shares its weights over K-1 passes) accumulates through autograd --
    unless the caller's kernel finishes through the deferred reductions (acc=True: conv weight gradients and GroupNorm
    affine gradients, whose queued reduce ADDS into the buffer): then every use goes straight into p.grad, all of them
    in the step's one stream-K launch / batched reduce."""
    st = step_state()
    if st.direct_param_grads and isinstance(p, torch.nn.Parameter) and p.grad is not None and p.grad.is_contiguous():
        if id(p) not in st.direct_written:
            st.direct_written.add(id(p))
            return p.grad
        if acc and hip.defer_state().on and not st.async_wgrad:
            return p.grad
        # a further use of a shared parameter accumulates through autograd on the MAIN stream: if the first-use direct
        # write was forked onto the side stream (async_wgrad), order the accumulation behind it
        if st.async_wgrad and st.side_stream is not None:
            torch.cuda.current_stream().wait_stream(st.side_stream)
    return None


# ---- weight gradients on a side stream -------------------------------------------------------------------
# Within one backward, layer l's weight gradient depends only on (saved input, dy_l) while the chain continues
# through dgrad_l -> norm-backward_{l-1} -> ...: when TrainStep turns this on, direct-write weight-gradient kernels
# are forked onto a second HIP stream (captured as a parallel branch of the step's HIP graph) so that the
# MFMA-bound wgrad overlaps the HBM-bound norm backward / the latency-bound small layers, and joined before the
# optimiser.  Temporaries they read are kept alive until the join (no allocator reuse hazard across streams).
# MEASURED (round 1, B=32 K=7 64x64): 4066 -> 3464 img/s with the fork on -- the 131 KB-LDS wgrad workgroups and
# the 62 KB tap-conv workgroups evict each other from the CUs and both are MFMA-bound -- so TrainStep leaves it OFF.
ASYNC_WGRAD_MAX_PIXELS = int(__import__('os').environ.get('GENESIS_ASYNC_WGRAD_MAX_PIXELS', 1 << 62))
def _side():
    st = step_state()
    if st.side_stream is None:
        st.side_stream = torch.cuda.Stream()
    return st.side_stream


def _wgrad(fn_, out, *reads):
    """Runs fn_() (a weight-gradient launch writing into `out`) on the side stream when allowed, else inline."""
    # only layers whose kernels cannot fill the chip are forked (ASYNC_WGRAD_MAX_PIXELS images*H*W of the layer input)
    if step_state().async_wgrad and out is not None and reads and \
            reads[0].shape[0] * reads[0].shape[2] * reads[0].shape[3] <= ASYNC_WGRAD_MAX_PIXELS:
        side = _side()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            r = fn_()
        step_state().keep_alive.extend(reads)
        return r
    return fn_()


def join_side_stream():
    """Called by TrainStep after backward: the optimiser must see every weight gradient."""
    st = step_state()
    if st.side_stream is not None:
        torch.cuda.current_stream().wait_stream(st.side_stream)
    st.keep_alive.clear()


# ---- the AR-prior branch on the side stream ----------------------------------------------------------------
# z -> LSTM -> linear -> log p(z) -> KL is a chain of ~12 launch-latency-bound kernels (32 workgroups each) that
# depends only on the posterior sample and joins the main path again at the loss; forward and backward (autograd runs
# a node's backward on its forward's stream) it can run in the shadow of the decoder's chip-filling kernels.  Inside
# the captured step it becomes a parallel branch of the HIP graph.
# MEASURED (round 1, B=32 K=7 64x64, A/B in one session): 5710 img/s without, 5640 with the fork -- the graph's
# fork/join synchronisation costs more than the ~150 us of tiny kernels it hides -- so TrainStep leaves it OFF
# (GENESIS_SIDE_PRIOR=1 / TrainStep(side_prior=True) turns it on).


class side_branch(object):
    """with side_branch(*tensors_read_inside): ...   -- forks the body onto the side stream; join_branch() before the
    results are consumed on the main stream.  Tensors read inside are kept alive until join_side_stream()."""

    def __init__(self, *reads):
        self.reads = reads

    def __enter__(self):
        side = _side()
        side.wait_stream(torch.cuda.current_stream())
        step_state().keep_alive.extend(self.reads)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def join_branch():
    torch.cuda.current_stream().wait_stream(_side())


def _decoder_backward_done():
    """End of DecoderFn.backward.  With TrainStep's early flush on (GENESIS_WGQ_EARLY_FLUSH=1) the weight-gradient / GroupNorm
    affine jobs queued so far -- the decoder's, i.e. the first half of the bucket to become final -- are finished by a stream-K
    launch of their own, and the loop's callback may start that half's all-reduce while the encoder's backward runs."""
    st = step_state()
    if st.early_flush is not None:
        hip.defer_flush()
        st.early_flush()


def _ret(out, value):
    """What a Function returns for a parameter: None if the kernel already wrote into p.grad."""
    return None if out is not None else value


@ctx_bound
class ConvGNReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta):
        x = x.contiguous()
        out = torch.empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3], device=x.device)
        ctx.am_x = getattr(x, '_gx_amax', None)       # (partial maxima of the input, left by the node that produced it)
        y, mean, rstd = hip.conv3x3_gn_relu_fwd(x, w, gamma, beta, GROUPS, EPS, (out, 0, 0), amax_in=ctx.am_x)
        out._gx_amax = hip.take_amax()
        ctx.save_for_backward(x, y, mean, rstd)
        ctx.params = (w, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, mean, rstd = ctx.saved_tensors
        w, gamma, beta = ctx.params
        ow, og, ob = _gout(w), _gout(gamma), _gout(beta)
        dy, dgamma, dbeta, _ = hip.gn_relu_bwd(y, gamma, beta, mean, rstd, GROUPS, (g.contiguous(), 0, 0),
                                               out=(og, ob, None))
        am = (hip.take_amax(), ctx.am_x)
        dw = _wgrad(lambda: hip.conv3x3_wgrad(x, dy, out=ow, amax=am), ow, x, dy)
        dx = hip.conv3x3_dgrad(dy, w, amax_in=am[0]) if ctx.needs_input_grad[0] else None
        return dx, _ret(ow, dw), _ret(og, dgamma), _ret(ob, dbeta)


@ctx_bound
class UNetEncoderFn(torch.autograd.Function):
    """args: x, nb, norm_groups (8 = GroupNorm(8); 0 = InstanceNorm, i.e. one group per channel), then 3*nb down
    params (w, gamma, beta), 3*nb up params, 6 MLP params."""

    @staticmethod
    def forward(ctx, x, nb, norm_groups, *params):
        def ngroups(C):
            return C if norm_groups == 0 else norm_groups
        ctx.ngroups = ngroups
        x = x.contiguous()
        down = [params[3 * i:3 * i + 3] for i in range(nb)]
        up = [params[3 * nb + 3 * j:3 * nb + 3 * j + 3] for j in range(nb)]
        mlp = params[6 * nb:6 * nb + 6]
        N, _, S, _ = x.shape
        dev = x.device
        # concat buffers for the up path: cat_j = [x_up (Cx_j) | skip from down block nb-1-j]
        cats = []
        for j in range(nb):
            cin = up[j][0].shape[1]
            res = S >> (nb - 1 - j)
            cats.append(torch.empty(N, cin, res, res, device=dev))
        saved_down = []
        cur = x
        mlp_in = None
        # partial maxima of every conv input (hip_ops.Amax; None = unknown): am_in[i] of down block i's input, am_cat[j] =
        # [of the up-sampled part, of the skip part] of concat buffer j -- for the fp16-piece weight gradients
        am_cur = getattr(x, '_gx_amax', None)
        am_in, am_cat = [], [[None, None] for _ in range(nb)]
        for i in range(nb):
            w, gamma, beta = down[i]
            C, Hc, Wc = w.shape[0], cur.shape[2], cur.shape[3]
            j = nb - 1 - i
            cx = cats[j].shape[1] - C
            if i < nb - 1:
                nxt = torch.empty(N, C, Hc // 2, Wc // 2, device=dev)
                y, mean, rstd = hip.conv3x3_gn_relu_fwd(cur, w, gamma, beta, ngroups(C), EPS, (cats[j], cx, 0),
                                                        (nxt, 0, 2), amax_in=am_cur)
            else:
                nxt = torch.empty(N, C, Hc, Wc, device=dev)
                y, mean, rstd = hip.conv3x3_gn_relu_fwd(cur, w, gamma, beta, ngroups(C), EPS, (cats[j], cx, 0),
                                                        (nxt, 0, 0), amax_in=am_cur)
                mlp_in = nxt
            am_in.append(am_cur)
            am_cur = am_cat[j][1] = hip.take_amax()          # (the skip slice and the resampled copy hold the same values)
            saved_down.append((cur, y, mean, rstd))
            cur = nxt
        # bottleneck MLP: three Linear+ReLU on the dense MFMA kernel (modules/unet.py:58-62,83)
        h = mlp_in.view(N, -1)
        mlp_acts = []
        fs = mlp_in.shape[2]
        cx0 = cats[0].shape[1] - mlp_in.shape[1]
        for q in range(3):
            # the last layer writes straight into its channels of the first up-block's concat buffer (row-strided)
            dst = cats[0].view(N, -1)[:, :cx0 * fs * fs] if q == 2 else None
            y_ = hip.linear_fwd(h, mlp[2 * q], mlp[2 * q + 1], 'relu', out=dst)
            mlp_acts.append((h, y_))
            h = y_
        saved_up = []
        out = None
        for j in range(nb):
            w, gamma, beta = up[j]
            if j < nb - 1:
                y, mean, rstd = hip.conv3x3_gn_relu_fwd(cats[j], w, gamma, beta, ngroups(w.shape[0]), EPS,
                                                        (cats[j + 1], 0, 1), amax_in=am_cat[j])
            else:
                out = torch.empty(N, w.shape[0], cats[j].shape[2], cats[j].shape[3], device=dev)
                y, mean, rstd = hip.conv3x3_gn_relu_fwd(cats[j], w, gamma, beta, ngroups(w.shape[0]), EPS,
                                                        (out, 0, 0), amax_in=am_cat[j])
            if j < nb - 1:
                am_cat[j + 1][0] = hip.take_amax()
            else:
                out._gx_amax = hip.take_amax()
            saved_up.append((y, mean, rstd))
        ctx.am_in, ctx.am_cat = am_in, am_cat
        ctx.nb = nb
        ctx.params = params
        ctx.cats = cats
        ctx.saved_down = saved_down
        ctx.saved_up = saved_up
        ctx.mlp = (mlp, mlp_acts, mlp_in.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        nb = ctx.nb
        params = ctx.params
        down = [params[3 * i:3 * i + 3] for i in range(nb)]
        up = [params[3 * nb + 3 * j:3 * nb + 3 * j + 3] for j in range(nb)]
        cats = ctx.cats
        g_down = [None] * nb
        g_up = [None] * nb
        dcat = [None] * nb
        gsrc = (g_out.contiguous(), 0, 0)
        for j in reversed(range(nb)):
            w, gamma, beta = up[j]
            y, mean, rstd = ctx.saved_up[j]
            ow, og, ob = _gout(w, True), _gout(gamma, True), _gout(beta, True)
            dy, dgamma, dbeta, _ = hip.gn_relu_bwd(y, gamma, beta, mean, rstd, ctx.ngroups(y.shape[1]), gsrc,
                                                   out=(og, ob, None))
            am = (hip.take_amax(), ctx.am_cat[j])
            dw = _wgrad(lambda cj=cats[j], dy=dy, ow=ow, am=am: hip.conv3x3_wgrad(cj, dy, out=ow, amax=am), ow, cats[j], dy)
            # (read by GroupNorm backward kernels only -- they sum split-K slabs on load; dcat[0] also feeds the MLP)
            dcat[j] = hip.conv3x3_dgrad_parts(dy, w, amax_in=am[0]) if j > 0 else hip.conv3x3_dgrad(dy, w, amax_in=am[0])
            g_up[j] = (_ret(ow, dw), _ret(og, dgamma), _ret(ob, dbeta))
            gsrc = (dcat[j], 0, 1)   # block j-1's output was 2x up-sampled into cat_j[:, :Cx]
        # MLP backward
        mlp, mlp_acts, mlp_in_shape = ctx.mlp
        cx0 = cats[0].shape[1] - mlp_in_shape[1]
        g_h = dcat[0].view(dcat[0].shape[0], -1)[:, :mlp_acts[2][1].shape[1]]     # row-strided view, like the output
        g_mlp = [None] * 6
        for q in reversed(range(3)):
            w_, b_ = mlp[2 * q], mlp[2 * q + 1]
            h_in, y_ = mlp_acts[q]
            # (a UNet used again in the iteration -- MONet's K-1 passes -- adds inside the dense kernel)
            (ow, acc_w), (ob, acc_b) = _gout_acc(w_), _gout_acc(b_)
            if ow is None or ob is None or acc_w != acc_b:
                ow = ob = None
                acc_w = False
            g_h, dw, db = hip.linear_bwd(h_in, w_, y_, g_h, 'relu', out_dw=ow, out_db=ob, accumulate_dw=acc_w)
            g_mlp[2 * q], g_mlp[2 * q + 1] = _ret(ow, dw), _ret(ob, db)
        d_mlp_in = g_h.view(mlp_in_shape)
        d_next = None
        dx = None
        for i in reversed(range(nb)):
            w, gamma, beta = down[i]
            cur, y, mean, rstd = ctx.saved_down[i]
            j = nb - 1 - i
            C = y.shape[1]
            cx = cats[j].shape[1] - C
            g0 = (dcat[j], cx, 0)
            g1 = (d_mlp_in, 0, 0) if i == nb - 1 else (d_next, 0, 2)
            ow, og, ob = _gout(w, True), _gout(gamma, True), _gout(beta, True)
            dy, dgamma, dbeta, _ = hip.gn_relu_bwd(y, gamma, beta, mean, rstd, ctx.ngroups(C), g0, g1,
                                                   out=(og, ob, None))
            am = (hip.take_amax(), ctx.am_in[i])
            dw = _wgrad(lambda cur=cur, dy=dy, ow=ow, am=am: hip.conv3x3_wgrad(cur, dy, out=ow, amax=am), ow, cur, dy)
            g_down[i] = (_ret(ow, dw), _ret(og, dgamma), _ret(ob, dbeta))
            if i > 0:
                d_next = hip.conv3x3_dgrad_parts(dy, w, amax_in=am[0])
            elif ctx.needs_input_grad[0]:
                dx = hip.conv3x3_dgrad(dy, w, amax_in=am[0])
        flat = []
        for t in g_down + g_up:
            flat.extend(t)
        flat.extend(g_mlp)
        return (dx, None, None) + tuple(flat)


@ctx_bound
class ICSBPFn(torch.autograd.Function):
    """colour head (SemiConv or plain 1x1 conv) + IC-SBP.  Returns
    (log_m [K,B,1,H,W], log_s [K,B,1,H,W], colour [B,8,H,W], seeds [K-1,B,8], seed_idx [K-1,B])."""

    @staticmethod
    def forward(ctx, feat, conv_w, conv_b, gate, uv, log_sigma, rand_pixel, K, kernel, seed_idx, min_mass=0.0):
        """min_mass > 0: dynamic_K; a sixth output nsteps [B] (int32) says how many steps each image ran."""
        feat = feat.contiguous()
        ctx.params = (conv_w, conv_b, gate, log_sigma)
        conv_w = conv_w.detach().view(conv_w.shape[0], -1)
        colour = hip.conv1x1_fwd(feat, conv_w, conv_b, gate, uv)
        # the kernel takes the bandwidth as an fp64 device scalar (it is an fp64 parameter for the default
        # kernel, modules/attention.py:150,155; fp32 only for epanechnikov)
        ctx.ls_dtype = log_sigma.dtype
        log_sigma = log_sigma.detach().to(torch.float64)
        res = hip.icsbp_fwd(colour, log_sigma, rand_pixel.contiguous(), K, kernel, seed_idx, min_mass)
        log_m, log_s, seeds, idx = res[:4]
        ctx.nsteps = res[4] if min_mass > 0.0 else None
        ctx.save_for_backward(feat, conv_w, conv_b, gate, log_sigma, colour, seeds, idx)
        ctx.kernel = kernel
        ctx.mark_non_differentiable(log_s, colour, seeds, idx)
        ctx.set_materialize_grads(False)   # no zero-filled gradients for the outputs nobody differentiates
        if ctx.nsteps is not None:
            ctx.mark_non_differentiable(ctx.nsteps)
            return log_m, log_s, colour, seeds, idx, ctx.nsteps
        return log_m, log_s, colour, seeds, idx

    @staticmethod
    def backward(ctx, g_log_m, *unused):
        feat, conv_w, conv_b, gate, log_sigma, colour, seeds, idx = ctx.saved_tensors
        if g_log_m is None:       # log_m unused downstream
            g_log_m = colour.new_zeros(seeds.shape[0] + 1, colour.shape[0], 1, colour.shape[2], colour.shape[3])
        pw, pb, pg, pls = ctx.params
        ow, ob, og = _gout(pw), _gout(pb), (_gout(pg) if pg is not None else None)
        ols = _gout(pls) if ctx.ls_dtype == torch.float64 and pls.dim() == 0 else None
        dcolour, dls = hip.icsbp_bwd(colour, log_sigma, seeds, idx, g_log_m.contiguous(), ctx.kernel, out_dls=ols,
                                     nsteps=ctx.nsteps)
        dfeat, dw, db, dgate = hip.conv1x1_bwd(feat, dcolour, conv_w, conv_b, gate, out=(ow, ob, og))
        return (dfeat, _ret(ow, dw.view(pw.shape)), _ret(ob, db), _ret(og, dgate), None,
                _ret(ols, dls.to(ctx.ls_dtype)), None, None, None, None, None)


# seg_head's GroupNorm+ReLU output is consumed by the colour head's 1x1 conv only: keep it out of memory (as the
# decoder head; GENESIS_FUSE_SEG_HEAD=0 restores ConvGNReLUFn + ICSBPFn)
FUSE_SEG_HEAD = __import__('os').environ.get('GENESIS_FUSE_SEG_HEAD', '1') == '1'


def seg_head_fusable(enc_feat, seg_w, conv_w):
    C, HW = seg_w.shape[0], enc_feat.shape[2] * enc_feat.shape[3]
    return FUSE_SEG_HEAD and C <= 64 and C % GROUPS == 0 and HW % 256 == 0 and conv_w.shape[0] <= 8


@ctx_bound
class SegICSBPFn(torch.autograd.Function):
    """seg_head (conv3x3 -> GroupNorm -> ReLU, genesisv2_config.py:66) + colour head + IC-SBP with the normalised
    seg_head activation never written: statistics only, the 1x1 conv normalises on load, its data gradient is formed
    inside the norm backward.  Same outputs as ConvGNReLUFn + ICSBPFn."""

    @staticmethod
    def forward(ctx, enc_feat, seg_w, seg_gamma, seg_beta, conv_w, conv_b, gate, uv, log_sigma, rand_pixel, K, kernel,
                seed_idx, min_mass=0.0):
        x = enc_feat.contiguous()
        ctx.params = (seg_w, seg_gamma, seg_beta, conv_w, conv_b, gate, log_sigma)
        w2 = conv_w.detach().view(conv_w.shape[0], -1)
        y, mean, rstd = hip.conv3x3_gn_relu_fwd(x, seg_w, seg_gamma, seg_beta, GROUPS, EPS, None)
        colour = hip.conv1x1_gn_fwd(y, mean, rstd, seg_gamma, seg_beta, GROUPS, w2, conv_b, gate, uv)
        ctx.ls_dtype = log_sigma.dtype
        ls64 = log_sigma.detach().to(torch.float64)
        res = hip.icsbp_fwd(colour, ls64, rand_pixel.contiguous(), K, kernel, seed_idx, min_mass)
        log_m, log_s, seeds, idx = res[:4]
        ctx.nsteps = res[4] if min_mass > 0.0 else None
        ctx.save_for_backward(x, y, mean, rstd, ls64, colour, seeds, idx)
        ctx.kernel = kernel
        ctx.mark_non_differentiable(log_s, colour, seeds, idx)
        ctx.set_materialize_grads(False)
        if ctx.nsteps is not None:
            ctx.mark_non_differentiable(ctx.nsteps)
            return log_m, log_s, colour, seeds, idx, ctx.nsteps
        return log_m, log_s, colour, seeds, idx

    @staticmethod
    def backward(ctx, g_log_m, *unused):
        x, y, mean, rstd, ls64, colour, seeds, idx = ctx.saved_tensors
        seg_w, seg_gamma, seg_beta, conv_w, conv_b, gate, log_sigma = ctx.params
        if g_log_m is None:
            g_log_m = colour.new_zeros(seeds.shape[0] + 1, colour.shape[0], 1, colour.shape[2], colour.shape[3])
        ols = _gout(log_sigma) if ctx.ls_dtype == torch.float64 and log_sigma.dim() == 0 else None
        dcolour, dls = hip.icsbp_bwd(colour, ls64, seeds, idx, g_log_m.contiguous(), ctx.kernel, out_dls=ols,
                                     nsteps=ctx.nsteps)
        w2 = conv_w.detach().view(conv_w.shape[0], -1)
        ow, ob, og = _gout(conv_w), _gout(conv_b), (_gout(gate) if gate is not None else None)
        osw, osg, osb = _gout(seg_w), _gout(seg_gamma), _gout(seg_beta)
        fused = hip.conv1x1_gn_bwd_fused(y, seg_gamma, seg_beta, mean, rstd, GROUPS, dcolour, w2, conv_b, gate, False,
                                         out_gn=(osg, osb, None), out_conv=(ow, ob, og))
        if fused is not None:
            dy, (dgamma, dbeta, _), (dw, db, dgate) = fused
        else:
            dw, db, dgate = hip.conv1x1_gn_wgrad(y, mean, rstd, seg_gamma, seg_beta, GROUPS, dcolour, w2, conv_b, gate,
                                                 out=(ow, ob, og))
            dy, dgamma, dbeta, _ = hip.gn_relu_bwd_proj(y, seg_gamma, seg_beta, mean, rstd, GROUPS, dcolour, w2, False,
                                                        out=(osg, osb, None), gate=gate)
        dsw = _wgrad(lambda: hip.conv3x3_wgrad(x, dy, out=osw), osw, x, dy)
        dx = hip.conv3x3_dgrad(dy, seg_w) if ctx.needs_input_grad[0] else None
        return (dx, _ret(osw, dsw), _ret(osg, dgamma), _ret(osb, dbeta), _ret(ow, dw.view(conv_w.shape)), _ret(ob, db),
                _ret(og, dgate), None, _ret(ols, dls.to(ctx.ls_dtype)), None, None, None, None, None)


PAIR_HEADS = __import__('os').environ.get('GENESIS_PAIR_HEADS', '1') == '1'


def heads_pairable(enc_feat, seg_w, conv_w, feat_w):
    return PAIR_HEADS and seg_head_fusable(enc_feat, seg_w, conv_w) and feat_w.shape[1:] == seg_w.shape[1:] \
        and hip.conv3x3_pair_supported(enc_feat, seg_w, feat_w)


@ctx_bound
class SegFeatHeadsFn(torch.autograd.Function):
    """SegICSBPFn and feat_head[0] (ConvGNReLUFn) as ONE node: both read the encoder features
    (models/genesisv2_config.py:110-149), so their two conv3x3 run as one layer with 2 x 64 output channels, and -- the
    point -- their two input gradients are one data-gradient launch over 128 reduction channels whose accumulators do
    the summing (instead of two launches and an accumulation pass over the 33 MB feature-map gradient).
    Returns SegICSBPFn's outputs followed by f = relu(gn(conv3x3(enc_feat, feat_w)))."""

    @staticmethod
    def forward(ctx, enc_feat, seg_w, seg_gamma, seg_beta, conv_w, conv_b, gate, uv, log_sigma, rand_pixel, K, kernel,
                seed_idx, min_mass, feat_w, feat_gamma, feat_beta):
        x = enc_feat.contiguous()
        ctx.am_x = getattr(enc_feat, '_gx_amax', None)
        ctx.params = (seg_w, seg_gamma, seg_beta, conv_w, conv_b, gate, log_sigma, feat_w, feat_gamma, feat_beta)
        w2 = conv_w.detach().view(conv_w.shape[0], -1)
        y, yf, ctx.pair_ws = hip.conv3x3_pair_fwd(x, seg_w, feat_w, amax_in=ctx.am_x)
        mean, rstd = hip.gn_relu_fwd(y, seg_gamma, seg_beta, GROUPS, EPS, None)          # statistics only
        colour = hip.conv1x1_gn_fwd(y, mean, rstd, seg_gamma, seg_beta, GROUPS, w2, conv_b, gate, uv)
        ctx.ls_dtype = log_sigma.dtype
        ls64 = log_sigma.detach().to(torch.float64)
        res = hip.icsbp_fwd(colour, ls64, rand_pixel.contiguous(), K, kernel, seed_idx, min_mass)
        log_m, log_s, seeds, idx = res[:4]
        ctx.nsteps = res[4] if min_mass > 0.0 else None
        f = torch.empty_like(yf)
        meanf, rstdf = hip.gn_relu_fwd(yf, feat_gamma, feat_beta, GROUPS, EPS, (f, 0, 0))
        ctx.save_for_backward(x, y, mean, rstd, ls64, colour, seeds, idx, yf, meanf, rstdf)
        ctx.kernel = kernel
        ctx.mark_non_differentiable(log_s, colour, seeds, idx)
        ctx.set_materialize_grads(False)
        if ctx.nsteps is not None:
            ctx.mark_non_differentiable(ctx.nsteps)
            return log_m, log_s, colour, seeds, idx, ctx.nsteps, f
        return log_m, log_s, colour, seeds, idx, f

    @staticmethod
    def backward(ctx, g_log_m, *rest):
        x, y, mean, rstd, ls64, colour, seeds, idx, yf, meanf, rstdf = ctx.saved_tensors
        seg_w, seg_gamma, seg_beta, conv_w, conv_b, gate, log_sigma, feat_w, feat_gamma, feat_beta = ctx.params
        g_f = rest[-1]
        if g_log_m is None:
            g_log_m = colour.new_zeros(seeds.shape[0] + 1, colour.shape[0], 1, colour.shape[2], colour.shape[3])
        if g_f is None:
            g_f = torch.zeros_like(yf)
        # --- seg head + colour head + IC-SBP (as SegICSBPFn.backward)
        ols = _gout(log_sigma) if ctx.ls_dtype == torch.float64 and log_sigma.dim() == 0 else None
        dcolour, dls = hip.icsbp_bwd(colour, ls64, seeds, idx, g_log_m.contiguous(), ctx.kernel, out_dls=ols,
                                     nsteps=ctx.nsteps)
        w2 = conv_w.detach().view(conv_w.shape[0], -1)
        ow, ob, og = _gout(conv_w), _gout(conv_b), (_gout(gate) if gate is not None else None)
        osw, osg, osb = _gout(seg_w), _gout(seg_gamma), _gout(seg_beta)
        fused = hip.conv1x1_gn_bwd_fused(y, seg_gamma, seg_beta, mean, rstd, GROUPS, dcolour, w2, conv_b, gate, False,
                                         out_gn=(osg, osb, None), out_conv=(ow, ob, og))
        if fused is not None:
            dy, (dgamma, dbeta, _), (dw, db, dgate) = fused
        else:
            dw, db, dgate = hip.conv1x1_gn_wgrad(y, mean, rstd, seg_gamma, seg_beta, GROUPS, dcolour, w2, conv_b, gate,
                                                 out=(ow, ob, og))
            dy, dgamma, dbeta, _ = hip.gn_relu_bwd_proj(y, seg_gamma, seg_beta, mean, rstd, GROUPS, dcolour, w2, False,
                                                        out=(osg, osb, None), gate=gate)
        am_s = (hip.take_amax(), ctx.am_x)        # (conv1x1_gn_bwd_fused ends in gn_relu_bwd_proj: the tap is its dy's)
        dsw = _wgrad(lambda: hip.conv3x3_wgrad(x, dy, out=osw, amax=am_s), osw, x, dy)
        # --- feat_head[0] (as ConvGNReLUFn.backward)
        ofw, ofg, ofb = _gout(feat_w), _gout(feat_gamma), _gout(feat_beta)
        dyf, dfgamma, dfbeta, _ = hip.gn_relu_bwd(yf, feat_gamma, feat_beta, meanf, rstdf, GROUPS, (g_f.contiguous(), 0, 0),
                                                  out=(ofg, ofb, None))
        am_f = (hip.take_amax(), ctx.am_x)
        dfw = _wgrad(lambda: hip.conv3x3_wgrad(x, dyf, out=ofw, amax=am_f), ofw, x, dyf)
        # --- both input gradients: one launch
        dx = hip.conv3x3_pair_dgrad(dy, dyf, seg_w, feat_w, ctx.pair_ws, amax_in=[am_s[0], am_f[0]]) if ctx.needs_input_grad[0] else None
        return (dx, _ret(osw, dsw), _ret(osg, dgamma), _ret(osb, dbeta), _ret(ow, dw.view(conv_w.shape)), _ret(ob, db),
                _ret(og, dgate), None, _ret(ols, dls.to(ctx.ls_dtype)), None, None, None, None, None,
                _ret(ofw, dfw), _ret(ofg, dfgamma), _ret(ofb, dfbeta))


@ctx_bound
class MaskPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, log_m):
        f = f.contiguous()
        log_m = log_m.contiguous()
        S, msum = hip.maskpool_fwd(f, log_m)
        ctx.save_for_backward(f, log_m)
        return S, msum

    @staticmethod
    def backward(ctx, gS, gmsum):
        f, log_m = ctx.saved_tensors
        df, dlog_m = hip.maskpool_bwd(f, log_m, gS.contiguous(), gmsum.contiguous())
        return df, dlog_m


# last decoder stage without its normalised activation in memory (GENESIS_FUSE_DEC_HEAD=0: the unfused kernels)
FUSE_DECODER_HEAD = __import__('os').environ.get('GENESIS_FUSE_DEC_HEAD', '1') == '1'
# its GroupNorm statistics out of the transposed conv's epilogue instead of a statistics-only pass over the output
EPILOGUE_STATS = __import__('os').environ.get('GENESIS_DECONV_STATS', '1') == '1'
# first decoder layer on the broadcast latent as one matrix product over tap-summed weights instead of a transposed conv on
# a materialised canvas (GENESIS_BCAST_DECONV=0: canvas + the generic kernels)
BCAST_DECONV = __import__('os').environ.get('GENESIS_BCAST_DECONV', '1') == '1'


@ctx_bound
class DecoderFn(torch.autograd.Function):
    """args: z [K*B, D], coords [1,2,d,d], then 4 x (deconv w, deconv b, gn gamma, gn beta), out w, out b."""

    @staticmethod
    def forward(ctx, z, coords, *params):
        N, D = z.shape
        d = coords.shape[-1]
        z = z.contiguous()
        ctx.bcast = BCAST_DECONV and params[0].shape[0] == D + 2
        h = None
        if not ctx.bcast:
            h = hip.broadcast_concat(z, coords)      # BroadcastLayer + PixelCoords: one launch, no torch.cat
        saved = []
        ow, ob = params[16], params[17]
        ow2 = ow.detach().view(ow.shape[0], -1)
        # last stage: the normalised 64-channel full-resolution activation (the largest tensor of the model) is never
        # written -- the output conv normalises the pre-norm tensor on load, forward and backward
        Cl, Sl = params[12].shape[1], 16 * d
        ctx.fused_head = FUSE_DECODER_HEAD and Cl <= 64 and Cl % GROUPS == 0 and (Sl * Sl) % 256 == 0 \
            and ow.shape[0] <= 8
        am_h = None           # partial maxima of the current layer's input (hip_ops.Amax), for the fp16-piece weight gradients
        ctx.am_h = [None] * 4
        for l in range(4):
            ctx.am_h[l] = am_h
            w, b, gamma, beta = params[4 * l:4 * l + 4]
            if l == 3 and ctx.fused_head:
                if EPILOGUE_STATS:
                    y, mean, rstd = hip.deconv5x5s2_gn_stats_fwd(h, w, b, gamma, beta, GROUPS, EPS)
                else:
                    y, mean, rstd = hip.deconv5x5s2_gn_relu_fwd(h, w, b, gamma, beta, GROUPS, EPS, None)
                saved.append((h, y, mean, rstd))
                out = hip.conv1x1_gn_fwd(y, mean, rstd, gamma, beta, GROUPS, ow2, ob)
                h = None
                break
            if l == 0 and ctx.bcast:
                # the input is constant over the pixels: out = z @ (tap-summed weights) + (bias + coordinate channels)
                wz, bz = hip.bcast_deconv_pack(w, b, coords)
                y = hip.linear_fwd(z, wz, bz).view(N, w.shape[1], 2 * d, 2 * d)
                a = torch.empty_like(y)
                mean, rstd = hip.gn_relu_fwd(y, gamma, beta, GROUPS, EPS, (a, 0, 0))
                am_h = hip.take_amax()
                saved.append(((z, wz, coords), y, mean, rstd))
                h = a
                continue
            a = torch.empty(N, w.shape[1], 2 * h.shape[2], 2 * h.shape[3], device=h.device)
            # (the activation's partial maxima go to the next layer's transposed conv: gx_kq_amax_link, DESIGN finding 40)
            y, mean, rstd = hip.deconv5x5s2_gn_relu_fwd(h, w, b, gamma, beta, GROUPS, EPS, (a, 0, 0), link_out=l < 3)
            am_h = hip.take_amax()
            saved.append((h, y, mean, rstd))
            h = a
        if not ctx.fused_head:
            out = hip.conv1x1_fwd(h, ow2, ob)
        ctx.saved = saved
        ctx.last = h
        ctx.params = params
        ctx.D = D
        return out

    @staticmethod
    def backward(ctx, g):
        params = ctx.params
        ow, ob = params[16], params[17]
        gow, gob = _gout(ow), _gout(ob)
        ow2 = ow.detach().view(ow.shape[0], -1)
        g = g.contiguous()
        head = None
        if ctx.fused_head:
            # norm backward, the conv's data gradient (formed on load) and its weight gradient behind ONE pass over y
            _, y3, mean3, rstd3 = ctx.saved[3]
            o3 = (_gout(params[14]), _gout(params[15]), _gout(params[13]))
            link = hip.amax_link(g.device, ctx.saved[3][1].numel())       # dy's partial maxima for the data gradient below (DESIGN finding 40 (a)); noqa: F841
            head = hip.conv1x1_gn_bwd_fused(y3, params[14], params[15], mean3, rstd3, GROUPS, g, ow2, ob, None, True,
                                            out_gn=o3, out_conv=(gow, gob, None))
            if head is None:
                dow, dob, _ = hip.conv1x1_gn_wgrad(y3, mean3, rstd3, params[14], params[15], GROUPS, g,
                                                   out=(gow, gob, None))
                head = hip.gn_relu_bwd_proj(y3, params[14], params[15], mean3, rstd3, GROUPS, g, ow2, True, out=o3)
                head = (head[0], head[1:], None)
            else:
                dow, dob = head[2][0], head[2][1]
            am_head = hip.take_amax()         # (both forms end in gx_gn_relu_bwd_proj: dy's partial maxima)
            da = None
        else:
            da, dow, dob, _ = hip.conv1x1_bwd(ctx.last, g, ow2, ob, out=(gow, gob, None))
        grads = [None] * 18
        grads[16], grads[17] = _ret(gow, dow.view(ow.shape)), _ret(gob, dob)
        for l in reversed(range(4)):
            w, b, gamma, beta = params[4 * l:4 * l + 4]
            h, y, mean, rstd = ctx.saved[l]
            if l == 3 and ctx.fused_head:
                ow, (og, ob, obias) = _gout(w), o3
                dy, (dgamma, dbeta, dbias) = head[0], head[1]
                am_dy = am_head
            else:
                ow, obias, og, ob = _gout(w), _gout(b), _gout(gamma), _gout(beta)
                link = hip.amax_link(da.device, y.numel()) if l > 0 and hip.AMAX_LINK_REG else None     # dy's partial maxima for this layer's data gradient; noqa: F841
                dy, dgamma, dbeta, dbias = hip.gn_relu_bwd(y, gamma, beta, mean, rstd, GROUPS, (da, 0, 0), None, True,
                                                           out=(og, ob, obias))
                am_dy = hip.take_amax()
            if l == 0 and ctx.bcast:
                # one launch: dz = dy Wz^T, d(Wz^T) = dy^T z, d(bias map) = column sums of dy; then the tap sums are folded
                # back onto the 5x5 weights (the bias gradient came out of the norm backward above)
                z, wz, coords = h
                dz, dwz, dbz = hip.linear_bwd(z, wz, None, dy.view(dy.shape[0], -1), None)
                dw, _ = hip.bcast_deconv_unpack(dwz, dbz, coords, w.shape[1], out_dw=ow)
                grads[0:4] = [_ret(ow, dw), _ret(obias, dbias), _ret(og, dgamma), _ret(ob, dbeta)]
                _decoder_backward_done()
                return (dz, None) + tuple(grads)
            am = (am_dy, ctx.am_h[l])
            dw = _wgrad(lambda h=h, dy=dy, ow=ow, am=am: hip.deconv5x5s2_wgrad(h, dy, out=ow, amax=am), ow, h, dy)
            # the first layer's input is the broadcast latent + 2 coordinate channels: only the D latent
            # channels need a gradient
            da = hip.deconv5x5s2_dgrad(dy, w, ctx.D if l == 0 else None)
            grads[4 * l:4 * l + 4] = [_ret(ow, dw), _ret(obias, dbias), _ret(og, dgamma), _ret(ob, dbeta)]
        dz = da.sum((2, 3))
        _decoder_backward_done()
        return (dz, None) + tuple(grads)


@ctx_bound
class MixtureFn(torch.autograd.Function):
    """returns (err [B], recon [B,3,H,W], x_r [K,B,3,H,W], log_m_r [K,B,1,H,W])."""

    @staticmethod
    def forward(ctx, x, dec, K, pixel_std, pixel_bound):
        x = x.contiguous()
        dec = dec.contiguous()
        err, recon, x_r, log_m_r = hip.mixture_fwd(x, dec, K, pixel_std, pixel_bound)
        ctx.save_for_backward(x, dec)
        ctx.cfg = (K, pixel_std, pixel_bound)
        ctx.mark_non_differentiable(recon, x_r, log_m_r)
        ctx.set_materialize_grads(False)
        return err, recon, x_r, log_m_r

    @staticmethod
    def backward(ctx, g_err, *unused):
        x, dec = ctx.saved_tensors
        K, pixel_std, pixel_bound = ctx.cfg
        if g_err is None:
            g_err = x.new_zeros(x.shape[0])
        ddec = hip.mixture_bwd(x, dec, g_err.contiguous(), K, pixel_std, pixel_bound)
        return None, ddec, None, None, None


# ---------------------------------------------------------------------------------------------- MONet / ComponentVAE
@ctx_bound
class Conv1x1Fn(torch.autograd.Function):
    """Small 1x1 conv with bias (Cout <= 8): the MONet UNet's final_conv (modules/unet.py:66,90)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        w2 = w.reshape(w.shape[0], -1)
        ctx.save_for_backward(x, w2)
        ctx.params = (w, b)
        return hip.conv1x1_fwd(x, w2, b)

    @staticmethod
    def backward(ctx, g):
        x, w2 = ctx.saved_tensors
        w, b = ctx.params
        # weight and bias gradients straight into the bucket; a conv used again in the iteration (MONet's K-1 UNet passes
        # share final_conv) ADDS inside the finishing kernel
        (ow, acc_w), (ob, acc_b) = _gout_acc(w), (_gout_acc(b) if b is not None else (None, False))
        if ow is None or (b is not None and (ob is None or acc_b != acc_w)):
            ow = ob = None
            acc_w = False
        dx, dw, db, _ = hip.conv1x1_bwd(x, g.contiguous(), w2, b, out=(ow.view(w2.shape) if ow is not None else None, ob, None),
                                        accumulate=acc_w)
        return dx, _ret(ow, dw.view(w.shape) if ow is None else None), (_ret(ob, db) if b is not None else None)


@ctx_bound
class MaskImageConvActFn(torch.autograd.Function):
    """The ComponentVAE encoder's first layer on its slot-major input [log_m_k | x] (modules/component_vae.py:59-66,
    modules/encoders.py:31-34): act(conv3x3 s2 p1(cat(log_m [K,B,1,H,W], x [B,3,H,W] repeated K times))) -> [K*B,Cout,H/2,W/2].
    The input is stacked by one kernel (no x.repeat + torch.cat), and the backward returns the mask channel's gradient as a
    compact [K,B,1,H,W] tensor (no zero-filled 4-channel dx whose first channel autograd then adds as a strided slice)."""

    @staticmethod
    def forward(ctx, log_m, x, w, b, act):
        log_m, x = log_m.contiguous(), x.contiguous()
        inp = hip.mask_image_stack(log_m, x)
        y = hip.conv2d_direct_fwd(inp, w, b, act, 2, 1)
        ctx.save_for_backward(inp, y)
        ctx.params = (w, b)
        ctx.cfg = (act, log_m.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        inp, y = ctx.saved_tensors
        w, b = ctx.params
        act, mshape = ctx.cfg
        ow, ob = _gout(w), _gout(b)
        dy, db = hip.bias_act_bwd(y, g.contiguous(), act, True, ob)
        H, W = inp.shape[2], inp.shape[3]
        even = H % 2 == 0 and W % 2 == 0
        if even:
            dw = hip.conv3x3s2_wgrad_small(inp, dy, out=ow)
        else:
            dw = hip.conv2d_direct_wgrad(inp, dy, 3, 2, 1, out=ow)
        dm = None
        if ctx.needs_input_grad[0]:
            if even and w.shape[0] <= 256:
                dm = hip.conv3x3s2_dgrad_lead(dy, w, H, W, 1).view(mshape)
            else:
                dm = hip.conv2d_direct_dgrad(dy, w, H, W, 2, 1)[:, :1].reshape(mshape)
        return dm, None, _ret(ow, dw), _ret(ob, db), None


@ctx_bound
class DirectConvActFn(torch.autograd.Function):
    """act(conv2d(x, w, b, stride, pad)) through the generic direct kernel (MONetCompEncoder's stride-2 convs,
    modules/encoders.py:31-34)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, dx_channels=None):
        """dx_channels: only the first dx_channels input channels carry a gradient (the encoder's first layer: the mask
        channel; the image channels are data) -- the others' dx is returned as zeros."""
        x = x.contiguous()
        y = hip.conv2d_direct_fwd(x, w, b, act, stride, pad)
        ctx.save_for_backward(x, y)
        ctx.params = (w, b)
        ctx.cfg = (stride, pad, act, dx_channels)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        w, b = ctx.params
        stride, pad, act, dx_channels = ctx.cfg
        ow, ob = _gout(w), _gout(b)
        dy, db = hip.bias_act_bwd(y, g.contiguous(), act, True, ob)
        H, W = x.shape[2], x.shape[3]
        s2 = w.shape[2] == 3 and stride == 2 and pad == 1 and H % 2 == 0 and W % 2 == 0
        if s2:
            dw = hip.conv3x3s2_wgrad_small(x, dy, out=ow)
        else:
            dw = hip.conv2d_direct_wgrad(x, dy, w.shape[2], stride, pad, out=ow)
        dx = None
        if ctx.needs_input_grad[0]:
            if s2 and w.shape[0] <= 256:
                dx = hip.conv3x3s2_dgrad_small(dy, w, H, W, dx_channels)      # no structural zeros, vector ALUs
            else:
                dx = hip.conv2d_direct_dgrad(dy, w, H, W, stride, pad)
        return dx, _ret(ow, dw), _ret(ob, db), None, None, None, None


_ROWCOL = {}


def _row_col_coords(coords):
    """(row vector, column vector) of a PixelCoords buffer [1,2,d,d], made contiguous once per buffer (it never changes)."""
    key = (coords.data_ptr(), coords.shape[-1], str(coords.device))
    rc = _ROWCOL.get(key)
    if rc is None or rc[2] is not coords:
        if len(_ROWCOL) > 64:
            _ROWCOL.clear()
        rc = _ROWCOL[key] = (coords[0, 0, :, 0].contiguous(), coords[0, 1, 0, :].contiguous(), coords)
    return rc[0], rc[1]


@ctx_bound
class BroadcastDecoderFn(torch.autograd.Function):
    """BroadcastDecoder (modules/decoders.py:21-35): z [N, L] -> [N, out, S, S].
    args: z, coords [1,2,S+2L,S+2L], act, out_act, then L x (w [h,cin,3,3], b [h]), out_w [out, h], out_b.
    The L VALID 3x3 convs run as 'same' convs on the (S+2L)^2 canvas; the centre crop of the final 1x1 conv equals the
    valid chain exactly (and so do all gradients: positions polluted by the canvas border never reach the crop).
    The spatial broadcast + coordinate concat (modules/blocks.py:104-130) is never materialised: the first conv is
    evaluated from z, the tap sums of its weights and the row / column coordinate vectors (gx_bcast_conv3x3_*), the
    remaining convs on the fp32 MFMA tap-conv kernel."""

    @staticmethod
    def forward(ctx, z, coords, act, out_act, *params):
        """out_act: activation applied to the final 1x1 conv's output (None; 'elu' for BaselineVAE's broadcast decoder,
        whose Sequential ends in nn.ELU, vae_config.py:54-60)."""
        nl = (len(params) - 2) // 2
        N, L = z.shape
        d = coords.shape[-1]
        S = d - 2 * nl
        z = z.contiguous()
        rowc, colc = _row_col_coords(coords)         # g_1 varies along rows, g_2 along columns (blocks.py:121-126)
        acts = []
        h = None
        am = None                  # partial maxima of h, left by the launch that wrote it (the next layer's fp16 scale without a pass)
        for l in range(nl):
            w, b = params[2 * l], params[2 * l + 1]
            if l == 0:
                tap = CHAIN_TAPS and nl > 1
                y = hip.bcast_conv3x3_fwd(z, w, b, rowc, colc, act, tap=tap)
                am = hip.take_amax() if tap else None
            else:
                tap = CHAIN_TAPS and l + 1 < nl
                y = hip.conv3x3_bias_act_fwd(h, w, b, act, amax_in=am, tap=tap)
                am = hip.take_amax() if tap else None
            acts.append((h, y))
            h = y
            # (test diagnostic: the reference's VALID conv l has the canvas minus l + 1 border pixels as its output)
            hip._probe_act(y[:, :, l + 1:d - l - 1, l + 1:d - l - 1], b, act)
        ow, ob = params[2 * nl], params[2 * nl + 1]
        wide = ow.shape[0] > 8 or out_act is not None      # the small-Cout 1x1 kernel serves <= 8 output channels
        if wide:
            full = hip.conv2d_direct_fwd(h, ow.reshape(ow.shape[0], -1, 1, 1), ob, out_act, 1, 0)
        else:
            full = hip.conv1x1_fwd(h, ow, ob)
        ctx.acts, ctx.params, ctx.cfg = acts, params, (nl, S, L, act, out_act, wide)
        ctx.bc = (z, rowc, colc)
        ctx.full = full if wide else None
        return full[:, :, nl:nl + S, nl:nl + S].contiguous()

    @staticmethod
    def backward(ctx, g):
        nl, S, L, act, out_act, wide = ctx.cfg
        params = ctx.params
        z, rowc, colc = ctx.bc
        ow, ob = params[2 * nl], params[2 * nl + 1]
        last = ctx.acts[-1][1]
        gfull = torch.zeros(last.shape[0], ow.shape[0], last.shape[2], last.shape[3], device=g.device)
        gfull[:, :, nl:nl + S, nl:nl + S] = g
        pre = None                 # (dy, db, db's buffer) of layer l, already formed by the data gradient of the layer after it
        am0 = None
        if wide:
            gow, gob = _gout(ow), _gout(ob)
            dyo, dob = hip.bias_act_bwd(ctx.full, gfull, out_act, True, gob)
            ow4 = ow.reshape(ow.shape[0], -1, 1, 1)
            dow = hip.conv2d_direct_wgrad(last, dyo, 1, 1, 0, out=None if gow is None else gow.view(ow4.shape))
            da = hip.conv2d_direct_dgrad(dyo, ow4, last.shape[2], last.shape[3], 1, 0)
            dow, dob = _ret(gow, dow.view(ow.shape)), _ret(gob, dob)
        elif nl >= 2 and hip.ACTS[act] in (1, 2) and os.environ.get('GENESIS_DGRAD_ACT_FUSE', '1') != '0':
            # the last 3x3 layer's bias + activation backward inside the 1x1 conv's data gradient (`last` is its output)
            gbp = _gout(params[2 * nl - 1])
            wp = params[2 * nl - 2]
            lazy = QUAD_BIAS and wp.shape[0] == wp.shape[1] and _quad_ok(last.shape[0], wp.shape[1], last.shape[2], last.shape[3])
            gow, gob = _gout(ow), _gout(ob)
            if gow is None or gob is None:
                gow = gob = None
            dyl, dow, dob, dbl = hip.conv1x1_bwd_act(last, gfull, ow, ob, act, out=(gow, gob), dbx_out=gbp, want_dbx=not lazy,
                                                     tap=CHAIN_TAPS)
            dow, dob = _ret(gow, dow), _ret(gob, dob)
            pre = (dyl, dbl, gbp)
            am0 = hip.take_amax() if CHAIN_TAPS else None      # (dyl's partial maxima: the first data gradient's fp16 scale)
        else:
            gow, gob = _gout(ow), _gout(ob)
            if gow is None or gob is None:
                gow = gob = None
            da, dow, dob, _ = hip.conv1x1_bwd(last, gfull, ow, ob, out=(gow, gob, None))
            dow, dob = _ret(gow, dow), _ret(gob, dob)
        grads = [None] * len(params)
        grads[2 * nl], grads[2 * nl + 1] = dow, dob
        dz = None
        am = am0                   # partial maxima of the current dy (see forward)
        for l in reversed(range(nl)):
            w, b = params[2 * l], params[2 * l + 1]
            h, y = ctx.acts[l]
            gw = _gout(w)
            gb = pre[2] if pre is not None else _gout(b)          # (_gout hands a parameter's buffer out once per iteration)
            if l == 0:
                # every gradient of the broadcast layer from seven sums per (slot, channel) plane: no canvas, no dy
                dz, dw, db = hip.bcast_conv3x3_bwd(y, da, z, w, rowc, colc, act, out=(gw, gb))
            else:
                dy, db = pre[:2] if pre is not None else hip.bias_act_bwd(y, da, act, True, gb)
                if pre is not None and db is None:
                    # the bias gradient was left to this layer's weight gradient (it reads all of dy anyway)
                    db = gb if gb is not None else torch.empty(b.shape, dtype=torch.float32, device=dy.device)
                    dw = _wgrad_paired(h, dy, gw, db)
                else:
                    dw = _wgrad_paired(h, dy, gw)
                pre = None
                if l >= 2 and hip.conv3x3_dgrad_act_supported(dy.shape[0], w.shape[1], w.shape[0], dy.shape[2], dy.shape[3]):
                    # layer l - 1's bias + activation backward in this data gradient's epilogue (h is its output)
                    gbp = _gout(params[2 * l - 1])
                    wp = params[2 * l - 2]
                    lazy = QUAD_BIAS and wp.shape[0] == wp.shape[1] and _quad_ok(h.shape[0], wp.shape[1], h.shape[2], h.shape[3])
                    pre = hip.conv3x3_dgrad_act(dy, w, h, act, gbp, want_dbias=not lazy, amax_in=am, tap=CHAIN_TAPS) + (gbp,)
                    am = hip.take_amax() if CHAIN_TAPS else None
                else:
                    da = hip.conv3x3_dgrad(dy, w, amax_in=am)
                    am = None
            grads[2 * l], grads[2 * l + 1] = _ret(gw, dw), _ret(gb, db)
        return (dz, None, None, None) + tuple(grads)


CHAIN_TAPS = os.environ.get('GENESIS_BCAST_CHAIN_TAPS', '1') != '0'   # 0: every canvas conv makes its own amax pass over its input
QUAD_BIAS = os.environ.get('GENESIS_QUAD_WGRAD_BIAS', '1') != '0'     # 0: the fused data gradients' own plane-sum pass


def _quad_ok(N, C, H, W):
    return hip.conv3x3_wgrad_quad_supported(N, C, H, W)


def _wgrad_paired(h, dy, gw, gb=None):
    """conv3x3 weight gradient of a layer with <= 32 channels on both sides (the BroadcastDecoder's 32 -> 32 convs on the
    canvas): the kernels work on 64 x 64 channel blocks, so such a layer fills a quarter of every MFMA.  Two consecutive
    images of an NCHW tensor ARE one image of twice the channels ([N, 32, H, W] viewed as [N / 2, 64, H, W]): the 64 x 64
    weight gradient of the paired tensors holds the even images' sum in its upper-left 32 x 32 block and the odd images'
    in its lower-right one (the off-diagonal blocks are cross-image products nobody wants) -- half of the MFMA work is
    useful instead of a quarter, with no kernel change.  Returns dw (or None after writing gw)."""
    N, Ci, Co = h.shape[0], h.shape[1], dy.shape[1]
    if Ci == Co and hip.conv3x3_wgrad_quad_supported(N, Ci, h.shape[2], h.shape[3]):
        # four images per workgroup tile, one per wave: every wave's 32 x 32 block is a wanted one (gb: a [C] buffer that
        # receives the layer's bias gradient, the channel sums of dy, from the same read of dy)
        dw = hip.conv3x3_wgrad_quad(h, dy, out=gw, dbias_out=gb)
        return None if gw is not None else dw
    assert gb is None, 'the bias gradient rides on the four-image weight gradient only'
    if N % 2 or Ci > 32 or Co > 32 or Ci != Co or Ci % 8:
        return hip.conv3x3_wgrad(h, dy, out=gw)
    d64 = hip.conv3x3_wgrad(h.view(N // 2, 2 * Ci, *h.shape[2:]), dy.view(N // 2, 2 * Co, *dy.shape[2:]))
    dw = d64[:Co, :Ci] + d64[Co:, Ci:]
    if gw is not None:
        gw.copy_(dw)
        return None
    return dw


@ctx_bound
class MixtureWFn(torch.autograd.Function):
    """Mixture likelihood with the ATTENTION masks as mixing weights (models/monet_config.py:94-105).
    returns (err [B], recon [B,3,H,W], x_r [K,B,3,H,W]); differentiable w.r.t. dec and log_w."""

    @staticmethod
    def forward(ctx, x, dec, log_w, K, std1, std2, pixel_bound):
        x, dec, log_w = x.contiguous(), dec.contiguous(), log_w.contiguous()
        err, recon, x_r = hip.mixture_w_fwd(x, dec, log_w, K, std1, std2, pixel_bound)
        ctx.save_for_backward(x, dec, log_w)
        ctx.cfg = (K, std1, std2, pixel_bound)
        ctx.mark_non_differentiable(recon, x_r)
        ctx.set_materialize_grads(False)
        return err, recon, x_r

    @staticmethod
    def backward(ctx, g_err, *unused):
        x, dec, log_w = ctx.saved_tensors
        K, std1, std2, pixel_bound = ctx.cfg
        if g_err is None:
            g_err = x.new_zeros(x.shape[0])
        ddec, dlog_w = hip.mixture_w_bwd(x, dec, log_w, g_err.contiguous(), K, std1, std2, pixel_bound)
        return None, ddec, dlog_w, None, None, None, None


# ---------------------------------------------------------------------------------------------- slot latents
def _c(t):
    return None if t is None else t.contiguous()


@ctx_bound
class PosteriorFn(torch.autograd.Function):
    """zh [B,K,2D] = z_head(obj), eps [K,B,D] -> (z, mu, sigma [K,B,D], log_q [K,B]): to_sigma, rsample and
    q_z.log_prob(z).sum(1) of models/genesisv2_config.py:154-160 / models/genesis_config.py:329 in one launch."""

    @staticmethod
    def forward(ctx, zh, eps):
        zh, eps = zh.contiguous(), eps.contiguous()
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(zh, eps)
        return hip.latent_posterior_fwd(zh, eps)

    @staticmethod
    def backward(ctx, gz, gmu, gsigma, glogq):
        zh, eps = ctx.saved_tensors
        return hip.latent_posterior_bwd(zh, eps, _c(gz), _c(gmu), _c(gsigma), _c(glogq)), None


@ctx_bound
class PriorLogPFn(torch.autograd.Function):
    """z [K,B,D], lin [K-1,B,2D] (or None), log_q [K,B] (or None) -> log_p [K,B] under N(0,1) for the first slot and
    N(tanh(lin[:D]), sigmoid(lin[D:] + 4) + 1e-4) for the others (models/genesis_config.py:297-330); with log_q the
    Monte-Carlo KL sample log_q - log_p of :329-331 (one launch for the whole KL term)."""

    @staticmethod
    def forward(ctx, z, lin, log_q=None, all_slots=False):
        """all_slots: lin [K,B,2D] -- every slot has a conditional prior (Genesis' component prior p(z_c | z_m),
        models/genesis_config.py:229-247)."""
        z = z.contiguous()
        lin = None if lin is None else lin.contiguous()
        log_q = None if log_q is None else log_q.contiguous()
        ctx.save_for_backward(z, lin)
        ctx.kl_mode = log_q is not None
        ctx.all_slots = bool(all_slots)
        return hip.latent_prior_logp_fwd(z, lin, log_q, all_slots)

    @staticmethod
    def backward(ctx, g):
        z, lin = ctx.saved_tensors
        g = g.contiguous()
        dz, dlin = hip.latent_prior_logp_bwd(z, lin, g, ctx.kl_mode, ctx.all_slots)
        return dz, dlin, (g if ctx.kl_mode else None), None


@ctx_bound
class ElboFn(torch.autograd.Function):
    """Loss aggregation of train.py:226-242: (err [B], kl [R,B] | None, beta [1] device scalar) ->
    (loss [1] = err_mean + beta kl_mean, out5 = (loss, err_mean + kl_mean, err_mean, kl_mean, beta), not differentiable).
    `loss` is its own one-element tensor so that loss.backward() starts the backward pass without a select/scatter."""

    @staticmethod
    def forward(ctx, err, kl, beta, tail):
        err = err.contiguous()
        kl = None if kl is None else kl.contiguous()
        ctx.beta = beta
        ctx.dims = (err.numel(), 0 if kl is None else kl.numel() // err.numel())
        ctx.kl_shape = None if kl is None else kl.shape
        loss, out5 = hip.elbo_fwd(err, kl, beta, tail)
        ctx.mark_non_differentiable(out5)
        ctx.set_materialize_grads(False)
        return loss, out5

    @staticmethod
    def backward(ctx, g, _unused):
        B, R = ctx.dims
        d_err, d_kl = hip.elbo_bwd(g.contiguous(), ctx.beta, B, R)
        return d_err, (d_kl.view(ctx.kl_shape) if d_kl is not None else None), None, None


@ctx_bound
class PooledHeadFn(torch.autograd.Function):
    """(lin [R,C], msum [R], feat_head[1].bias, LayerNorm weight, bias, eps) -> LayerNorm((lin + msum b)/(msum+1e-5))
    (models/genesisv2_config.py:146-154 and z_head[0], :76), one launch forward, two backward."""

    @staticmethod
    def forward(ctx, lin, msum, fbias, gamma, beta, eps):
        shape = lin.shape
        lin2 = lin.contiguous().view(-1, shape[-1])
        msum1 = msum.contiguous().view(-1)
        y, stats = hip.pooled_head_fwd(lin2, msum1, fbias, gamma, beta, eps)
        ctx.save_for_backward(lin2, msum1, stats)
        ctx.params = (fbias, gamma, beta)
        ctx.shapes = (shape, msum.shape)
        return y.view(shape)

    @staticmethod
    def backward(ctx, g):
        lin2, msum1, stats = ctx.saved_tensors
        fbias, gamma, beta = ctx.params
        outs = (_gout(fbias), _gout(gamma), _gout(beta))
        dlin, dmsum, dfb, dga, dbe = hip.pooled_head_bwd(lin2, msum1, fbias, gamma, stats,
                                                         g.contiguous().view(lin2.shape), out=outs)
        return (dlin.view(ctx.shapes[0]), dmsum.view(ctx.shapes[1]), _ret(outs[0], dfb), _ret(outs[1], dga),
                _ret(outs[2], dbe), None)


@ctx_bound
class SBPScanFn(torch.autograd.Function):
    """T stick-breaking steps in log space (modules/attention.py:42-48,118-124): logits [T, ...], log_s0 [...] or None ->
    (log_m [T, ...], log_s [T, ...] = the scope after each step); last_scope: the last mask is the remaining scope."""

    @staticmethod
    def forward(ctx, logits, log_s0, last_scope):
        logits = logits.contiguous()
        log_s0 = None if log_s0 is None else log_s0.contiguous()
        ctx.save_for_backward(logits)
        ctx.last_scope = bool(last_scope)
        ctx.has_s0 = log_s0 is not None
        ctx.set_materialize_grads(False)
        return hip.sbp_scan_fwd(logits, log_s0, last_scope)

    @staticmethod
    def backward(ctx, g_m, g_s):
        logits, = ctx.saved_tensors
        want_s0 = ctx.has_s0 and ctx.needs_input_grad[1]
        g_logits, g_s0 = hip.sbp_scan_bwd(logits, _c(g_m), _c(g_s), ctx.last_scope, want_s0)
        return g_logits, g_s0, None


@ctx_bound
class CategoricalKLFn(torch.autograd.Function):
    """MONet.kl_m_loss (models/monet_config.py:157-170): log_m, log_m_r [K,B,1,H,W] -> kl_m [B]; the gradient reaches
    log_m_r only if it requires one (detach_mr_in_klm = False, models/genesisv2_config.py:172-176)."""

    @staticmethod
    def forward(ctx, log_m, log_m_r):
        log_m, log_m_r = log_m.contiguous(), log_m_r.contiguous()
        ctx.save_for_backward(log_m, log_m_r)
        return hip.categorical_kl_fwd(log_m, log_m_r)

    @staticmethod
    def backward(ctx, g):
        log_m, log_m_r = ctx.saved_tensors
        g_m, g_r = hip.categorical_kl_bwd(log_m, log_m_r, g.contiguous(), ctx.needs_input_grad[1])
        return g_m, g_r


@ctx_bound
class MaskReconFn(torch.autograd.Function):
    """log_m_r as a differentiable function of the decoder output: the mixture kernel already produced the values
    (log_softmax over K of dec's last channel, monet_config.py:137-139); this node only routes a gradient on them back
    into dec (needed when the mask KL does not detach the reconstructed masks, genesisv2_config.py:172-176)."""

    @staticmethod
    def forward(ctx, dec, log_m_r):
        ctx.save_for_backward(log_m_r)
        ctx.C = dec.shape[1]
        return log_m_r.view_as(log_m_r)

    @staticmethod
    def backward(ctx, g):
        log_m_r, = ctx.saved_tensors
        return hip.logsoftmax_k_bwd(log_m_r, g.contiguous(), ctx.C), None


@ctx_bound
class LSTMCellFn(torch.autograd.Function):
    """One nn.LSTM cell step whose input depends on the previous step's output (LatentSBP, modules/attention.py:103-110:
    the sampled z_{k-1} is fed back): inp [B,Din], h_prev / c_prev [B,H] or None (zero state) -> (h, c).  Input
    projection on the dense kernel, recurrent GEMM + gate / cell update in gx_lstm_step_fwd; backward = the step's
    backward kernel + two dense backward launches."""

    @staticmethod
    def forward(ctx, inp, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh):
        inp = inp.contiguous()
        B = inp.shape[0]
        H = w_hh.shape[1]
        gx = hip.linear_fwd(inp, w_ih, b_ih)
        act = torch.empty(B, 4 * H, device=inp.device)
        c = torch.empty(B, H, device=inp.device)
        h = torch.empty(B, H, device=inp.device)
        hp = None if h_prev is None else h_prev.contiguous()
        cp = None if c_prev is None else c_prev.contiguous()
        hip.lstm_step_fwd(gx, hp, cp, w_hh, b_hh, act, c, h)
        ctx.save_for_backward(inp, hp, cp, act, c)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        ctx.set_materialize_grads(False)
        return h, c

    @staticmethod
    def backward(ctx, g_h, g_c):
        inp, hp, cp, act, c = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh = ctx.params
        B, H = c.shape
        dev = inp.device
        if g_h is None:
            g_h = torch.zeros(B, H, device=dev)
        dgates = torch.empty(B, 4 * H, device=dev)
        dc_prev = torch.empty(B, H, device=dev)
        hip.lstm_step_bwd(g_h.contiguous(), None, w_hh, act, c, cp, _c(g_c), dgates, dc_prev)
        # the cell's four parameters travel together; a cell used again in the iteration (LatentSBP's K-1 steps share one
        # LSTM) ADDS its gradients inside the dense kernels instead of through autograd's accumulation launches
        outs = [_gout_acc(p) for p in (w_ih, w_hh, b_ih, b_hh)]
        acc = outs[0][1]
        if any(o is None for o, _ in outs) or any(a != acc for _, a in outs):
            outs, acc = [(None, False)] * 4, False
        o_wih, o_whh, o_bih, o_bhh = (o for o, _ in outs)
        dinp, dw_ih, db = hip.linear_bwd(inp, w_ih, None, dgates, None, need_dx=ctx.needs_input_grad[0],
                                         out_dw=o_wih, out_db=o_bih, out_db2=o_bhh, accumulate_dw=acc)
        if hp is not None:
            dh_prev, dw_hh, _ = hip.linear_bwd(hp, w_hh, None, dgates, None, need_dx=True, need_db=False, out_dw=o_whh,
                                               accumulate_dw=acc)
        else:
            dh_prev = None
            dw_hh = torch.zeros_like(w_hh) if o_whh is None else (o_whh if acc else o_whh.zero_())
        return (dinp, dh_prev, dc_prev if cp is not None else None, _ret(o_wih, dw_ih), _ret(o_whh, dw_hh),
                _ret(o_bih, db), _ret(o_bhh, db))


@ctx_bound
class LatentSBPPosteriorFn(torch.autograd.Function):
    """The recurrent posterior of GENESIS' LatentSBP (modules/attention.py:84-118) as ONE autograd node: encoder features h
    [B,F], eps [K,B,D] -> (z, mu, sigma [K,B,D], log_q [K,B]).  Slot 0 from the VAE's own heads (q_z_mean | q_z_var of
    third_party/sylvester/VAE.py:118-121), slot k >= 1 from lstm(cat(h, z_{k-1})) -> linear -> (mean | pre-sigma), every
    slot sampled by gx_latent_posterior_fwd_ex.  Same launches per slot as LSTMCellFn -> LinearFn -> PosteriorFn chained,
    but the uses of h (K + 1 of them), z_{k-1}, the LSTM state and the K small outputs no longer meet in autograd:
      * the LSTM input rows [h | z_{k-1}] live in one [K-1,B,F+D] buffer (h broadcast once, the posterior kernel writes z's
        second copy into the next row block: no torch.cat per step, no cat backward);
      * backward is the time loop in reverse, the z columns of the input-projection gradient ADDED inside the posterior's
        backward kernel (gz2), the recurrent dh inside gx_lstm_step_bwd -- no accumulation launches;
      * the weight gradients of the cell and of the output linear are ONE dense launch each over all K-1 steps."""

    @staticmethod
    def forward(ctx, h, eps, w_m, b_m, w_v, b_v, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin):
        h, eps = h.contiguous(), eps.contiguous()
        K, B, D = eps.shape
        F_, T, H = h.shape[1], K - 1, w_hh.shape[1]
        dev = h.device
        lin = torch.empty(K, B, 2 * D, device=dev)
        z, mu, sigma = (torch.empty(K, B, D, device=dev) for _ in range(3))
        log_q = torch.empty(K, B, device=dev)
        inp = torch.empty(max(T, 1), B, F_ + D, device=dev)
        if T:
            inp[:, :, :F_] = h                                                   # (one broadcast copy)
        act = torch.empty(max(T, 1), B, 4 * H, device=dev)
        c = torch.empty(max(T, 1), B, H, device=dev)
        hs = torch.empty(max(T, 1), B, H, device=dev)
        hip.linear_fwd(h, w_m, b_m, None, out=lin[0][:, :D])
        hip.linear_fwd(h, w_v, b_v, None, out=lin[0][:, D:])
        for k in range(K):
            if k:
                t = k - 1
                gx = hip.linear_fwd(inp[t], w_ih, b_ih)
                hip.lstm_step_fwd(gx, hs[t - 1] if t else None, c[t - 1] if t else None, w_hh, b_hh, act[t], c[t], hs[t])
                hip.linear_fwd(hs[t], w_lin, b_lin, None, out=lin[k])
            hip.latent_posterior_step_fwd(lin[k], eps[k], z[k], mu[k], sigma[k], log_q[k],
                                          inp[k][:, F_:] if k < T else None)
        ctx.save_for_backward(h, eps, lin, inp, act, c, hs)
        ctx.params = (w_m, b_m, w_v, b_v, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin)
        ctx.set_materialize_grads(False)
        return z, mu, sigma, log_q

    @staticmethod
    def backward(ctx, gz, gmu, gsigma, glogq):
        h, eps, lin, inp, act, c, hs = ctx.saved_tensors
        w_m, b_m, w_v, b_v, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin = ctx.params
        K, B, D = eps.shape
        F_, T, H = h.shape[1], K - 1, w_hh.shape[1]
        dev = h.device
        gz, gmu, gsigma, glogq = _c(gz), _c(gmu), _c(gsigma), _c(glogq)
        dlin = torch.empty(K, B, 2 * D, device=dev)
        dinp = torch.empty(max(T, 1), B, F_ + D, device=dev)
        dgates = torch.empty(max(T, 1), B, 4 * H, device=dev)
        dhs = torch.empty(B, H, device=dev)
        dc = [torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)]

        def sl(g, k):
            return None if g is None else g[k]
        for k in reversed(range(K)):
            # slot k's z also fed LSTM step t = k (input row block k): its input-projection gradient is in dinp[k] by now
            hip.latent_posterior_step_bwd(lin[k], eps[k], sl(gz, k), sl(gmu, k), sl(gsigma, k), sl(glogq, k),
                                          dinp[k][:, F_:] if k < T else None, dlin[k])
            if k:
                t = k - 1
                hip.linear_bwd(hs[t], w_lin, None, dlin[k], None, need_dw=False, need_db=False, out_dx=dhs)
                hip.lstm_step_bwd(dhs, dgates[t + 1] if t + 1 < T else None, w_hh, act[t], c[t], c[t - 1] if t else None,
                                  dc[(t + 1) & 1] if t + 1 < T else None, dgates[t], dc[t & 1])
                # dinp[t] = dgates[t] w_ih  (the z columns go to slot t's posterior backward, the h columns are summed below)
                hip.linear_bwd(inp[t], w_ih, None, dgates[t], None, need_dw=False, need_db=False, out_dx=dinp[t])
        # ---- parameter gradients: one launch per weight over all steps
        o_wm, o_bm, o_wv, o_bv = _gout(w_m), _gout(b_m), _gout(w_v), _gout(b_v)
        if o_wm is None or o_bm is None:
            o_wm = o_bm = None
        if o_wv is None or o_bv is None:
            o_wv = o_bv = None
        need_h = ctx.needs_input_grad[0]
        dh = dinp[:, :, :F_].sum(0) if (T and need_h) else None
        g0 = dlin[0]
        dh, dw_m, db_m = hip.linear_bwd(h, w_m, None, g0[:, :D], None, need_dx=need_h, out_dw=o_wm, out_db=o_bm,
                                        accumulate_dx=dh)
        dh, dw_v, db_v = hip.linear_bwd(h, w_v, None, g0[:, D:], None, need_dx=need_h, out_dw=o_wv, out_db=o_bv,
                                        accumulate_dx=dh)
        if T:
            o_wl, o_bl = _gout(w_lin), _gout(b_lin)
            if o_wl is None or o_bl is None:
                o_wl = o_bl = None
            _, dw_lin, db_lin = hip.linear_bwd(hs.view(T * B, H), w_lin, None, dlin[1:].view(T * B, 2 * D), None,
                                               need_dx=False, out_dw=o_wl, out_db=o_bl)
            outs = [_gout(p) for p in (w_ih, w_hh, b_ih, b_hh)]
            if any(o is None for o in outs):
                outs = [None] * 4
            o_wih, o_whh, o_bih, o_bhh = outs
            _, dw_ih, db = hip.linear_bwd(inp.view(T * B, F_ + D), w_ih, None, dgates.view(T * B, 4 * H), None, need_dx=False,
                                          out_dw=o_wih, out_db=o_bih, out_db2=o_bhh)
            if T > 1:
                _, dw_hh, _ = hip.linear_bwd(hs[:-1].view((T - 1) * B, H), w_hh, None, dgates[1:].view((T - 1) * B, 4 * H),
                                             None, need_dx=False, need_db=False, out_dw=o_whh)
            else:
                dw_hh = torch.zeros_like(w_hh) if o_whh is None else o_whh.zero_()
            rec = (_ret(o_wih, dw_ih), _ret(o_whh, dw_hh), _ret(o_bih, db), _ret(o_bhh, db), _ret(o_wl, dw_lin),
                   _ret(o_bl, db_lin))
        else:
            rec = (None,) * 6
        return (dh, None, _ret(o_wm, dw_m), _ret(o_bm, db_m), _ret(o_wv, dw_v), _ret(o_bv, db_v)) + rec


# ---------------------------------------------------------------------------------------------- dense layers
@ctx_bound
class LinearFn(torch.autograd.Function):
    """act(F.linear(x, w, b)) on the 16x16-tile fp32 MFMA dense kernel; x [..., K] (leading dims flattened)."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x2 = x.contiguous().view(-1, x.shape[-1])
        y = hip.linear_fwd(x2, w.view(w.shape[0], -1), b, act)   # w may be a 1x1 conv weight [N,K,1,1]
        ctx.save_for_backward(x2, y if act else None)
        ctx.params = (w, b)
        ctx.act = act
        ctx.xshape = x.shape
        return y.view(x.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x2, y = ctx.saved_tensors
        w, b = ctx.params
        g2 = g.contiguous().view(-1, w.shape[0])
        need_w = ctx.needs_input_grad[1]
        ow, acc_w = _gout_acc(w) if need_w else (None, False)
        ob, acc_b = _gout_acc(b) if (need_w and b is not None) else (None, acc_w)
        if acc_w != acc_b or (ow is None) != (ob is None and b is not None):
            # (a weight and its bias always travel together; anything else goes the plain way)
            ow = ob = None
            acc_w = False
        dx, dw, db = hip.linear_bwd(x2, w.view(w.shape[0], -1), y, g2, ctx.act, need_dx=ctx.needs_input_grad[0],
                                    need_dw=need_w, need_db=b is not None and need_w,
                                    out_dw=ow.view(w.shape[0], -1) if ow is not None else None, out_db=ob,
                                    accumulate_dw=acc_w)
        if dw is not None and ow is None:
            dw = dw.view(w.shape)
        return (dx.view(ctx.xshape) if dx is not None else None, _ret(ow, dw) if need_w else None,
                _ret(ob, db) if (b is not None and need_w) else None, None)


def linear(x, w, b=None, act=None):
    return LinearFn.apply(x, w, b, act)


@ctx_bound
class MatmulNNFn(torch.autograd.Function):
    """x [M,K] @ w.flatten(1) [K,N] with the weight in [in, out...] layout: the gated ConvTranspose2d 'fc' layer of the
    sylvester stacks on a 1 x 1 input (third_party/sylvester/VAE.py:27-33) as the matrix product it is (gx_matmul_nn_*)."""

    @staticmethod
    def forward(ctx, x, w):
        x2 = x.contiguous()
        y = hip.matmul_nn_fwd(x2, w.view(w.shape[0], -1))
        ctx.save_for_backward(x2)
        ctx.w = w
        return y

    @staticmethod
    def backward(ctx, g):
        x2, = ctx.saved_tensors
        w = ctx.w
        need_w = ctx.needs_input_grad[1]
        ow = _gout(w) if need_w else None
        dx, dw = hip.matmul_nn_bwd(x2, w.view(w.shape[0], -1), g.contiguous(), need_dx=ctx.needs_input_grad[0],
                                   out_dw=ow.view(w.shape[0], -1) if ow is not None else None, need_dw=need_w)
        if dw is not None and ow is None:
            dw = dw.view(w.shape)
        return dx, (_ret(ow, dw) if need_w else None)


@ctx_bound
class TwoHeadLinearFn(torch.autograd.Function):
    """(F.linear(h, w1, b1) | F.linear(h, w2, b2)) side by side in one [M, N1 + N2] buffer -- the (mean | pre-sigma) heads of
    the sylvester posterior (VAE.py:118-121) in the layout the posterior kernel reads (gx_latent_posterior_*): both dense
    launches write / read their columns of the shared buffer through row strides, the second backward adds its dh."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2):
        h2 = h.contiguous()
        n1, n2 = w1.shape[0], w2.shape[0]
        out = torch.empty(h2.shape[0], n1 + n2, device=h2.device)
        hip.linear_fwd(h2, w1, b1, None, out=out[:, :n1])
        hip.linear_fwd(h2, w2, b2, None, out=out[:, n1:])
        ctx.save_for_backward(h2)
        ctx.params = (w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, g):
        h2, = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        g = g.contiguous()
        n1 = w1.shape[0]
        o1w, o1b, o2w, o2b = _gout(w1), _gout(b1), _gout(w2), _gout(b2)
        if o1w is None or o1b is None:
            o1w = o1b = None
        if o2w is None or o2b is None:
            o2w = o2b = None
        dh, dw1, db1 = hip.linear_bwd(h2, w1, None, g[:, :n1], None, out_dw=o1w, out_db=o1b)
        _, dw2, db2 = hip.linear_bwd(h2, w2, None, g[:, n1:], None, out_dw=o2w, out_db=o2b, accumulate_dx=dh)
        return dh, _ret(o1w, dw1), _ret(o1b, db1), _ret(o2w, dw2), _ret(o2b, db2)


@ctx_bound
class LogSoftmaxKFn(torch.autograd.Function):
    """dec [K*B, C, H, W] -> log_m_r [K,B,1,H,W] = log_softmax over the K slots of the last channel (MONet.get_mask_recon_stack,
    models/monet_config.py:137-139)."""

    @staticmethod
    def forward(ctx, dec, K):
        dec = dec.contiguous()
        out = hip.logsoftmax_k_fwd(dec, K)
        ctx.save_for_backward(out)
        ctx.C = dec.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        return hip.logsoftmax_k_bwd(out, g.contiguous(), ctx.C), None


@ctx_bound
class LSTMFn(torch.autograd.Function):
    """nn.LSTM (one layer, zero initial state) over x [T,B,D] -> h [T,B,H]: input projection for all steps on the
    dense kernel, then one fused (recurrent GEMM + cell update) launch per step; backward mirrors it
    (models/genesis_config.py:297-307 prior_lstm)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        x = x.contiguous()
        T, B, D = x.shape
        H = w_hh.shape[1]
        dev = x.device
        gx = hip.linear_fwd(x.view(T * B, D), w_ih, b_ih)                    # [T*B, 4H]
        act = torch.empty(T, B, 4 * H, device=dev)
        c = torch.empty(T, B, H, device=dev)
        h = torch.empty(T, B, H, device=dev)
        gx3 = gx.view(T, B, 4 * H)
        if 1 < T <= hip.lstm_seq_steps(B, H):
            hip.lstm_seq_fwd(gx3, w_hh, b_hh, act, c, h)
        else:
            for t in range(T):
                hip.lstm_step_fwd(gx3[t], h[t - 1] if t else None, c[t - 1] if t else None, w_hh, b_hh, act[t], c[t], h[t])
        ctx.save_for_backward(x, act, c, h)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        return h

    @staticmethod
    def backward(ctx, g):
        x, act, c, h = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh = ctx.params
        T, B, D = x.shape
        H = w_hh.shape[1]
        g = g.contiguous()
        dgates = torch.empty(T, B, 4 * H, device=x.device)
        dc = torch.empty(2, B, H, device=x.device)
        if 1 < T <= hip.lstm_seq_steps(B, H):
            hip.lstm_seq_bwd(g, w_hh, act, c, dgates, dc)
        else:
            for t in reversed(range(T)):
                hip.lstm_step_bwd(g[t], dgates[t + 1] if t + 1 < T else None, w_hh, act[t], c[t],
                                  c[t - 1] if t else None, dc[(t + 1) & 1] if t + 1 < T else None, dgates[t], dc[t & 1])
        o_wih, o_whh, o_bih, o_bhh = _gout(w_ih), _gout(w_hh), _gout(b_ih), _gout(b_hh)
        dg2 = dgates.view(T * B, 4 * H)
        dx, dw_ih, db = hip.linear_bwd(x.view(T * B, D), w_ih, None, dg2, None, need_dx=ctx.needs_input_grad[0],
                                       out_dw=o_wih, out_db=o_bih)
        if T > 1:
            _, dw_hh, _ = hip.linear_bwd(h[:-1].view((T - 1) * B, H), w_hh, None, dgates[1:].view((T - 1) * B, 4 * H),
                                         None, need_dx=False, need_db=False, out_dw=o_whh)
        else:
            dw_hh = torch.zeros_like(w_hh) if o_whh is None else o_whh.zero_()
        if o_bhh is not None:
            o_bhh.copy_(db)
        return (dx.view(T, B, D) if dx is not None else None, _ret(o_wih, dw_ih), _ret(o_whh, dw_hh), _ret(o_bih, db),
                _ret(o_bhh, db))


@ctx_bound
class ARPriorKLFn(torch.autograd.Function):
    """The whole autoregressive-prior KL term of GENESIS-V2 as ONE autograd node (models/genesis_config.py:297-331 with
    prior_lstm / prior_linear): z [K,B,D], log_q [K,B] -> kl [K,B] = log_q - log p(z_k | z_<k).  Same launches as
    LSTMFn -> LinearFn -> PriorLogPFn chained, but z's uses inside the term (LSTM input z[:-1], the Gaussian's
    argument) no longer meet in autograd: the LSTM's input gradient is ADDED to the log-density's dz by the dense kernel
    (no slice-backward fill + copy, no accumulation launch), and b_ih / b_hh receive their common gradient from one
    launch."""

    @staticmethod
    def forward(ctx, z, log_q, w_ih, w_hh, b_ih, b_hh, w_lin, b_lin, want_lin=False):
        """want_lin: also return prior_linear's output [K-1,B,2D] (not differentiable: the prior's mean / scale for the
        returned statistics)."""
        z = z.contiguous()
        log_q = log_q.contiguous()
        K, B, D = z.shape
        T, H = K - 1, w_hh.shape[1]
        dev = z.device
        gx3 = hip.linear_fwd(z.view(K * B, D)[:T * B], w_ih, b_ih).view(T, B, 4 * H)
        act = torch.empty(T, B, 4 * H, device=dev)
        c = torch.empty(T, B, H, device=dev)
        h = torch.empty(T, B, H, device=dev)
        if 1 < T <= hip.lstm_seq_steps(B, H):
            hip.lstm_seq_fwd(gx3, w_hh, b_hh, act, c, h)                # (all steps in one launch)
        else:
            for t in range(T):
                hip.lstm_step_fwd(gx3[t], h[t - 1] if t else None, c[t - 1] if t else None, w_hh, b_hh, act[t], c[t], h[t])
        lin = hip.linear_fwd(h.view(T * B, H), w_lin, b_lin).view(T, B, -1)
        kl = hip.latent_prior_logp_fwd(z, lin, log_q)
        ctx.save_for_backward(z, act, c, h, lin)
        ctx.params = (w_ih, w_hh, b_ih, b_hh, w_lin, b_lin)
        if want_lin:
            lin_out = lin.detach()
            ctx.mark_non_differentiable(lin_out)
            ctx.set_materialize_grads(False)  # (no zero-filled gradient tensor for `lin`)
            return kl, lin_out
        return kl

    @staticmethod
    def backward(ctx, g, _g_lin=None):
        z, act, c, h, lin = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh, w_lin, b_lin = ctx.params
        K, B, D = z.shape
        T, H = K - 1, w_hh.shape[1]
        dev = z.device
        g = g.contiguous()
        dz, dlin = hip.latent_prior_logp_bwd(z, lin, g, True)
        o_wl, o_bl = _gout(w_lin), _gout(b_lin)
        dh, dw_lin, db_lin = hip.linear_bwd(h.view(T * B, H), w_lin, None, dlin.view(T * B, -1), None, out_dw=o_wl,
                                            out_db=o_bl)
        dh = dh.view(T, B, H)
        dgates = torch.empty(T, B, 4 * H, device=dev)
        dc = torch.empty(2, B, H, device=dev)
        if 1 < T <= hip.lstm_seq_steps(B, H):
            hip.lstm_seq_bwd(dh, w_hh, act, c, dgates, dc)
        else:
            for t in reversed(range(T)):
                hip.lstm_step_bwd(dh[t], dgates[t + 1] if t + 1 < T else None, w_hh, act[t], c[t],
                                  c[t - 1] if t else None, dc[(t + 1) & 1] if t + 1 < T else None, dgates[t], dc[t & 1])
        o_wih, o_whh, o_bih, o_bhh = _gout(w_ih), _gout(w_hh), _gout(b_ih), _gout(b_hh)
        _, dw_ih, db = hip.linear_bwd(z.view(K * B, D)[:T * B], w_ih, None, dgates.view(T * B, 4 * H), None,
                                      out_dw=o_wih, out_db=o_bih, out_db2=o_bhh,
                                      accumulate_dx=dz.view(K * B, D)[:T * B])
        if T > 1:
            _, dw_hh, _ = hip.linear_bwd(h[:-1].view((T - 1) * B, H), w_hh, None, dgates[1:].view((T - 1) * B, 4 * H),
                                         None, need_dx=False, need_db=False, out_dw=o_whh)
        else:
            dw_hh = torch.zeros_like(w_hh) if o_whh is None else o_whh.zero_()
        return (dz, g, _ret(o_wih, dw_ih), _ret(o_whh, dw_hh), _ret(o_bih, db), _ret(o_bhh, db), _ret(o_wl, dw_lin),
                _ret(o_bl, db_lin), None)
