"""One training iteration of the reference's loop (train.py:215-263: zero_grad, forward, loss aggregation,
GECO, backward, Adam) as a device-resident step: flat parameter / gradient / Adam-state buffers, a fused
HIP Adam kernel, the on-device GECO update, optional HIP-graph replay of the whole step, and -- with one
process per GPU -- a single RCCL all-reduce of the flat gradient bucket (the batch-mean err / KL ride in
its tail so every rank applies the identical GECO update)."""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from . import functions as _fn
from . import hip_ops as _hip
from .dp import FlatBucket
from .geco import make_geco


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class TrainStep(object):

    OPTIMISERS = ('adam', 'rmsprop', 'sgd')       # train.py:170-176 (config.optimiser)

    def __init__(self, model, img_size, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, geco=None, use_geco=True,
                 beta_fixed=0.5, process_group=None, graph=False, async_wgrad=False, weight_cache=True, defer_reduces=True,
                 side_prior=None, optimiser='adam', beta_warmup=False, train_iter=None, log_mse=False,
                 materialise_stats=False):
        """optimiser: 'adam' (torch.optim.Adam(lr); betas / eps as given), 'rmsprop' (torch.optim.RMSprop(lr): alpha 0.99,
        eps 1e-8) or 'sgd' (torch.optim.SGD(lr, 0.9)) -- train.py:170-176.  use_geco=False: the fixed-beta objective
        err + beta_fixed * kl, with beta_warmup the linear ramp beta_fixed * iter / (0.2 * train_iter) of train.py:252-258.
        log_mse: step() also returns train.py:244-246's (mse, rmse) behind (elbo, err, kl, beta).
        materialise_stats: evaluate, inside every step, the outputs of the forward that a training iteration never reads and
        this build therefore computes on first access (stats.mx_r_k / instance_seg / instance_seg_r,
        genesisv2_config.py:184-188; att_stats.delta) -- the reference's forward computes them unconditionally; a measurement
        switch (bench.py: value_as_written), nothing consumes the values."""
        if optimiser not in self.OPTIMISERS:
            raise ValueError('optimiser must be one of %s (train.py:170-176), got %r' % (self.OPTIMISERS, optimiser))
        if beta_warmup and (use_geco and geco is None or geco is not None):
            raise ValueError('beta_warmup belongs to the fixed-beta objective (use_geco=False), train.py:249-259')
        if beta_warmup and not train_iter:
            raise ValueError('beta_warmup needs train_iter (the ramp covers 0.2 * train_iter iterations)')
        self.model = model
        self.optimiser = optimiser
        self.beta_warmup, self.train_iter, self.log_mse = bool(beta_warmup), train_iter, bool(log_mse)
        self.materialise_stats = bool(materialise_stats)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.device = next(model.parameters()).device
        if self.device.type != 'cuda':
            raise _lib.GenesisHipError('TrainStep needs the model on a HIP device; there is no CPU path')
        self.geco = geco if geco is not None else (make_geco(img_size, device=self.device) if use_geco else None)
        self.beta_fixed = beta_fixed
        self.async_wgrad = async_wgrad
        self.side_prior = (os.environ.get('GENESIS_SIDE_PRIOR', '0') == '1') if side_prior is None else side_prior
        self.defer_reduces = defer_reduces
        self._beta_fixed_t = torch.tensor(float(beta_fixed), device=self.device)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self._flatten()
        # this loop's library context: its deferred-reduction queues, queued weight-gradient jobs, packed-weight cache
        # bookkeeping and step flags are its own (two TrainSteps can coexist in one process)
        self._ctx = int(_lib.load().gx_ctx_create())
        if self._ctx <= 0:
            raise _lib.GenesisHipError('gx_ctx_create failed: ' + _lib.last_error())
        self._wcache = _lib.query('gx_weight_cache_create') if weight_cache else None
        self._wcache_ready = False
        self.step_t = torch.zeros((), dtype=torch.int64, device=self.device)
        self._mse_ws = self._mse_out = None
        # the step's noise (rand_pixel, eps) from one counter-based launch keyed by (torch's seed + rank, the step counter)
        # instead of torch.rand + torch.randn and their graph-RNG bookkeeping (GENESIS_HIP_NOISE=0: torch's generators)
        # The hook is on the model only WHILE an iteration of this loop runs (_enter .. _leave): validation / visualisation
        # forwards between two steps draw fresh torch.rand / randn like the reference's (they would otherwise all see the
        # next training step's noise), torch.save(model) keeps working, and two loops on one model cannot clear each
        # other's hook.
        self._hip_noise = os.environ.get('GENESIS_HIP_NOISE', '1') != '0' and hasattr(model, 'noise')
        self._noise_hook = None           # (set only inside an iteration: no reference cycle self -> bound method -> self)
        self.graph = None
        self.graph2 = None
        self._split = False
        # GENESIS_WGQ_EARLY_FLUSH=1 (GENESIS-V2): a stream-K flush of its own for the decoder's weight gradients at the end of the
        # decoder's backward, and -- several ranks, eager or collective-in-graph launch modes -- that range of the bucket
        # all-reduced on a second stream while the encoder's backward runs ("as soon as the last grad is
        # written, overlapped"); the rest of the bucket follows at the end of the backward as before
        dec = getattr(model, 'decoder_module', None)          # (GenesisV2: ConvTranspose + GroupNorm stack + the 1x1 head)
        self._early_range = None
        if os.environ.get('GENESIS_WGQ_EARLY_FLUSH') == '1' and isinstance(dec, torch.nn.Module):
            self._early_range = self.bucket.param_range(list(dec.parameters()))
        self._early_side = None
        self._early_done = False
        self._early_collective_ok = True      # (False while capturing the two-graph form: no collective inside those graphs)
        self.use_graph = graph
        # GENESIS_SYNC_BN=1, several ranks, a model with BatchNorm (GENESIS v1 / BaselineVAE: genesis_config.py:39-40): batch
        # statistics over ALL ranks' shards (genesis_amd/sylvester.sync_bn) -- the single-device reference at the global batch;
        # default: per-replica statistics, like the reference's own nn.DataParallel.  Its small per-layer collectives sit
        # inside forward and backward, so the step is issued eagerly (no HIP graph around collectives in the split form).
        self._sync_bn = (os.environ.get('GENESIS_SYNC_BN') == '1' and self.world > 1 and
                         any(isinstance(m, torch.nn.BatchNorm2d) for m in model.modules()))
        if self._sync_bn:
            self.use_graph = False
        self._static_x = None
        self._out = None
        self.iters = 0
        self.sync_from_rank0()

    # ------------------------------------------------------------------ flat buffers
    def _flatten(self):
        # tail scalars of the exchange: the rank's batch-mean err, kl [, mse, rmse with log_mse: train.py:244-246 logs them over
        # the WHOLE batch, so with several ranks they ride the same all-reduce as err / kl -- advisor finding, round 4]
        self.bucket = FlatBucket(self.model.parameters(), n_tail=4 if self.log_mse else 2,
                                 mean_buffers=list(self.model.buffers()) if self.world > 1 else ())
        b = self.bucket
        self.n32, self.n64 = b.n32, b.n64
        self.flat_p, self.flat_g, self.flat_p64, self.flat_g64 = b.flat_p, b.flat_g, b.flat_p64, b.flat_g64
        self.m32, self.v32 = torch.zeros_like(b.flat_p), torch.zeros_like(b.flat_p)
        self.m64, self.v64 = torch.zeros_like(b.flat_p64), torch.zeros_like(b.flat_p64)

    def _train_state(self):
        """Everything an iteration mutates: parameters, Adam moments, step counter, GECO state, model buffers
        (GENESIS' BatchNorm running statistics)."""
        st = [self.flat_p, self.flat_p64, self.m32, self.v32, self.m64, self.v64, self.step_t]
        if self.geco is not None:
            st.append(self.geco.state)
        return st + list(self.model.buffers())

    def sync_from_rank0(self):
        """Several ranks: every rank starts from rank 0's parameters, optimiser and GECO state and buffers (as
        DistributedDataParallel broadcasts at construction).  Ranks then only need DIFFERENT noise seeds
        (torch.manual_seed(base + rank) after building the model) so that rand_pixel / eps differ per shard."""
        if self.world > 1:
            self.bucket.broadcast_state(self._train_state()[2:], self.pg)

    def _check_grad_views(self):
        assert self.bucket.grads_in_bucket(), 'a gradient left the flat bucket'

    # ------------------------------------------------------------------ one iteration
    def _enter(self):
        """Makes this loop's library context current and arms its step switches."""
        self._prev_ctx = _lib.current_ctx()
        _lib.make_current(self._ctx)
        st = _fn.step_state()
        st.direct_param_grads = True     # the bucket is zeroed first; kernels write weight grads straight into it
        _fn.begin_direct_grads()
        _hip.defer_state().on = self.defer_reduces
        st.async_wgrad = self.async_wgrad
        st.side_prior = self.side_prior
        st.early_flush = self._early_collective if self._early_range is not None else None
        if self._sync_bn:
            from . import sylvester
            sylvester.sync_bn(self.pg, True)
        if self._hip_noise:
            self._noise_prev = self.model.__dict__.get('noise')
            self._noise_hook = self._draw_noise
            self.model.noise = self._noise_hook

    def _leave(self):
        if self._sync_bn:
            from . import sylvester
            sylvester.sync_bn(None, False)
        if self._noise_hook is not None and self.model.__dict__.get('noise') is self._noise_hook:
            if getattr(self, '_noise_prev', None) is not None:
                self.model.noise = self._noise_prev
            else:
                del self.model.noise            # back to the class attribute (None): torch's generators
        self._noise_prev = self._noise_hook = None
        _lib.make_current(self._ctx)     # (an exception may have left another context current)
        st = _fn.step_state()
        st.direct_param_grads = False
        st.async_wgrad = False
        st.side_prior = False
        st.early_flush = None
        _hip.defer_state().on = False
        _hip.defer_discard()             # no-op after a completed iteration (the queue was flushed)

    def _zero_grads(self):
        """The bucket must be clean before a backward pass accumulates into it.  After a completed iteration it is: the
        fused Adam launch zeroes the gradients it consumes (no separate fill launch per step)."""
        if not getattr(self, '_grads_clean', False):
            self.bucket.zero_grad()

    def _iteration(self, x, **forward_kwargs):
        self._zero_grads()
        self._enter()
        # packed-weight cache: the first (never graph-captured) iteration records which weight tensors the conv
        # entry points pack; later iterations re-pack all of them in one launch up front
        recording = False
        if self._wcache is not None:
            if not self._wcache_ready:
                if not torch.cuda.is_current_stream_capturing():
                    _lib.call('gx_weight_cache_record', self._wcache, 1)
                    recording = True
            else:
                _lib.call('gx_weight_cache_refresh', self._wcache,
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        try:
            return self._iteration_body(x, **forward_kwargs)
        finally:
            self._leave()
            if self._wcache is not None:
                if recording:
                    _lib.call('gx_weight_cache_record', self._wcache, 0)
                    self._wcache_ready = True
                else:
                    _lib.call('gx_weight_cache_release')
            _lib.make_current(self._prev_ctx)

    def _draw_noise(self, uniform_shape, normal_shape, device):
        rank = dist.get_rank(self.pg) if (self.world > 1 and dist.is_initialized()) else 0
        seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1)) & 0xFFFFFFFFFFFFFFFF
        return _hip.philox_noise(uniform_shape, normal_shape, seed, self.step_t, device)

    def close(self):
        """Releases the library context (at most 31 live ones), the packed-weight cache and the per-context Python
        state.  Idempotent; __del__ calls it, but a loop that builds many TrainSteps should call it itself (a
        reference cycle or a stored traceback can keep the object -- and its context -- alive)."""
        if getattr(self, '_wcache', None) is not None:
            _lib.call('gx_weight_cache_destroy', self._wcache)
            self._wcache = None
        self._noise_hook = None
        ctx = getattr(self, '_ctx', 0)
        if ctx > 0:
            self._ctx = 0
            self.graph = self.graph2 = None
            _fn.drop_ctx_state(ctx)
            _lib.call('gx_ctx_destroy', ctx)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _early_collective(self):
        """DecoderFn's backward is complete and its queued weight gradients have been flushed: the decoder's range of the
        bucket is final.  Several ranks: start its all-reduce on the second stream (a parallel branch when captured)."""
        if not (self._early_collective_ok and self.bucket.collective_needed(self.pg)):
            return
        if torch.cuda.is_current_stream_capturing():
            # no collective on a forked stream INSIDE the captured step: that form existed behind an opt-in in round 5 (it replayed
            # bit-identically to the eager form), but one of two full GPU-suite runs aborted inside hipStreamEndCapture of exactly
            # that capture and the abort was never reproduced in isolation -- an unexplained abort next to the default multi-rank
            # path is worse than the ~40 us the fork could hide, so the path was deleted in round 6 (review item 9a).  The
            # captured step keeps the early flush and sends the bucket in one piece; the EAGER step overlaps as described.
            return
        if self._early_side is None:
            self._early_side = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self._early_side.wait_stream(cur)
        with torch.cuda.stream(self._early_side), torch.no_grad():
            self.bucket.all_reduce_range(self._early_range[0], self._early_range[1], self.pg)
        self._early_done = True

    def _final_all_reduce(self, packed=False):
        """The collective at the end of the backward: the whole bucket, or what the early one left."""
        done = None
        if self._early_done:
            torch.cuda.current_stream().wait_stream(self._early_side)
            done, self._early_done = self._early_range, False
        return self.bucket.all_reduce(self.pg, packed=packed, done=done)

    def _iteration_body(self, x, **forward_kwargs):
        st = self._forward_backward(x, **forward_kwargs)
        with torch.no_grad():
            gscale = self._final_all_reduce()
            return self._update(st, gscale)

    def _forward_backward(self, x, **forward_kwargs):
        """zero-grad'ed bucket -> forward -> loss -> backward; leaves (err, kl) batch means in the bucket tail."""
        self._grads_clean = False
        self._early_done = False
        if self.beta_warmup:       # this iteration's beta from the device step counter (= the iteration index, train.py:254)
            _lib.call('gx_beta_warmup', _p(self.step_t), float(self.beta_fixed), 0.2 * float(self.train_iter),
                      _p(self._beta_fixed_t), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        recon, losses, stats, att_stats, comp_stats = self.model(x, **forward_kwargs)
        if self.materialise_stats:
            with torch.no_grad():
                self._materialised = [d_._resolve_all() or d_ for d_ in (stats, att_stats) if hasattr(d_, '_resolve_all')]
        if self.log_mse:
            with torch.no_grad():
                xr, rr = x.detach().contiguous(), recon.detach().contiguous()
                if self._mse_ws is None or self._mse_ws.numel() < xr.shape[0] + 4:
                    self._mse_ws = torch.zeros(xr.shape[0] + 4, device=self.device)
                    self._mse_out = torch.zeros(2, device=self.device)
                _lib.call('gx_mse_rmse', _p(xr), _p(rr), xr.shape[0], xr[0].numel(), _p(self._mse_out), _p(self._mse_ws),
                          self._mse_ws.numel() * 4, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                if self.bucket.collective_needed(self.pg):
                    self.bucket.flat_g[self.n32 + 2:self.n32 + 4].copy_(self._mse_out)
        # loss aggregation of train.py:226-242 (every model family: kl_m | kl_m_k, kl_l | kl_l_k)
        beta_t = self.geco.state[0:1] if self.geco is not None else self._beta_fixed_t.view(1)
        # (if / elif per stage like the reference: a model returning both forms of a term must not be counted twice)
        keys = [k for k in (('kl_m' if 'kl_m' in losses else 'kl_m_k'), ('kl_l' if 'kl_l' in losses else 'kl_l_k'))
                if k in losses]
        # every KL term as rows [R_i, B]: sum_k mean_b is the same for the rows of all terms stacked on top of each other,
        # so one ElboFn launch serves MONet (kl_m + kl_l_k) and GENESIS (kl_m_k + kl_l_k) too
        fused, rows = losses.err.dim() == 1, []
        for k in keys:
            v = dict.__getitem__(losses, k)
            if k.endswith('_k'):
                v = getattr(v, 'stacked', None)              # [K,B] tensor the per-slot list was unbound from
                if v is None:
                    fused = False
                    break
            if v.dim() == 0 or v.numel() % losses.err.shape[0] != 0:
                fused = False                                # (a scalar / oddly shaped KL: the plain aggregation below)
                break
            rows.append(v.reshape(-1, losses.err.shape[0]))
        kl_rows = None
        if fused and rows:
            kl_rows = rows[0] if len(rows) == 1 else torch.cat(rows, 0)
        if fused:
            # one launch: batch means, the GECO-weighted objective, and (err, kl) straight into the bucket tail
            # (the objective is linear in err / kl: its gradients 1 / B and beta / B come out of the same launch and the
            #  backward pass starts from them -- no autograd node, no second launch)
            err_c = losses.err.contiguous()
            kl_c = kl_rows.contiguous() if kl_rows is not None else None
            with torch.no_grad():
                out5, d_err, d_kl = _hip.elbo_fwd_grads(err_c, kl_c, beta_t, self.bucket.flat_g[self.n32:self.n32 + 2])
            roots = [(t, g) for t, g in ((err_c, d_err), (kl_c, d_kl)) if t is not None and t.requires_grad]
            if roots:
                torch.autograd.backward([t for t, _ in roots], [g for _, g in roots])
            beta_used = out5.detach()
        else:
            err = losses.err.mean(0)
            kl = err.new_zeros(())
            if 'kl_m' in losses:
                kl = kl + losses.kl_m.mean(0)
            elif 'kl_m_k' in losses:
                kl = kl + torch.stack(losses.kl_m_k, dim=1).mean(dim=0).sum()
            if 'kl_l' in losses:
                kl = kl + losses.kl_l.mean(0)
            elif 'kl_l_k' in losses:
                kl = kl + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
            beta = beta_t[0].clone()
            loss = err + beta * kl
            loss.backward()
            with torch.no_grad():
                self.bucket.set_tail(err, kl)
            beta_used = beta.detach()
        _fn.join_side_stream()     # weight-gradient kernels forked onto the side stream
        _hip.defer_flush()         # all queued weight-gradient / GroupNorm-affine reductions: one launch per kind
        return fused, beta_used

    def _update(self, st, gscale):
        """(all-reduced) bucket -> device GECO update -> fused Adam; returns [elbo, err, kl, beta used]."""
        fused, beta_used = st
        local = fused and gscale == 1.0
        tail = self.bucket.flat_g[self.n32:self.n32 + 2] if gscale == 1.0 else self.bucket.tail(gscale)   # global batch means
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.geco is not None:
            self.geco.update(tail[0], step=self.step_t)      # GECO multiplier and the optimiser's step counter: one launch
        else:
            _lib.call('gx_step_increment', _p(self.step_t), stream)
        # fp32 and fp64 parameter groups in one launch; it zeroes the gradients it consumed
        if self.optimiser == 'adam':
            _lib.call('gx_adam_step_pair', _p(self.flat_p), _p(self.flat_g), _p(self.m32), _p(self.v32), self.n32,
                      _p(self.flat_p64) if self.n64 else None, _p(self.flat_g64) if self.n64 else None,
                      _p(self.m64) if self.n64 else None, _p(self.v64) if self.n64 else None, self.n64,
                      _p(self.step_t), self.lr, self.betas[0], self.betas[1], self.eps, gscale, 1, stream)
        else:
            # RMSprop(lr): alpha 0.99, eps 1e-8; SGD(lr, momentum 0.9) -- torch's defaults as train.py:170-176 constructs them
            kind, hp, eps = (1, 0.99, 1e-8) if self.optimiser == 'rmsprop' else (2, 0.9, 0.0)
            _lib.call('gx_optimiser_step_pair', kind, _p(self.flat_p), _p(self.flat_g), _p(self.m32), self.n32,
                      _p(self.flat_p64) if self.n64 else None, _p(self.flat_g64) if self.n64 else None,
                      _p(self.m64) if self.n64 else None, self.n64, _p(self.step_t), self.lr, hp, eps, gscale, 1, stream)
        self._grads_clean = True
        if local:
            out = beta_used[1:5]                             # ElboFn's (elbo, err, kl, beta used)
        else:
            bu = beta_used[4] if fused else beta_used
            out = torch.stack((tail[0] + tail[1], tail[0], tail[1], bu))
        if not self.log_mse:
            return out
        # (several ranks: the all-reduced sums of the ranks' shard means / world = the whole batch's mse, rmse)
        return torch.cat((out, self._mse_out if gscale == 1.0 else self.bucket.flat_g[self.n32 + 2:self.n32 + 4] * gscale))

    # ------------------------------------------------------------------ HIP-graph replay
    def _begin(self):
        self._zero_grads()
        self._enter()
        if self._wcache is not None and self._wcache_ready:
            _lib.call('gx_weight_cache_refresh', self._wcache,
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def _end(self):
        self._leave()
        if self._wcache is not None and self._wcache_ready:
            _lib.call('gx_weight_cache_release')
        _lib.make_current(self._prev_ctx)

    def _capture(self, x):
        """Warm up (kernel attributes, allocator pools, weight-cache recording), capture the iteration into HIP
        graphs, restore the pre-warm-up training state and replay once: the call is exactly one training step.
        One process: a single graph.  Several ranks: two graphs (forward+backward | GECO+Adam) with the RCCL
        all-reduce of the gradient bucket issued between the two replays -- nothing else runs on the host."""
        self._static_x = x.clone()
        state = self._train_state()
        snap = [t.clone() for t in state]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._out = self._iteration(self._static_x)
        torch.cuda.current_stream().wait_stream(side)
        self._split = self._needs_collective()
        self.collective_in_graph = False
        # (only RCCL enqueues on the captured stream; a gloo collective inside a capture invalidates it for good)
        # Several ranks: the default is the two-graph form below -- forward+backward | collective | GECO+Adam, three
        # enqueue-only calls per step, no host synchronisation, valid on any RCCL.  GENESIS_GRAPH_ALLREDUCE=1 asks for the
        # collective INSIDE the one graph (RCCL enqueues on the captured stream; saves two launch latencies): every rank
        # attempts the capture, then the ranks agree (all-reduce MIN of a success flag, outside any capture) -- the
        # in-graph form is used only if EVERY rank captured it, otherwise every rank takes the two-graph form, so the
        # ranks can never issue different collective sequences.
        want = os.environ.get('GENESIS_GRAPH_ALLREDUCE', '1' if self.world == 1 else '0') != '0'
        if self._split and want and (dist.get_backend(self.pg) == 'nccl' or os.environ.get('GENESIS_CABI_ALLREDUCE')):
            self._gscale = 1.0 / self.world
            g, ok = None, 1
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._begin()
                    try:
                        st = self._forward_backward(self._static_x)
                        with torch.no_grad():
                            self.bucket.pack64()
                            self._final_all_reduce(packed=True)
                            self.bucket.unpack64(self._gscale)
                            self._out = self._update(st, self._gscale)
                    finally:
                        self._end()
            except Exception as e:          # noqa: BLE001  (whatever the capture raised: the split form is always valid)
                self.capture_fallback_reason = '%s: %s' % (type(e).__name__, str(e)[:200])
                ok = 0
                torch.cuda.synchronize()
                self._grads_clean = False
            if self.world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
                if ok and int(flag.item()) == 0:
                    self.capture_fallback_reason = 'another rank could not capture the collective'
                ok = int(flag.item())
            if ok:
                self.graph, self._split, self.collective_in_graph = g, False, True
            else:
                del g
        if self.collective_in_graph:
            pass
        elif not self._split:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self._out = self._iteration(self._static_x)
        else:
            self._gscale = 1.0 / self.world
            self.graph = torch.cuda.CUDAGraph()
            self._early_collective_ok = False      # (the early flush stays; its collective needs the eager / in-graph forms)
            # thread_local: the process group's watchdog thread polls events while we capture
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self._begin()
                try:
                    self._st = self._forward_backward(self._static_x)
                    with torch.no_grad():
                        self.bucket.pack64()           # fp64 gradients into the fp32 tail: one collective per step
                finally:
                    self._end()
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph2, pool=self.graph.pool(), capture_error_mode='thread_local'):
                with torch.no_grad():
                    self.bucket.unpack64(self._gscale)
                    self._out = self._update(self._st, self._gscale)
        with torch.no_grad():
            for t, s in zip(state, snap):
                t.copy_(s)
        self._replay()
        self.iters += 1

    def prepare(self, x):
        """Builds everything a first step() would (weight-cache recording, kernel attributes, graph capture) WITHOUT
        advancing the training state: parameters, Adam moments, step counter and GECO state are restored afterwards."""
        if not self.use_graph or self.graph is not None:
            return
        state = self._train_state()
        snap = [t.clone() for t in state]
        iters = self.iters
        self._capture(x)
        torch.cuda.synchronize()
        with torch.no_grad():
            for t, s in zip(state, snap):
                t.copy_(s)
        self.iters = iters

    def _needs_collective(self):
        import os
        return dist.is_available() and dist.is_initialized() and \
            (self.world > 1 or bool(os.environ.get('GENESIS_FORCE_ALLREDUCE')))

    def _replay(self):
        # the captured graphs hold no bucket fill (the Adam launch zeroes what it consumed): if an eager iteration
        # died between its backward and its Adam launch, its partial gradients are still in the bucket -- clear them
        # here, or the replayed backward (+=) would accumulate onto them
        if not getattr(self, '_grads_clean', False):
            self.bucket.zero_grad()
            self._grads_clean = True
        self.graph.replay()
        if self._split:
            with torch.no_grad():
                self.bucket.all_reduce(self.pg, packed=True)      # the ONE collective of the step
            self.graph2.replay()

    def step(self, x, **forward_kwargs):
        """x [B,3,S,S] on the device.  Returns a device tensor [elbo, err, kl, beta_used] (no host sync).
        forward_kwargs (rand_pixel / eps / seed_idx injection, parity tests) force the eager path."""
        if self.use_graph and not forward_kwargs:
            if self.graph is None:
                self._capture(x)
                return self._out
            self._static_x.copy_(x)
            self._replay()
            self.iters += 1
            return self._out
        out = self._iteration(x, **forward_kwargs)
        if self.iters == 0:
            self._check_grad_views()
        self.iters += 1
        return out

    # ------------------------------------------------------------------ checkpoint (train.py:410-420 wire format)
    def _adam_slices(self, p):
        is64, off, n = self.bucket.slot[id(p)]
        m, v = (self.m64, self.v64) if is64 else (self.m32, self.v32)
        return m[off:off + n].view(p.shape), v[off:off + n].view(p.shape)

    def state_dict(self, iter_idx):
        """The reference's checkpoint dict (train.py:405-420): `optimiser_state_dict` is a genuine
        torch.optim.Adam state_dict over model.parameters() in order (exp_avg / exp_avg_sq / step per parameter), so
        a checkpoint written here resumes in the reference's loop (train.py:179-207) and vice versa."""
        params = list(self.model.parameters())
        step = int(self.step_t)
        if self.optimiser == 'adam':
            opt = torch.optim.Adam(params, self.lr, betas=self.betas, eps=self.eps)
        elif self.optimiser == 'rmsprop':
            opt = torch.optim.RMSprop(params, self.lr)
        else:
            opt = torch.optim.SGD(params, self.lr, 0.9)
        if step > 0:
            for p in params:
                m, v = self._adam_slices(p)
                if self.optimiser == 'adam':
                    opt.state[p] = {'step': torch.tensor(float(step)), 'exp_avg': m.clone(), 'exp_avg_sq': v.clone()}
                elif self.optimiser == 'rmsprop':
                    opt.state[p] = {'step': torch.tensor(float(step)), 'square_avg': m.clone()}
                else:
                    opt.state[p] = {'momentum_buffer': m.clone()}
        return {'model_state_dict': self.model.state_dict(),
                'optimiser_state_dict': opt.state_dict(),
                # (fixed-beta objective with the warm-up ramp: the ramp value of iteration `iter_idx`, the `beta` that
                #  train.py:252-258 computed in the iteration whose end save_checkpoint records)
                'beta': self.geco.beta.detach().clone() if self.geco is not None else self._current_fixed_beta(iter_idx),
                'err_ema': (self.geco.err_ema.detach().clone() if self.geco.err_ema is not None else None)
                if self.geco is not None else None,
                'iter_idx': iter_idx}

    def _current_fixed_beta(self, iter_idx):
        if not self.beta_warmup:
            return self.beta_fixed
        return min(self.beta_fixed, max(0.0, self.beta_fixed * iter_idx / (0.2 * float(self.train_iter))))

    def load_state_dict(self, ckpt):
        """Restores model, Adam moments / step, GECO state from a checkpoint of the reference's format; returns the
        iteration to continue from (train.py:207).  Parameters keep living in the flat bucket."""
        sd = dict(ckpt['model_state_dict'])
        sd.pop('comp_vae.decoder_module.seq.0.pixel_coords.g_1', None)     # legacy entries, train.py:191-192
        sd.pop('comp_vae.decoder_module.seq.0.pixel_coords.g_2', None)
        osd = ckpt['optimiser_state_dict']
        params = list(self.model.parameters())
        ids = [i for g in osd['param_groups'] for i in g['params']]
        if len(ids) != len(params):
            raise ValueError('optimiser state has %d parameters, the model %d' % (len(ids), len(params)))
        steps = set()
        # what the checkpoint holds is decided BEFORE anything is copied: a checkpoint of another optimiser must leave this
        # loop's moment buffers untouched (advisor finding, round 4)
        kinds = {('adam' if 'exp_avg' in st else 'rmsprop' if 'square_avg' in st else 'sgd')
                 for st in (osd['state'].get(i) for i in ids) if st is not None}
        if kinds - {self.optimiser}:
            raise ValueError('the checkpoint holds %s state, this loop runs %s' % ('/'.join(sorted(kinds)), self.optimiser))
        self.model.load_state_dict(sd)
        assert self.bucket.grads_in_bucket()
        with torch.no_grad():
            self.m32.zero_(); self.v32.zero_(); self.m64.zero_(); self.v64.zero_()
            for i, p in zip(ids, params):
                st = osd['state'].get(i)
                if st is None:
                    continue
                m, v = self._adam_slices(p)
                if 'exp_avg' in st:
                    m.copy_(st['exp_avg']); v.copy_(st['exp_avg_sq'])
                elif 'square_avg' in st:
                    m.copy_(st['square_avg'])
                elif st.get('momentum_buffer') is not None:
                    m.copy_(st['momentum_buffer'])
                if 'step' in st:
                    steps.add(int(st['step']))
            if len(steps) > 1:
                raise ValueError('per-parameter optimiser step counts differ: %s' % sorted(steps))
            # (torch.optim.SGD keeps no step count: the iteration index of the checkpoint stands in)
            self.step_t.fill_(steps.pop() if steps else (int(ckpt.get('iter_idx', -1)) + 1 if self.optimiser == 'sgd' else 0))
        g0 = osd['param_groups'][0]
        self.lr = g0['lr']
        if self.optimiser == 'adam':
            self.betas, self.eps = tuple(g0['betas']), g0['eps']
        if self.geco is not None:
            if ckpt.get('beta') is not None:
                self.geco.beta = ckpt['beta']
            if ckpt.get('err_ema') is not None:
                self.geco.err_ema = ckpt['err_ema']
        self.graph = self.graph2 = None          # re-capture: lr / betas are baked into the captured launches
        self.sync_from_rank0()                   # several ranks: rank 0's checkpoint wins
        return ckpt['iter_idx'] + 1
