"""Data-parallel plumbing of the training step (one process per GPU).

Every GENESIS-V2 op is per-image independent, so the batch shards over ranks with ONE exchange per
step: a sum all-reduce (RCCL over xGMI; `nccl` backend) of a flat fp32 gradient bucket whose tail
carries the rank's batch-mean reconstruction error and KL, so that after the all-reduce every rank
holds the global gradient AND the global err / KL and applies the identical GECO + Adam update.
The reference has no equivalent (only single-process nn.DataParallel, train.py:153-155).

Device-agnostic on purpose: the same class runs under `gloo` on CPU in the world_size-2 tests."""
import ctypes
import os

import torch
import torch.distributed as dist


class CabiAllReduce(object):
    """The bucket's collective through the library's own C-ABI entry points (gx_allreduce_*, include/genesis_hip.h: RCCL
    resolved by the library, no torch.distributed in the data path) -- what a host that is not PyTorch binds.  The 128-byte
    RCCL unique id travels from rank 0 to the others over the process group that is already up (any backend: it is host
    bytes); after that the group is not used for gradients.  GENESIS_CABI_ALLREDUCE=1 selects it in FlatBucket.all_reduce."""

    def __init__(self, group=None, device=None):
        from . import _lib
        self._lib = _lib
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        nbytes = int(_lib.load().gx_allreduce_unique_id_bytes())
        ident = ctypes.create_string_buffer(nbytes)
        if rank == 0:
            _lib.call('gx_allreduce_unique_id', ident, nbytes)
        if world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = ctypes.create_string_buffer(box[0], nbytes)
        if device is not None:
            torch.cuda.set_device(device)
        h = ctypes.c_void_p()
        _lib.call('gx_allreduce_init', ident, nbytes, rank, world, ctypes.byref(h))
        self.handle, self.rank, self.world = h, rank, world

    def run(self, flat):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        self._lib.call('gx_allreduce_run', self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.call('gx_allreduce_destroy', self.handle)
            self.handle = None


class FlatBucket(object):
    """Re-homes parameters (and their .grad) as views of flat per-dtype buffers.

    fp32 params -> flat_p / flat_g (+ `n_tail` piggy-backed scalars at the end of flat_g);
    fp64 params (att_process.log_sigma) -> flat_p64 / flat_g64; for the exchange their gradients ride in the fp32
    bucket's tail as (hi, mid, lo) float triples (g = hi + mid + lo exactly: 24 + 24 + 5 mantissa bits), so that a step
    has ONE collective -- exact for a single contribution; the sum over ranks is an fp32 sum of the hi parts (~6e-8
    relative on that gradient, which was computed from fp32 activations in the first place)."""

    def __init__(self, params, n_tail=2, align=16, mean_buffers=()):
        params = list(params)
        self.p32 = [p for p in params if p.dtype == torch.float32]
        self.p64 = [p for p in params if p.dtype == torch.float64]
        assert len(self.p32) + len(self.p64) == len(params), 'only fp32 / fp64 parameters are supported'
        dev = params[0].device
        # every parameter starts on a 64-byte boundary of the flat buffer (16-byte vector loads in the dense / LSTM
        # kernels need aligned rows); the padding stays zero in parameters, gradients and Adam state
        self.align = align
        self.slot = {}
        self.n32 = sum(self._pad(p.numel()) for p in self.p32)
        self.n64 = sum(p.numel() for p in self.p64)
        self.n_scalars = n_tail                    # err, kl
        # floating-point module buffers that every rank updates from its own shard (GENESIS' BatchNorm running
        # statistics, genesis_config.py:39-40): averaged over ranks through the same collective
        self.mean_buffers = [b for b in mean_buffers if b.is_floating_point()]
        self.n_buf = sum(b.numel() for b in self.mean_buffers)
        n_tail = n_tail + 3 * self.n64 + self.n_buf     # + (hi, mid, lo) per fp64 gradient element + the buffers
        self.n_tail = n_tail
        self.flat_p = torch.zeros(self.n32, dtype=torch.float32, device=dev)
        self.flat_p64 = torch.zeros(max(self.n64, 1), dtype=torch.float64, device=dev)
        # both gradient buffers are views of ONE allocation: zero_grad() is a single fill
        nb32 = 4 * (self.n32 + n_tail)
        off64 = (nb32 + 63) // 64 * 64
        self._graw = torch.zeros(off64 + 8 * max(self.n64, 1), dtype=torch.uint8, device=dev)
        self.flat_g = self._graw[:nb32].view(torch.float32)
        self.flat_g64 = self._graw[off64:].view(torch.float64)
        for plist, fp, fg, padded in ((self.p32, self.flat_p, self.flat_g, True),
                                      (self.p64, self.flat_p64, self.flat_g64, False)):
            off = 0
            for p in plist:
                n = p.numel()
                self.slot[id(p)] = (not padded, off, n)        # (is fp64, offset in its flat buffer, numel)
                fp[off:off + n].copy_(p.data.reshape(-1))
                p.data = fp[off:off + n].view(p.shape)
                p.grad = fg[off:off + n].view(p.shape)
                off += self._pad(n) if padded else n

    def _pad(self, n):
        return (n + self.align - 1) // self.align * self.align

    def zero_grad(self):
        self._graw.zero_()

    def set_tail(self, *scalars):
        for i, s in enumerate(scalars):
            self.flat_g[self.n32 + i] = s.detach()

    def collective_needed(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            return False
        # (the env var keeps the collective in the step on a 1-GPU box, to exercise RCCL)
        return dist.get_world_size(group) > 1 or bool(os.environ.get('GENESIS_FORCE_ALLREDUCE'))

    def pack64(self):
        """fp64 gradients -> (hi, mid, lo) float triples in the fp32 tail; averaged buffers behind them (before the collective)."""
        o = self.n32 + self.n_scalars
        if self.n64:
            t = self.flat_g[o:o + 3 * self.n64].view(3, self.n64)
            g = self.flat_g64[:self.n64]
            hi = g.to(torch.float32)
            r = g - hi.double()
            mid = r.to(torch.float32)
            t[0].copy_(hi)
            t[1].copy_(mid)
            t[2].copy_((r - mid.double()).to(torch.float32))
        o += 3 * self.n64
        for b in self.mean_buffers:
            self.flat_g[o:o + b.numel()].copy_(b.reshape(-1))
            o += b.numel()

    def unpack64(self, scale=None):
        """(hi, mid, lo) sums -> fp64 gradient sums; buffer sums -> buffer means (after the collective)."""
        o = self.n32 + self.n_scalars
        if self.n64:
            t = self.flat_g[o:o + 3 * self.n64].view(3, self.n64)
            self.flat_g64[:self.n64].copy_(t[0].double() + t[1].double() + t[2].double())
        o += 3 * self.n64
        if self.mean_buffers:
            if scale is None:
                scale = 1.0 / dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1.0
            for b in self.mean_buffers:
                b.copy_((self.flat_g[o:o + b.numel()] * scale).view(b.shape))
                o += b.numel()

    def _sum(self, flat, group):
        if os.environ.get('GENESIS_CABI_ALLREDUCE') and flat.is_cuda:
            if getattr(self, '_cabi', None) is None:
                self._cabi = CabiAllReduce(group, flat.device)
            self._cabi.run(flat)
        else:
            dist.all_reduce(flat, group=group)

    def all_reduce_range(self, lo, hi, group=None):
        """Sum over ranks of flat_g[lo:hi] alone (the part of the bucket that is final early: TrainStep's early flush)."""
        self._sum(self.flat_g[lo:hi], group)

    def all_reduce(self, group=None, packed=False, done=None):
        """Sum over ranks -- ONE collective for parameters' gradients, the err / kl scalars and the fp64 gradients;
        returns the scale (1/world) that turns the sums into means.  packed: the caller already ran pack64() and will
        run unpack64() itself (the graph-replay path captures them with the neighbouring kernels).  done = (lo, hi): that
        range was already summed by all_reduce_range -- only the rest travels now."""
        if not self.collective_needed(group):
            return 1.0
        world = dist.get_world_size(group)
        if not packed:
            self.pack64()
        if done is None:
            self._sum(self.flat_g, group)
        else:
            lo, hi = done
            if lo > 0:
                self._sum(self.flat_g[:lo], group)
            if hi < self.flat_g.numel():
                self._sum(self.flat_g[hi:], group)
        if not packed:
            self.unpack64(1.0 / world)
        return 1.0 / world

    def param_range(self, params):
        """(lo, hi) of flat_g covered by `params` if they are fp32 and lie back to back in the bucket, else None."""
        slots = sorted(self.slot[id(p)] for p in params)
        if not slots or any(s[0] for s in slots):
            return None
        lo, hi = slots[0][1], slots[-1][1] + self._pad(slots[-1][2])
        return (lo, hi) if sum(self._pad(n) for _, _, n in slots) == hi - lo else None

    def broadcast_state(self, extra=(), group=None, src=0):
        """Parameters (and any `extra` tensors: optimiser moments, step counter, GECO state, model buffers) from rank
        `src` to every rank -- what DistributedDataParallel does at construction: ranks whose seed or checkpoint differs
        would otherwise train different models while exchanging gradients."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        for t in [self.flat_p, self.flat_p64] + list(extra):
            dist.broadcast(t, src, group=group)

    def tail(self, scale=1.0):
        return self.flat_g[self.n32:self.n32 + self.n_scalars] * scale

    def grads_in_bucket(self):
        lo, hi = self.flat_g.data_ptr(), self.flat_g.data_ptr() + self.flat_g.numel() * 4
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.p32)
