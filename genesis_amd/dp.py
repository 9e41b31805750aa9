"""Data-parallel plumbing of the training step (one process per GPU).

Every GENESIS-V2 op is per-image independent, so the batch shards over ranks with ONE exchange per
step: a sum all-reduce (RCCL over xGMI; `nccl` backend) of a flat fp32 gradient bucket whose tail
carries the rank's batch-mean reconstruction error and KL, so that after the all-reduce every rank
holds the global gradient AND the global err / KL and applies the identical GECO + Adam update.
The reference has no equivalent (only single-process nn.DataParallel, train.py:153-155).

Device-agnostic on purpose: the same class runs under `gloo` on CPU in the world_size-2 tests."""
import os

import torch
import torch.distributed as dist


class FlatBucket(object):
    """Re-homes parameters (and their .grad) as views of flat per-dtype buffers.

    fp32 params -> flat_p / flat_g (+ `n_tail` piggy-backed scalars at the end of flat_g);
    fp64 params (att_process.log_sigma) -> flat_p64 / flat_g64 (a separate 8-byte message)."""

    def __init__(self, params, n_tail=2, align=16):
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

    def all_reduce(self, group=None):
        """Sum over ranks; returns the scale (1/world) that turns the sums into means."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        if world == 1 and not os.environ.get('GENESIS_FORCE_ALLREDUCE'):
            return 1.0   # (the env var keeps the collective in the step on a 1-GPU box, to exercise RCCL)
        dist.all_reduce(self.flat_g, group=group)
        if self.n64:
            dist.all_reduce(self.flat_g64, group=group)
        return 1.0 / world

    def tail(self, scale=1.0):
        return self.flat_g[self.n32:] * scale

    def grads_in_bucket(self):
        lo, hi = self.flat_g.data_ptr(), self.flat_g.data_ptr() + self.flat_g.numel() * 4
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.p32)
