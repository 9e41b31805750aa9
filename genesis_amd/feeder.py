"""On-device batch feeder: the reference converts every sample to fp32 on the host
(datasets/multid_config.py:131-135: ToTensor + F.interpolate; multi_object_config.py:176-186) and copies the fp32 batch
to the GPU inside the training loop (train.py:218-220).  Here the uint8 HWC frames are staged in pinned memory, copied
on a side stream one batch ahead (a ring of slots) and converted to the fp32 NCHW batch in [0,1] by one HIP launch, so
the step after compute does not wait on the host."""
import ctypes
import time

import torch

from . import _lib
from ._lib import GenesisHipError


def u8hwc_to_f32chw(frames_u8, img_size=None, out=None):
    """frames_u8: uint8 device tensor [B, Hs, Ws, C] -> float32 [B, C, S, S] (S = img_size or Hs), values / 255,
    nearest-neighbour resampled like F.interpolate(size=S)."""
    if not frames_u8.is_cuda:
        raise GenesisHipError('feeder: frames must be on the HIP device; there is no CPU path')
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or not frames_u8.is_contiguous():
        raise GenesisHipError('feeder: expected a contiguous uint8 [B,H,W,C] tensor')
    B, Hs, Ws, C = frames_u8.shape
    H = W = int(img_size) if img_size else Hs
    if img_size is None:
        W = Ws
    if out is None:
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=frames_u8.device)
    _lib.call('gx_u8hwc_to_f32chw', ctypes.c_void_p(frames_u8.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, Hs, Ws,
              C, H, W, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out


class DeviceFeeder(object):
    """Iterates fp32 device batches from an iterable of uint8 HWC host batches (numpy arrays or CPU tensors
    [B, H, W, C]).  A ring of `depth` slots (pinned staging buffer + uint8 device buffer); the host->device copy of
    batch i+1 runs on a side stream while batch i is being consumed, and the consumer's stream waits for it with one
    event (copy -> compute).

    Slot reuse.  TrainStep.step never host-syncs, so the host runs far ahead of the device: the conversion kernel that
    reads a uint8 slot may still be queued behind many training steps when the ring comes round to that slot again.
    A slot is therefore refilled only after the HOST has seen both of its events complete -- the copy out of its pinned
    buffer and the conversion kernel that read its device buffer (`consumed`, recorded on the consumer's stream) -- by
    polling; nothing on the device waits compute -> copy.  Measured on MI355X / ROCm 7 under HIP-graph replay
    (tools/feeder_probe.py): a device-side compute -> copy event wait per batch costs 22 % img/s, and a host that blocks
    on an event fewer than ~30 batches old starves the graph-launch queue (depth 2: 1450 img/s, depth 8: 3700, against
    6535 resident) -- hence the deep ring: at depth 32 the poll passes immediately in steady state (64x64: 26 MB)."""

    def __init__(self, host_batches, img_size, device='cuda', depth=32):
        self.it = iter(host_batches)
        self.img_size = img_size
        self.device = torch.device(device)
        self.depth = max(2, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.pinned = [None] * self.depth
        self.dev_u8 = [None] * self.depth
        self.ready = [None] * self.depth     # copy-stream event: the H2D copy into dev_u8[s] has executed
        self.consumed = [None] * self.depth  # consumer-stream event: the conversion kernel has read dev_u8[s]
        self.filled = [False] * self.depth
        self.head = 0                        # slot the next __next__ consumes
        self.tail = 0                        # slot the next prefetch fills
        self._prefetch()

    @staticmethod
    def _host_wait(ev):
        if ev is not None:
            while not ev.query():
                time.sleep(2e-4)

    def _prefetch(self):
        try:
            nxt = next(self.it)
        except StopIteration:
            return
        t = torch.as_tensor(nxt)
        if t.dtype != torch.uint8 or t.dim() != 4:
            raise GenesisHipError('feeder: host batches must be uint8 [B,H,W,C]')
        s = self.tail
        self._host_wait(self.ready[s])       # pinned[s] is free: its previous copy has executed
        self._host_wait(self.consumed[s])    # dev_u8[s] is free: the conversion kernel that read it has run
        if self.pinned[s] is None or self.pinned[s].shape != t.shape:
            # the whole ring at once, the first time a batch shape is seen: a pinned allocation costs ~1 ms of host time,
            # and 32 of them spread over the first 32 steps let the device queue run dry (the host needs the whole next
            # segment to get ahead again: 1.5 k -> 2.8 k -> 6.1 k img/s over the first 300 steps, measured)
            for q in range(self.depth):
                if self.pinned[q] is None or self.pinned[q].shape != t.shape:
                    self._host_wait(self.ready[q]); self._host_wait(self.consumed[q])
                    self.pinned[q] = torch.empty(t.shape, dtype=torch.uint8, pin_memory=True)
                    self.dev_u8[q] = torch.empty(t.shape, dtype=torch.uint8, device=self.device)
        self.pinned[s].copy_(t)
        with torch.cuda.stream(self.copy_stream):
            self.dev_u8[s].copy_(self.pinned[s], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.ready[s] = ev
        self.filled[s] = True
        self.tail = (s + 1) % self.depth

    def __iter__(self):
        return self

    def __next__(self):
        s = self.head
        if not self.filled[s]:
            raise StopIteration
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ready[s])
        x = u8hwc_to_f32chw(self.dev_u8[s], self.img_size)
        done = torch.cuda.Event()
        done.record(cur)
        self.consumed[s] = done
        self.filled[s] = False
        self.head = (s + 1) % self.depth
        self._prefetch()          # refills the next free slot while the caller trains on x
        return x
