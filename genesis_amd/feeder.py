"""On-device batch feeder: the reference converts every sample to fp32 on the host
(datasets/multid_config.py:131-135: ToTensor + F.interpolate; multi_object_config.py:176-186) and copies the fp32 batch
to the GPU inside the training loop (train.py:218-220).  Here the uint8 HWC frames are staged in pinned memory, copied
on a side stream one batch ahead (double buffering) and converted to the fp32 NCHW batch in [0,1] by one HIP launch,
so the step after compute does not wait on the host."""
import ctypes

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
    [B, H, W, C]).  Two pinned staging buffers and two device buffers; the host->device copy of batch i+1 runs on a
    side stream while batch i is being consumed."""

    def __init__(self, host_batches, img_size, device='cuda'):
        self.it = iter(host_batches)
        self.img_size = img_size
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.pinned = [None, None]
        self.dev_u8 = [None, None]
        self.ready = [None, None]
        self.slot = 0
        self._prefetch()

    def _prefetch(self):
        try:
            nxt = next(self.it)
        except StopIteration:
            self.ready[self.slot] = None
            return
        t = torch.as_tensor(nxt)
        if t.dtype != torch.uint8 or t.dim() != 4:
            raise GenesisHipError('feeder: host batches must be uint8 [B,H,W,C]')
        s = self.slot
        if self.pinned[s] is None or self.pinned[s].shape != t.shape:
            self.pinned[s] = torch.empty(t.shape, dtype=torch.uint8).pin_memory()
            self.dev_u8[s] = torch.empty(t.shape, dtype=torch.uint8, device=self.device)
        self.pinned[s].copy_(t)
        with torch.cuda.stream(self.copy_stream):
            self.dev_u8[s].copy_(self.pinned[s], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.ready[s] = ev

    def __iter__(self):
        return self

    def __next__(self):
        s = self.slot
        ev = self.ready[s]
        if ev is None:
            raise StopIteration
        torch.cuda.current_stream().wait_event(ev)
        x = u8hwc_to_f32chw(self.dev_u8[s], self.img_size)
        self.slot = 1 - s
        self._prefetch()          # refills the OTHER slot while the caller trains on x
        return x
