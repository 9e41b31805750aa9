"""Input feeder: the device conversion against the torch ops the reference's datasets apply on the host
(ToTensor = HWC uint8 -> CHW float / 255, datasets/multid_config.py:131-135; F.interpolate(size=...) nearest)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def host_pipeline(frames, size):
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float().div(255)       # ToTensor, per batch
    if size != frames.shape[1]:
        x = F.interpolate(x, size=size)                                     # default mode: nearest
    return x


@pytest.mark.parametrize('B,Hs,C,size', [(4, 64, 3, 64), (3, 64, 3, 32), (2, 64, 3, 128), (2, 96, 3, 64), (1, 7, 1, 5),
                                         (5, 128, 3, 64)])
def test_conversion_is_bit_exact(B, Hs, C, size):
    from genesis_amd.feeder import u8hwc_to_f32chw
    rng = np.random.RandomState(B * 100 + Hs)
    frames = rng.randint(0, 256, (B, Hs, Hs, C)).astype(np.uint8)
    got = u8hwc_to_f32chw(torch.from_numpy(frames).cuda(), size).cpu()
    assert torch.equal(got, host_pipeline(frames, size))


def test_feeder_double_buffering_yields_every_batch_in_order():
    from genesis_amd.feeder import DeviceFeeder
    rng = np.random.RandomState(0)
    batches = [rng.randint(0, 256, (4, 64, 64, 3)).astype(np.uint8) for _ in range(5)]
    out = list(DeviceFeeder(batches, 64))
    assert len(out) == 5
    for x, b in zip(out, batches):
        assert x.shape == (4, 3, 64, 64) and torch.equal(x.cpu(), host_pipeline(b, 64))


def test_feeder_slot_reuse_waits_for_the_compute_stream():
    """The host runs ahead of the device in the training loop (TrainStep.step never host-syncs): the conversion of batch
    n may still be queued behind earlier work when the copy of batch n+2 wants to overwrite its uint8 source slot.
    A long spin kernel on the compute stream before every next() reproduces that; every batch must still come out intact."""
    from genesis_amd.feeder import DeviceFeeder
    rng = np.random.RandomState(1)
    batches = [rng.randint(0, 256, (8, 64, 64, 3)).astype(np.uint8) for _ in range(7)]
    feeder = DeviceFeeder(batches, 64)
    out = []
    for _ in range(len(batches)):
        torch.cuda._sleep(40_000_000)        # ~20 ms of device time queued ahead of the conversion kernel
        out.append(next(feeder))
    with pytest.raises(StopIteration):
        next(feeder)
    torch.cuda.synchronize()
    for x, b in zip(out, batches):
        assert torch.equal(x.cpu(), host_pipeline(b, 64))


def test_no_cpu_path():
    from genesis_amd.feeder import u8hwc_to_f32chw
    with pytest.raises(Exception):
        u8hwc_to_f32chw(torch.zeros(1, 4, 4, 3, dtype=torch.uint8))
