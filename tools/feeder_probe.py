"""PCIe-inclusive step rate through genesis_amd.feeder.DeviceFeeder against the resident-input rate (HIP-graph replay).
Usage: python tools/feeder_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import genesis_amd.feeder as F
from genesis_amd.trainer import TrainStep
import bench
class A: pass
a = A(); a.model = 'genesisv2'; a.K = 7; a.img = 64; a.feat_dim = 64
m = bench.build_model(a, 'cuda')
ts = TrainStep(m, 64, graph=True)
x = torch.rand(32, 3, 64, 64, device='cuda')
for _ in range(3): ts.step(x)
torch.cuda.synchronize()
n = 60
t0 = time.perf_counter()
for _ in range(n): ts.step(x)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('resident input               %.0f img/s' % (32 * n / dt), flush=True)
frames = [torch.randint(0, 256, (32, 64, 64, 3), dtype=torch.uint8) for _ in range(4)]
for depth in (2, 4, 8, 32):
    feeder = F.DeviceFeeder((frames[i % 4] for i in range(n + 5)), 64, device='cuda', depth=depth)
    for _ in range(5): ts.step(next(feeder))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tn = 0.0
    for _ in range(n):
        t1 = time.perf_counter(); xb = next(feeder); tn += time.perf_counter() - t1
        ts.step(xb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('feeder depth %d               %.0f img/s   (next(): %.2f ms avg)' % (depth, 32 * n / dt, tn / n * 1e3), flush=True)
