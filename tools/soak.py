"""Soak: 3000 replayed training steps at the metric configuration -- throughput, allocator growth, finiteness."""
import sys, time; sys.path.insert(0, '.')
import torch, bench
from genesis_amd.trainer import TrainStep
sys.argv=[sys.argv[0]]
args=bench.parse()
m=bench.build_model(args,'cuda')
ts=TrainStep(m,64,graph=True)
xs=[torch.rand(32,3,64,64,device='cuda') for _ in range(4)]
ts.prepare(xs[0])
torch.cuda.synchronize(); m0=torch.cuda.memory_allocated(); r0=torch.cuda.memory_reserved()
t0=time.perf_counter()
for i in range(3000):
    out=ts.step(xs[i%4])
    if i%500==0:
        torch.cuda.synchronize(); print(i, [round(float(v),3) for v in out], torch.cuda.memory_allocated()-m0, torch.cuda.memory_reserved()-r0, flush=True)
torch.cuda.synchronize(); dt=time.perf_counter()-t0
print('3000 steps', 32*3000/dt, 'img/s', 'finite', bool(torch.isfinite(ts.flat_p).all()), 'mem delta', torch.cuda.memory_allocated()-m0)
