"""Keeps the GPU busy with GENESIS training iterations for N seconds (a co-tenant for the tests of finding 48):
python tools/load_gpu.py [seconds]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_fullbatch_gpu import Full  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
gold = Full('genesis_cfg3_b32')
x, nz = gold.x(), gold.noise()
model = gold.build()
t0 = time.time()
n = 0
while time.time() - t0 < seconds:
    out = gold.forward(model, x, nz)
    err, kl = gold.aggregate(out[1])
    (err + kl).backward()
    torch.cuda.synchronize()
    n += 1
print('load: %d iterations' % n)
