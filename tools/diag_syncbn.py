"""GENESIS with cross-replica BatchNorm switched on in ONE process (world 1: the exchange is the identity): gradients against the
ordinary path, twice (determinism)."""
import os
import sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genesis_amd import sylvester  # noqa: E402
from tests.test_fullbatch_gpu import Full  # noqa: E402

dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29599', rank=0, world_size=1)
gold = Full('genesis_cfg3_b32')
x = gold.x()
nz = gold.noise()
SH = int(os.environ.get('SHARD', '0'))
if SH:
    K, B = gold.K, gold.B
    sl = slice(0, SH)
    x = x[sl]
    nz = [n[sl].contiguous() for n in nz[:K]] + [nz[K].view(K, B, -1)[:, sl].reshape(-1, nz[K].shape[-1]).contiguous()]


def run(sync):
    model = gold.build()
    sylvester.sync_bn(None, sync)
    try:
        out = gold.forward(model, x, nz)
        err, kl = gold.aggregate(out[1])
        (err + kl).backward()
    finally:
        sylvester.sync_bn(None, False)
    return float(err + kl), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


e0, g0 = run(False)
e1, g1 = run(True)
e2, g2 = run(True)
e3, g3 = run(False)
print('elbo plain %.6f sync %.6f sync again %.6f plain again %.6f' % (e0, e1, e2, e3))
big = max(float(v.double().norm()) for v in g0.values())
for tag, a, b in (('plain vs plain', g0, g3), ('sync vs sync', g1, g2), ('sync vs plain', g1, g0)):
    worst = sorted(((float((a[n].double() - b[n].double()).norm()) / (float(b[n].double().norm()) + 1e-6 * big), n) for n in b), reverse=True)[:4]
    print(tag, ' '.join('%s %.2e' % (n, e) for e, n in worst))
