"""Per-parameter gradient error of the HIP path and of the CPU fp32 oracle against the fp64 oracle on a full-batch
fixture's weights / inputs / noise (tests/golden/full_*.npz): python tools/diag_fullbatch.py v2_cfg2_b64 [top]"""
import sys
import os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from tests.test_fullbatch_gpu import Full
from oracle import v2_oracle as VO, genesis_oracle as GO, monet_oracle as MO

name = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
gold = Full(name)
model = gold.build()
x, nz = gold.x(), gold.noise()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
isp = lambda k: not (k == 'std' or k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))
seeds = list(torch.from_numpy(gold.g['seed_idx']).unbind(0)) if gold.fam == 'v2' else None
gr = {}
for dt in (torch.float64, torch.float32):
    p = {k: (v.clone().to(dt if v.dtype == torch.float32 else v.dtype).requires_grad_(True) if isp(k) and v.is_floating_point()
             else v.clone().to(dt if v.dtype == torch.float32 else v.dtype)) for k, v in sd.items()}
    n = [t.to(dt) for t in nz]
    if gold.fam == 'v2':
        o = VO.v2_forward(p, x.to(dt), gold.cfg, n[0], n[1:], seed_idx=seeds, reference_form=False)
    elif gold.fam == 'genesis':
        o = GO.genesis_forward(p, x.to(dt), gold.cfg, n[:gold.K], n[gold.K])
    else:
        o = MO.monet_forward(p, x.to(dt), gold.cfg, n[0])
    e, kl = gold.aggregate(o[1])
    (e + kl).backward()
    gr[dt] = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in p.items() if v.requires_grad}
out = gold.forward(model, x, nz, torch.from_numpy(gold.g['seed_idx']).cuda() if gold.fam == 'v2' else None)
e, kl = gold.aggregate(out[1])
(e + kl).backward()
rows = []
gmax = max(float(v.norm()) for v in gr[torch.float64].values())
for k, p in model.named_parameters():
    g64 = gr[torch.float64][k]
    den = max(float(g64.norm()), 1e-6 * gmax)
    gh = (p.grad if p.grad is not None else torch.zeros_like(p)).double().cpu()
    rows.append((float((gh - g64).norm()) / den, float((gr[torch.float32][k] - g64).norm()) / den, k))
rows.sort(reverse=True)
print('%-52s %12s %12s' % ('parameter', 'hip-vs-f64', 'cpu32-vs-f64'))
for r in rows[:top]:
    print('%-52s %12.3e %12.3e' % (r[2], r[0], r[1]))
