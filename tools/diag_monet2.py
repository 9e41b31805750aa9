import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np, torch
from tests.test_monet_oracle import MonetGolden
from tests.test_monet_gpu import build
from genesis_amd.trainer import TrainStep
from oracle import monet_oracle as M, v2_oracle as O
DEV='cuda'
gold = MonetGolden('tiny')
model = build(gold)
sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
x, _ = gold.inputs()
_, eps = gold.inputs(1)
# oracle step
p = {k: v.clone().requires_grad_(k != 'std') for k, v in sd0.items()}
opt = torch.optim.Adam([v for k, v in p.items() if k != 'std'], 1e-4)
geco = O.make_geco(gold.S)
_, losses, _, _, _ = M.monet_forward(p, x, gold.cfg, eps)
err, kl_l, kl_m = M.aggregate_losses(losses)
geco.loss(err, kl_l + kl_m).backward()
gref = {k: v.grad.clone() for k, v in p.items() if k != 'std'}
opt.step()
# hip step
ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
ts.step(x.to(DEV), eps=eps.to(DEV))
torch.cuda.synchronize()
print('%-52s %10s %10s %12s %12s' % ('param', 'max|dp|/lr', 'frac>0.5lr', 'ref|g|med', 'hip-ref g rel'))
named = dict(model.named_parameters())
for k in gref:
    dp = (named[k].detach().cpu() - p[k].detach()).abs() / 1e-4
    gh = named[k].grad.detach().cpu()
    rel = float((gh - gref[k]).norm() / (gref[k].norm() + 1e-30))
    print('%-52s %10.3f %10.4f %12.3e %12.3e' % (k, float(dp.max()), float((dp > 0.5).float().mean()), float(gref[k].abs().median()), rel))
