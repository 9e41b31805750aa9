"""fp64 error budget for MONet cfg4-shaped gradients: HIP fp32 vs CPU fp32, both against the fp64 oracle."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np, torch
from tests.test_monet_oracle import MonetGolden
from tests.test_monet_gpu import build
from oracle import monet_oracle as M
DEV='cuda'
gold = MonetGolden(sys.argv[1] if len(sys.argv) > 1 else 'cfg4')
model = build(gold)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
x, eps = gold.inputs()
def oracle(dt):
    p = {k: v.clone().to(dt).requires_grad_(k != 'std') for k, v in sd.items()}
    out = M.monet_forward(p, x.to(dt), gold.cfg, eps.to(dt))
    e, kl, km = M.aggregate_losses(out[1])
    (e + kl + km).backward()
    return out, {k: v.grad.double() for k, v in p.items() if k != 'std'}
o64, g64 = oracle(torch.float64)
o32, g32 = oracle(torch.float32)
recon, losses, stats, _, comp = model(x.to(DEV), eps.to(DEV))
(losses.err.mean(0) + torch.stack(losses.kl_l_k, 1).mean(0).sum() + losses.kl_m.mean(0)).backward()
rel = lambda a, r: float((a.detach().cpu().double() - r).norm() / (r.norm() + 1e-30))
print('recon hip %.2e cpu32 %.2e' % (rel(recon, o64[0].double()), rel(o32[0], o64[0].double())))
print('log_m hip %.2e cpu32 %.2e' % (rel(torch.stack(list(stats.log_m_k)), torch.stack(o64[2]['log_m_k']).double()), rel(torch.stack(o32[2]['log_m_k']), torch.stack(o64[2]['log_m_k']).double())))
for n, prm in model.named_parameters():
    print('%-50s hip %.2e  cpu32 %.2e' % (n, rel(prm.grad, g64[n]), rel(g32[n], g64[n])))
