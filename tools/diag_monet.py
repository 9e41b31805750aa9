import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import numpy as np, torch
from tests.test_monet_oracle import MonetGolden
from tests.test_monet_gpu import build
from genesis_amd.trainer import TrainStep
from genesis_amd import testing as T
DEV='cuda'
for case in ('tiny','cfg4'):
    gold = MonetGolden(case)
    model = build(gold)
    x, eps = gold.inputs()
    recon, losses, stats, _, comp = model(x.to(DEV), eps.to(DEV))
    err = losses.err.mean(0); kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum() + losses.kl_m.mean(0)
    (err+kl).backward()
    names=[str(n) for n in gold.g['param_names']]; norms=gold.g['grad_norms']
    named=dict(model.named_parameters())
    print(case, 'grad norm rel diffs > 1e-3:')
    for i,n in enumerate(names):
        g=named[n].grad; got=float(g.double().norm())
        s=T.summarize(g); ref=gold.g['grad/%s/samples'%n].astype(np.float64)
        l2=np.linalg.norm(s['samples']-ref)/(np.linalg.norm(ref)+1e-30)
        if abs(got-norms[i])>1e-3*norms[i] or l2>2e-3: print('   %-50s norm %.5g vs %.5g  sampleL2 %.2e'%(n,got,norms[i],l2))
    model = build(gold)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    for it in range(3):
        _, e = gold.inputs(1+it)
        out = ts.step(x.to(DEV), eps=e.to(DEV)).cpu().numpy()
        print('  step', it, 'hip elbo/err/kl/beta', out, ' ref', gold.g['train_hist'][it,:4])
