import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tests.common import Golden
from genesis_amd import testing as T
import tests.test_model_gpu as M
for case in M.DEFAULT_CASES:
    gold = Golden(case); model = M.build(gold)
    x, rp, eps = gold.inputs()
    recon, losses, stats, att, comp = M.run(model, gold, x, rp, eps)
    err = losses.err.mean(0); kl = torch.stack(losses.kl_l_k, 1).mean(0).sum()
    if 'kl_m' in losses: kl = kl + losses.kl_m.mean(0)
    (err + kl).backward()
    norms = gold.g['grad_norms']; big = float(np.max(norms)); wn = 0; wl = 0
    for i, (n, p) in enumerate(model.named_parameters()):
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        got = float(g.double().norm()); ref = float(norms[i])
        if ref > 1e-4 * big: wn = max(wn, abs(got - ref) / ref)
        s = T.summarize(g); r = gold.g['grad/%s/samples' % n].astype(np.float64)
        if np.linalg.norm(r) > 1e-4 * big * np.sqrt(len(r) / max(1, int(s['n']))):
            wl = max(wl, np.linalg.norm(s['samples'].astype(np.float64) - r) / np.linalg.norm(r))
    print('%-20s worst norm rel %.2e   worst sample rel-L2 %.2e' % (case, wn, wl), flush=True)
