"""Per-parameter fp32 error budget of the closed-form golden cases: how far the fp32 gradient of the ORACLE sits from its fp64
gradient on the fixture's own weights, inputs and noise -- |g32 - g64| / max(|g64|, 1e-6 max |g64|) per parameter.  Written to
tests/golden/v2_<case>_budget.npz; tests/test_model_gpu.py::test_golden_parity_with_every_conv3x3_on_the_winograd_kernel derives its
per-parameter gradient bar from it (review, round 5: "the cfg5 bar derived from a budget instead of the 3e-2").  No reference needed:
the oracle (oracle/v2_oracle.py) is pinned to the reference by tests/test_oracle_vs_golden.py.

    python tests/golden/make_golden_budget.py cfg5 metric"""
import json
import os.path as osp
import sys

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, osp.dirname(osp.dirname(HERE)))
from genesis_amd import testing as T  # noqa: E402
from oracle import v2_oracle as O  # noqa: E402


def run(case):
    g = np.load(osp.join(HERE, 'v2_%s.npz' % case), allow_pickle=False)
    cfg = json.loads(str(g['cfg_json']))
    cfg['pixel_std2'] = cfg['pixel_std1']
    B, K, S, D = int(g['B']), cfg['K_steps'], cfg['img_size'], cfg['feat_dim']
    keys = [str(k) for k in g['sd_keys']]
    numel = [int(n) for n in g['sd_numel']]
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    torch.manual_seed(0)
    model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False)))
    sd = T.formula_state_dict(model.state_dict())
    assert list(sd.keys()) == keys and [v.numel() for v in sd.values()] == numel
    x = T.make_input(int(g['x_seed']), B, S)
    rp, eps = T.draw_noise(int(g['noise_seed']), B, S, D, K)
    seeds = list(torch.from_numpy(g['seed_idx']).unbind(0)) if 'seed_idx' in g.files else None
    grads = {}
    for dt in (torch.float32, torch.float64):
        p = {k: (v.clone().to(dt if v.dtype == torch.float32 else v.dtype).requires_grad_(True)) for k, v in sd.items()}
        out = O.v2_forward(p, x.to(dt), cfg, rp.to(dt), [e.to(dt) for e in eps], seed_idx=seeds, reference_form=False)
        err, kl_l, kl_m = O.aggregate_losses(out[1])
        (err + kl_l + kl_m).backward()
        grads[dt] = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in p.items()}
    gmax = max(float(v.norm()) for v in grads[torch.float64].values())
    budget = [float((grads[torch.float32][k] - grads[torch.float64][k]).norm()) / max(float(grads[torch.float64][k].norm()), 1e-6 * gmax)
              for k in keys]
    np.savez_compressed(osp.join(HERE, 'v2_%s_budget.npz' % case), param_names=np.array(keys), budget=np.array(budget),
                        grad_norms_f64=np.array([float(grads[torch.float64][k].norm()) for k in keys]), grad_max_f64=np.float64(gmax))
    print(case, 'budget max %.2e median %.2e' % (max(budget), sorted(budget)[len(budget) // 2]))
    for i in np.argsort(-np.array(budget))[:5]:
        print('   %-44s %.2e' % (keys[i], budget[i]))


if __name__ == '__main__':
    for c in sys.argv[1:] or ['cfg5', 'metric']:
        run(c)
