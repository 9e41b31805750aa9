"""Build-container script: how fast is the oracle (reference-equivalent form) against the REAL reference on the same
host?  bench.py's cpu_baseline leg times the oracle on the GPU box (the reference cannot travel); this ratio says what
that number means in terms of the reference itself.  Same weights (the reference's own init under torch.manual_seed(0)),
same batch, same thread count, full training step (forward, loss aggregation, GECO, backward, Adam), interleaved
repetitions; writes profiles/cpu_port_vs_reference.json.

    python tools/cpu_ratio.py [--batch 32] [--reps 4] [--threads 8]
"""
import argparse
import json
import os
import os.path as osp
import sys
import time

REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from oracle import ref_import as R  # noqa: E402
from oracle import v2_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--threads', type=int, default=min(8, os.cpu_count() or 8))
    ap.add_argument('--K', type=int, default=7)
    ap.add_argument('--img', type=int, default=64)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    mods = R.import_reference()
    cfg = R.reference_cfg(K_steps=args.K, img_size=args.img)
    torch.manual_seed(0)
    model = mods['genesisv2_config'].load(cfg)
    model.train()
    S = args.img
    geco_r = mods['geco'].GECO(0.5655 * 3 * S * S, 1e-5 * (64 ** 2 / S ** 2), 0.99, 1.0, 1e-10, 10)
    opt_r = torch.optim.Adam(model.parameters(), 1e-4)
    ocfg = O.make_cfg(K_steps=args.K, img_size=S, feat_dim=64)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt_o = torch.optim.Adam(list(p.values()), 1e-4)
    geco_o = O.make_geco(S)
    x = torch.rand(args.batch, 3, S, S, generator=torch.Generator().manual_seed(1234))

    def ref_step():
        opt_r.zero_grad()
        _, losses, _, _, _ = model(x)
        err = losses.err.mean(0)
        kl = torch.stack(list(losses.kl_l_k), dim=1).mean(dim=0).sum()      # train.py:226-242
        geco_r.loss(err, kl).backward()
        opt_r.step()

    def ora_step():
        O.train_step(p, opt_o, geco_o, x, ocfg)

    ref_step(); ora_step()                                                    # warm-up
    tr, to = [], []
    for _ in range(args.reps):
        t0 = time.time(); ref_step(); tr.append(time.time() - t0)
        t0 = time.time(); ora_step(); to.append(time.time() - t0)
    best_r, best_o = min(tr), min(to)
    out = {'reference_s_per_step': tr, 'oracle_s_per_step': to, 'batch': args.batch, 'K': args.K, 'img': S,
           'threads': args.threads, 'host_cpus': os.cpu_count(),
           'reference_images_per_sec': args.batch / best_r, 'oracle_images_per_sec': args.batch / best_o,
           'port_vs_reference_speed_ratio': best_r / best_o,
           'note': 'ratio = reference step time / oracle step time (best of the interleaved repetitions); > 1: the oracle '
                   'is faster than the reference on this host',
           'torch': torch.__version__}
    path = osp.join(REPO, 'profiles', 'cpu_port_vs_reference.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
