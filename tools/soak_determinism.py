"""Two identically seeded training runs of N replayed steps must end in bit-identical parameters (a race inside any kernel --
LDS fills against reads, slabs against their reduce, deferred records sharing a destination -- shows up as differing bits; the
first version of the two-rows-per-tile weight gradients was caught this way).  usage: soak_determinism.py <model> [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from genesis_amd.trainer import TrainStep  # noqa: E402

model_name = sys.argv[1] if len(sys.argv) > 1 else 'genesisv2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sys.argv = [sys.argv[0], '--model', model_name]
args = bench.parse()


def run():
    torch.manual_seed(7)
    m = bench.build_model(args, 'cuda')
    ts = TrainStep(m, args.img, graph=True)
    g = torch.Generator().manual_seed(11)
    xs = [torch.rand(args.batch, 3, args.img, args.img, generator=g).cuda() for _ in range(4)]
    torch.manual_seed(99)
    ts.prepare(xs[0])
    snaps = []
    for i in range(steps):
        out = ts.step(xs[i % 4])
        if (i + 1) % 100 == 0 or i + 1 == steps:
            torch.cuda.synchronize()
            snaps.append((i + 1, ts.flat_p.clone(), out.clone()))
    ts.close()
    return snaps


a, b = run(), run()
ok = True
for (i, pa, oa), (_, pb, ob) in zip(a, b):
    same = torch.equal(pa, pb) and torch.equal(oa, ob)
    ok = ok and same and bool(torch.isfinite(pa).all())
    print('%s step %4d: parameters %s, outputs %s' % (model_name, i, 'identical' if torch.equal(pa, pb) else 'DIFFER (%d of %d)'
          % (int((pa != pb).sum()), pa.numel()), [round(float(v), 4) for v in oa]))
print('DETERMINISTIC' if ok else 'NOT DETERMINISTIC')
sys.exit(0 if ok else 1)
