"""Does any kernel READ memory nobody wrote?  Every tensor the package allocates with torch.empty / empty_like / _ws is filled
with NaN (GENESIS_POISON=1 semantics, patched here) and an EAGER training step (forward, backward, optimiser) is compared, bit for
bit, with the same step on ordinary allocations.  A difference -- or a NaN -- is an uninitialised read whose value reaches a
result.  usage: diag_poison.py <model> [batch] [sync_bn 0|1]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

_real_empty, _real_empty_like = torch.empty, torch.empty_like
POISON = [False]


def _fill(t):
    if POISON[0] and t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float('nan'))
        elif t.dtype == torch.uint8:
            t.fill_(0xFF)          # (a float view of it: NaN)
    return t


def _empty(*a, **k):
    return _fill(_real_empty(*a, **k))


def _empty_like(*a, **k):
    return _fill(_real_empty_like(*a, **k))


torch.empty, torch.empty_like = _empty, _empty_like
import bench  # noqa: E402
from genesis_amd.trainer import TrainStep  # noqa: E402

model_name = sys.argv[1] if len(sys.argv) > 1 else 'genesisv2'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sys.argv = [sys.argv[0], '--model', model_name, '--batch', str(batch)]
args = bench.parse()


def run(poison, steps=3):
    POISON[0] = poison
    torch.manual_seed(7)
    m = bench.build_model(args, 'cuda')
    ts = TrainStep(m, args.img, graph=False)
    g = torch.Generator().manual_seed(11)
    xs = [torch.rand(args.batch, 3, args.img, args.img, generator=g).cuda() for _ in range(2)]
    torch.manual_seed(99)
    ts.prepare(xs[0])
    outs = []
    for i in range(steps):
        out = ts.step(xs[i % 2])
        torch.cuda.synchronize()
        outs.append((ts.flat_p.clone(), ts.flat_g.clone() if hasattr(ts, 'flat_g') else None, out.clone()))
    ts.close()
    POISON[0] = False
    return outs


a, b = run(False), run(True)
ok = True
for i, ((pa, ga, oa), (pb, gb, ob)) in enumerate(zip(a, b)):
    same = torch.equal(pa, pb) and torch.equal(oa, ob)
    fin = bool(torch.isfinite(pb).all()) and bool(torch.isfinite(ob).all())
    ok = ok and same and fin
    extra = ''
    if ga is not None and gb is not None and not torch.equal(ga, gb):
        bad = (ga != gb) | ~torch.isfinite(gb)
        idx = bad.nonzero().flatten()
        extra = '; gradient bucket differs at %d of %d elements, first %d .. last %d' % (int(bad.sum()), ga.numel(), int(idx[0]), int(idx[-1]))
    print('%s B=%d step %d: parameters %s, outputs %s, finite %s%s' % (model_name, batch, i + 1, 'identical' if torch.equal(pa, pb) else
          'DIFFER (%d of %d)' % (int((pa != pb).sum()), pa.numel()), 'identical' if torch.equal(oa, ob) else 'DIFFER', fin, extra))
print('NO UNINITIALISED READ REACHES A RESULT' if ok else 'POISON REACHED A RESULT')
sys.exit(0 if ok else 1)
