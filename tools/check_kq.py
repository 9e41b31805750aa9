"""k-quad tap-conv kernels (gx_kq.hip) against the round-1 tap-conv kernels: same inputs, max relative difference and
wall time per call (HIP events).  Usage: python tools/check_kq.py"""
import os, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
os.environ['GENESIS_KQ'] = '2'
import torch
from genesis_amd import hip_ops as hip, _lib

DEV = 'cuda'
torch.manual_seed(0)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def both(fn):
    _lib.call('gx_kq_policy', 0)
    ref = fn(); t0 = timeit(fn)
    _lib.call('gx_kq_policy', 2)
    got = fn(); t1 = timeit(fn)
    if isinstance(ref, tuple):
        err = max(float((g - r).abs().max() / r.abs().max()) for g, r in zip(got, ref))
    else:
        err = float((got - ref).abs().max() / ref.abs().max())
    return err, t0, t1


def report(name, fl, err, t0, t1):
    print('%-34s rel.err %.2e   old %7.1f us %5.1f TF | kq %7.1f us %5.1f TF' % (name, err, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6), flush=True)
    assert err < 2e-5, name


for N, ci, co, s in [(3, 64, 64, 16), (5, 16, 8, 8), (2, 72, 64, 32), (224, 64, 64, 16), (224, 64, 64, 32), (32, 64, 64, 64)]:
    x = torch.randn(N, ci, s, s, device=DEV)
    w = torch.randn(ci, co, 5, 5, device=DEV) * 0.05
    b = torch.randn(co, device=DEV)
    dy = torch.randn(N, co, 2 * s, 2 * s, device=DEV)
    fl = 2.0 * N * ci * co * 25 * s * s
    report('deconv fwd   N%d %d->%d @%d' % (N, ci, co, s), fl, *both(lambda: hip.deconv5x5s2_fwd(x, w, b)))
    report('deconv dgrad N%d %d->%d @%d' % (N, ci, co, s), fl, *both(lambda: hip.deconv5x5s2_dgrad(dy, w)))
    if co % 8 == 0:
        g, be = torch.rand(co, device=DEV) + 0.5, torch.randn(co, device=DEV)
        def st():
            r = hip.deconv5x5s2_gn_stats_fwd(x, w, b, g, be, 8 if co % 64 == 0 else 1, 1e-5)
            return tuple(t for t in r if torch.is_tensor(t) and t.is_floating_point())
        report('deconv fwd+gn stats', fl, *both(st))
for N, ci, co, s in [(2, 16, 8, 16), (3, 64, 128, 32), (32, 64, 64, 64), (32, 128, 64, 64), (32, 128, 64, 32)]:
    x = torch.randn(N, ci, s, s, device=DEV)
    w = torch.randn(co, ci, 3, 3, device=DEV) * 0.05
    dy = torch.randn(N, co, s, s, device=DEV)
    fl = 2.0 * N * ci * co * 9 * s * s
    report('conv3x3 fwd   N%d %d->%d @%d' % (N, ci, co, s), fl, *both(lambda: hip.conv3x3_fwd(x, w)))
    report('conv3x3 dgrad N%d %d->%d @%d' % (N, ci, co, s), fl, *both(lambda: hip.conv3x3_dgrad(dy, w)))
print('OK')
