"""Times the big transposed-conv / conv3x3 launches (HIP events).  argv: list of which:N:S triples"""
import os, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip, _lib
pol = int(os.environ.get('KQ_POLICY', '1'))
_lib.call('gx_kq_policy', pol)
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for spec in sys.argv[1:]:
    which, N, s = spec.split(':'); N = int(N); s = int(s)
    data = os.environ.get('KQ_DATA', 'randn')
    gen = {'randn': torch.randn, 'zeros': torch.zeros, 'uniform': lambda *a, **k: torch.rand(*a, **k) - 0.5,
           'ones': torch.ones}[data]
    x = gen(N, 64, s, s, device='cuda')
    w = gen(64, 64, 5, 5, device='cuda') * 0.05
    b = gen(64, device='cuda')
    dy = gen(N, 64, 2 * s, 2 * s, device='cuda')
    w3 = gen(64, 64, 3, 3, device='cuda') * 0.05
    taps = 9 if which in ('c3', 'wgrad3') else 25
    fn = {'fwd': lambda: hip.deconv5x5s2_fwd(x, w, b), 'dgrad': lambda: hip.deconv5x5s2_dgrad(dy, w),
          'c3': lambda: hip.conv3x3_fwd(x, w3), 'wgrad': lambda: hip.deconv5x5s2_wgrad(x, dy),
          'wgrad3': lambda: hip.conv3x3_wgrad(x, x)}[which]
    fl = 2.0 * N * 64 * 64 * taps * s * s
    t = timeit(fn)
    extra = ''
    if os.environ.get('KQ_ABL'):
        o = fn().flatten()[:2].tolist()
        extra = '   [workgroup 0: main loop %.1f us at %.2f GHz]' % (o[1], o[0] * 0.1)
    print('%-8s N%-4d @%-3d %8.1f us  %6.1f TF%s' % (which, N, s, t, fl / t / 1e6, extra), flush=True)
