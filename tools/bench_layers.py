"""Per-layer timing of every conv / deconv shape of the metric workload (GENESIS-V2, B=32, K=7, 64x64):
fwd, dgrad and wgrad entry points, wall time per call (HIP events, back-to-back launches) and fp32 TFLOP/s.
Usage: python tools/bench_layers.py [c3|dc|all]"""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip

DEV = 'cuda'
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


which = sys.argv[1] if len(sys.argv) > 1 else 'all'
B, K = 32, 7
tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
gf_tot = 0.0
if which in ('all', 'c3'):
    # (Cin, Cout, S): UNet down, up, seg_head, feat_head
    layers = [(3, 64, 64), (64, 64, 32), (64, 128, 16), (128, 128, 8), (128, 128, 4),
              (256, 128, 4), (256, 128, 8), (256, 64, 16), (128, 64, 32), (128, 64, 64), (64, 64, 64), (64, 64, 64)]
    print('conv3x3 (N=%d)        fwd us   TF | dgrad us   TF | wgrad us   TF' % B)
    for ci, co, s in layers:
        x = torch.randn(B, ci, s, s, device=DEV)
        w = torch.randn(co, ci, 3, 3, device=DEV) * 0.05
        dy = torch.randn(B, co, s, s, device=DEV)
        fl = 2.0 * B * ci * co * 9 * s * s
        tf = timeit(lambda: hip.conv3x3_fwd(x, w))
        td = timeit(lambda: hip.conv3x3_dgrad(dy, w)) if ci > 3 else 0.0
        tw = timeit(lambda: hip.conv3x3_wgrad(x, dy))
        tot['fwd'] += tf; tot['dgrad'] += td; tot['wgrad'] += tw; gf_tot += fl / 1e9
        print('%3d->%3d @%2d  %7.2f GF  %7.1f %5.1f | %7.1f %5.1f | %7.1f %5.1f' %
              (ci, co, s, fl / 1e9, tf, fl / tf / 1e6, td, fl / td / 1e6 if td else 0, tw, fl / tw / 1e6))
if which in ('all', 'dc'):
    N = B * K
    print('deconv5x5s2 (N=%d)    fwd us   TF | dgrad us   TF | wgrad us   TF' % N)
    for ci, co, s in [(66, 64, 4), (64, 64, 8), (64, 64, 16), (64, 64, 32)]:
        x = torch.randn(N, ci, s, s, device=DEV)
        w = torch.randn(ci, co, 5, 5, device=DEV) * 0.05
        b = torch.zeros(co, device=DEV)
        dy = torch.randn(N, co, 2 * s, 2 * s, device=DEV)
        fl = 2.0 * N * ci * co * 25 * s * s
        tf = timeit(lambda: hip.deconv5x5s2_fwd(x, w, b), 10)
        td = timeit(lambda: hip.deconv5x5s2_dgrad(dy, w), 10)
        tw = timeit(lambda: hip.deconv5x5s2_wgrad(x, dy), 10)
        tot['fwd'] += tf; tot['dgrad'] += td; tot['wgrad'] += tw; gf_tot += fl / 1e9
        print('%3d->%3d @%2d  %7.2f GF  %7.1f %5.1f | %7.1f %5.1f | %7.1f %5.1f' %
              (ci, co, s, fl / 1e9, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6))
print('total us: fwd %.0f dgrad %.0f wgrad %.0f  (%.1f GF per pass; at 100 TF/s one pass = %.0f us)' %
      (tot['fwd'], tot['dgrad'], tot['wgrad'], gf_tot, gf_tot * 10))
