"""One process repeats single ops on fixed inputs while ANOTHER process runs GENESIS training iterations on the same GPU: is an op's
result bit-reproducible under that load?  (diag_shared_gpu.py: both processes run the small op -- no failures.)
usage: diag_shared_gpu2.py [seconds]"""
import os
import sys
import time
import torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(stop, ready):
    from tests.test_fullbatch_gpu import Full
    gold = Full('genesis_cfg3_b32')
    x, nz = gold.x(), gold.noise()
    model = gold.build()
    ready.set()
    n = 0
    while not stop.is_set():
        out = gold.forward(model, x, nz)
        err, kl = gold.aggregate(out[1])
        (err + kl).backward()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print('load process: %d iterations' % n, flush=True)


def tester(stop, ready, seconds):
    from genesis_amd import hip_ops as hip
    dev = 'cuda'
    g = torch.Generator().manual_seed(3)
    cases = {}
    for name, (N, Ci, Co, S) in {'wgrad_small 32->32 @32': (112, 32, 32, 32), 'wgrad_small 64->64 @8': (112, 64, 64, 8),
                                  'wgrad_small 4->32 @64': (112, 4, 32, 64), 'wgrad_small 32->64 @16': (112, 32, 64, 16)}.items():
        x = torch.randn(N, Ci, S, S, generator=g).to(dev)
        dy = torch.randn(N, Co, S // 2, S // 2, generator=g).to(dev)
        cases[name] = (lambda x=x, dy=dy: hip.conv3x3s2_wgrad_small(x, dy))
    x1 = torch.ones(112, 32, 32, 32, device=dev)
    dy1 = torch.ones(112, 32, 16, 16, device=dev)
    cases['wgrad_small ones (32->32 @32)'] = lambda: hip.conv3x3s2_wgrad_small(x1, dy1)
    a = torch.randn(1024, 1024, generator=g).to(dev)
    cases['torch matmul (control)'] = lambda: a @ a
    xs = torch.randn(112, 32, 32, 32, generator=g).to(dev)
    cases['torch sum(dim) (control)'] = lambda: xs.sum((0, 2, 3))
    refs = {k: f() for k, f in cases.items()}
    torch.cuda.synchronize()
    ready.wait()
    bad = {k: [0, 0, 0.0, None] for k in cases}
    seen_vals = set()
    by_c = [0] * 8
    t0 = time.time()
    while time.time() - t0 < seconds:
        for k, f in cases.items():
            out = f()
            bad[k][1] += 1
            if not torch.equal(out, refs[k]):
                bad[k][0] += 1
                d = (out.double() - refs[k].double()).abs()
                bad[k][2] = max(bad[k][2], float(d.norm() / refs[k].double().norm()))
                if 'ones' in k and d.dim() == 4:
                    vals = sorted(set((out.double() - refs[k].double())[d > 0].tolist()))
                    seen_vals.update(vals)
                    for co in (d > 0).nonzero()[:, 0].tolist():
                        by_c[co % 8] += 1
                if bad[k][3] is None and d.dim() == 4:
                    idx = (d > 0).nonzero()
                    bad[k][3] = 'differing %d: co %s ci %s taps %s' % (idx.shape[0], sorted(set(idx[:, 0].tolist()))[:12], sorted(set(idx[:, 1].tolist()))[:12],
                                                                       sorted(set((int(u), int(v)) for u, v in idx[:, 2:].tolist())))
    stop.set()
    print('all-ones operands: wrong elements by output channel modulo 8 (c = 0 .. 7): %s' % by_c, flush=True)
    print('all-ones operands: distinct (result - reference) values: %s' % sorted(seen_vals)[:40], flush=True)
    for k, (b, n, w, pat) in bad.items():
        print('%-28s %d of %d repetitions differ from the first%s' % (k, b, n, ' (worst rel %.2e; first: %s)' % (w, pat) if b else ''), flush=True)


if __name__ == '__main__':
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    ctx = mp.get_context('spawn')
    stop, ready = ctx.Event(), ctx.Event()
    if os.environ.get('NOLOAD') == '1':
        ready.set()
        tester(stop, ready, seconds)
    elif os.environ.get('LOAD') == 'thread':       # the load in THIS process: another thread, another HIP stream
        import threading
        ts, tr = threading.Event(), threading.Event()

        def load_thread():
            with torch.cuda.stream(torch.cuda.Stream()):
                load(ts, tr)
        th = threading.Thread(target=load_thread)
        th.start()
        tester(ts, tr, seconds)
        th.join()
    else:
        p = ctx.Process(target=load, args=(stop, ready))
        p.start()
        tester(stop, ready, seconds)
        p.join()
