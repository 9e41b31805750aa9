"""What a fork / join inside a replayed HIP graph costs: a chain of N tiny kernels on one stream against the same
kernels with a side branch of M of them, with and without a big kernel on the main branch to hide behind."""
import sys, os.path as osp, time
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch

dev = 'cuda'
a = torch.zeros(1024, device=dev)
b = torch.zeros(1024, device=dev)
big = torch.randn(4096, 4096, device=dev)


def tiny(t, n):
    for _ in range(n):
        t.add_(1.0)


def build(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def timeit(g, n=200):
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


side = torch.cuda.Stream()


def chain(n_main, n_side, big_mm=False):
    def f():
        tiny(a, 5)
        if n_side:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                tiny(b, n_side)
        if big_mm:
            torch.mm(big, big)
        tiny(a, n_main)
        if n_side:
            torch.cuda.current_stream().wait_stream(side)
        tiny(a, 5)
    return f


for big_mm in (False, True):
    base = timeit(build(chain(30, 0, big_mm)))
    print('big=%s  serial 40 tiny: %.1f us' % (big_mm, base))
    ser = timeit(build(chain(50, 0, big_mm)))
    print('big=%s  serial 60 tiny: %.1f us' % (big_mm, ser))
    fork = timeit(build(chain(30, 20, big_mm)))
    print('big=%s  40 tiny + 20 on a side branch: %.1f us  (serial 60: %.1f; fork/join overhead vs ideal %.1f)' % (big_mm, fork, ser, fork - base))
