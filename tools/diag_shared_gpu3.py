"""A whole forward + backward of one model family repeated on fixed inputs while ANOTHER PROCESS (default; LOAD=thread: another thread
and stream of this process -- then the library's host-side state is shared too) runs GENESIS iterations on the same GPU: which
parameter gradients are not bit-reproducible under that load?  NOLOAD=1: the control.
usage: diag_shared_gpu3.py <case: v2_metric_b32 | genesis_cfg3_b32 | monet_cfg4_b32> [repetitions]"""
import os
import sys
import threading
import torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_fullbatch_gpu import Full  # noqa: E402


def load(stop, ready, own_ctx=False):
    if own_ctx:       # a second loop in the same process works in a library context of its own (include/genesis_hip.h, gx_ctx_*)
        from genesis_amd import _lib
        _lib.make_current(int(_lib.load().gx_ctx_create()))
    with torch.cuda.stream(torch.cuda.Stream()):
        gold = Full('genesis_cfg3_b32')
        x, nz = gold.x(), gold.noise()
        model = gold.build()
        ready.set()
        while not stop.is_set():
            out = gold.forward(model, x, nz)
            err, kl = gold.aggregate(out[1])
            (err + kl).backward()
            torch.cuda.current_stream().synchronize()


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'v2_metric_b32'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    mode = 'none' if os.environ.get('NOLOAD') == '1' else os.environ.get('LOAD', 'process')
    if mode == 'process':
        ctx = mp.get_context('spawn')
        stop, ready = ctx.Event(), ctx.Event()
        worker = ctx.Process(target=load, args=(stop, ready))
    else:
        stop, ready = threading.Event(), threading.Event()
        worker = threading.Thread(target=load, args=(stop, ready, os.environ.get('SHARED_CTX') != '1'))
    gold = Full(case)
    x, nz = gold.x(), gold.noise()
    body(case, reps, mode, stop, ready, worker, gold, x, nz)


def body(case, reps, mode, stop, ready, worker, gold, x, nz):


    def run():
        model = gold.build()
        out = gold.forward(model, x, nz)
        err, kl = gold.aggregate(out[1])
        (err + kl).backward()
        torch.cuda.current_stream().synchronize()
        return float(err + kl), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


    ref = run()
    if mode != 'none':
        worker.start()
        ready.wait()
    bad = {}
    nbad = 0
    for i in range(reps):
        e, g = run()
        diff = [n for n in ref[1] if not torch.equal(g[n], ref[1][n])]
        nbad += bool(diff) or e != ref[0]
        for n in diff:
            d = (g[n].double() - ref[1][n].double()).abs()
            r = bad.setdefault(n, [0, 0.0, 0])
            r[0] += 1
            r[1] = max(r[1], float(d.norm() / (ref[1][n].double().norm() + 1e-30)))
            r[2] = max(r[2], int((d > 0).sum()))
    stop.set()
    print('%s (load: %s): %d of %d repetitions differ from the first (loss or any gradient)' % (case, mode, nbad, reps), flush=True)
    for n in ref[1]:
        if n in bad:
            print('   %-52s in %3d repetitions, worst rel %.2e, up to %d of %d elements' % (n, bad[n][0], bad[n][1], bad[n][2], ref[1][n].numel()), flush=True)

    if mode != 'none':
        worker.join()


if __name__ == '__main__':
    main()
