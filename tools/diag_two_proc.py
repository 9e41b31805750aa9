"""Two processes on ONE GPU, each running GENESIS forward + backward on its shard several times (no exchange at all unless SYNC=1):
are a process's gradients bit-reproducible while another process shares the GPU?"""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from genesis_amd import sylvester
    from tests.test_fullbatch_gpu import Full
    gold = Full(os.environ.get('CASE', 'genesis_cfg3_b32'))
    K, B = gold.K, gold.B
    sl = slice(rank * B // world, (rank + 1) * B // world)
    x = gold.x()[sl]
    nz = gold.noise()
    if gold.fam == 'genesis':
        nz = [n[sl].contiguous() for n in nz[:K]] + [nz[K].view(K, B, -1)[:, sl].reshape(-1, nz[K].shape[-1]).contiguous()]
    else:
        nz = [nz[0][sl].contiguous()] + [n[sl].contiguous() for n in nz[1:]]
    sync = os.environ.get('SYNC') == '1'
    runs = []
    logs = []
    # TRACE=1: checksums of the operands and the result of every small stride-2 weight-gradient call (and of the kernels around it)
    trace = []
    if os.environ.get('TRACE') == '1':
        from genesis_amd import hip_ops as hip

        def cks(t):
            return None if t is None else (float(t.double().sum()), float(t.double().abs().sum()))

        def wrap(name, nin):
            orig = getattr(hip, name)

            def f(*a, **k):
                pre = [cks(t) for t in a[:nin] if torch.is_tensor(t)]
                r = orig(*a, **k)
                outs = r if isinstance(r, tuple) else (r,)
                post_in = [cks(t) for t in a[:nin] if torch.is_tensor(t)]
                trace[-1].append((name, pre, post_in, [cks(t) for t in outs if torch.is_tensor(t)]))
                return r
            setattr(hip, name, f)
        for nm, nin in (('conv3x3s2_wgrad_small', 2), ('bias_act_bwd', 2), ('conv3x3s2_dgrad_small', 2), ('conv3x3s2_dgrad_lead', 2)):
            wrap(nm, nin)
    for it in range(4):
        trace.append([])
        model = gold.build()
        sylvester.sync_bn(None, sync)
        sylvester._SYNC['log'] = []
        try:
            out = gold.forward(model, x, nz)
            err, kl = gold.aggregate(out[1])
            (err + kl).backward()
        finally:
            sylvester.sync_bn(None, False)
        logs.append(sylvester._SYNC['log'])
        torch.cuda.synchronize()
        print('rank %d run %d elbo %.6f err0 %.6f' % (rank, it, float(err + kl), float(out[1]['err'][0])), flush=True)
        runs.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        if os.environ.get('BARRIER') == '1':
            dist.barrier()
    big = max(float(v.double().norm()) for v in runs[0].values())
    for it in range(1, 4):
        worst = sorted(((float((runs[it][n].double() - runs[0][n].double()).norm()) / (float(runs[0][n].double().norm()) + 1e-6 * big), n)
                        for n in runs[0]), reverse=True)[:3]
        print('rank %d run %d vs run 0: %s' % (rank, it, ' '.join('%s %.2e' % (n, e) for e, n in worst)), flush=True)
    if os.environ.get('TRACE') == '1':
        for it in range(1, 4):
            for i, (a, b) in enumerate(zip(trace[0], trace[it])):
                if a != b:
                    what = [w for w, u, v in (('operands before', a[1], b[1]), ('operands after', a[2], b[2]), ('results', a[3], b[3])) if u != v]
                    print('rank %d run %d vs run 0: traced call #%d %s differs in: %s\n      run 0: %s\n      run %d: %s' % (
                        rank, it, i, a[0], ', '.join(what), a[1:], it, b[1:]), flush=True)
                    break
    if os.environ.get('ALL') == '1':        # every parameter whose gradient differs between run 0 and run 1, in module order
        # (the reference run: the one most others agree with)
        ref = max(range(4), key=lambda i: sum(all(torch.equal(runs[i][n], runs[j][n]) for n in runs[0]) for j in range(4)))
        for it in range(4):
            for n in runs[0]:
                a, b = runs[ref][n].double(), runs[it][n].double()
                if not torch.equal(a, b):
                    d = (a - b).abs()
                    print('rank %d run %d (vs run %d)  %-48s rel %.2e  differing %d of %d  max |d| %.3e at %s' % (
                        rank, it, ref, n, float(d.norm() / (a.norm() + 1e-30)), int((d > 0).sum()), d.numel(), float(d.max()),
                        tuple(int(v) for v in torch.unravel_index(d.argmax(), d.shape))), flush=True)
                    if int((d > 0).sum()) <= 512 and d.dim() == 4:       # a sparse set: which (co, ci, kh, kw)?
                        idx = (d > 0).nonzero()
                        print('      co %s\n      ci %s\n      kh,kw %s\n      values ref %s\n      values run %s' % (
                            sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist())),
                            sorted(set((int(u), int(v)) for u, v in idx[:, 2:].tolist())),
                            [round(float(v), 5) for v in a[d > 0][:8]], [round(float(v), 5) for v in b[d > 0][:8]]), flush=True)
    for it in range(1, 4):
        for i, (a, b) in enumerate(zip(logs[0], logs[it])):
            if a != b:
                print('rank %d run %d: first differing exchange #%d of %d: %s' % (rank, it, i, len(logs[0]), (a, b)), flush=True)
                break
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)
