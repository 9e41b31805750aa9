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
    for it in range(4):
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
    for it in range(1, 4):
        for i, (a, b) in enumerate(zip(logs[0], logs[it])):
            if a != b:
                print('rank %d run %d: first differing exchange #%d of %d: %s' % (rank, it, i, len(logs[0]), (a, b)), flush=True)
                break
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)
