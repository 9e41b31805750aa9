"""Two processes on ONE GPU, each repeating single ops on fixed inputs: is an op's result bit-reproducible while another process
keeps the GPU busy?  Ops: the small stride-2 weight gradient (LDS tree reduction + split-K slabs), the gated unit's backward, a
torch matmul and a torch reduction (controls: not our kernels).  usage: diag_shared_gpu.py [reps]"""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, reps):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from genesis_amd import hip_ops as hip
    dev = 'cuda'
    g = torch.Generator().manual_seed(3 + rank)
    x = torch.randn(80, 32, 32, 32, generator=g).to(dev)
    dy = torch.randn(80, 32, 16, 16, generator=g).to(dev)
    yg = torch.randn(16, 64, 64, 64, generator=g).to(dev)
    dout = torch.randn(16, 32, 64, 64, generator=g).to(dev)
    gam = [torch.rand(32, generator=g).to(dev) + 0.5 for _ in range(4)]
    a = torch.randn(2048, 2048, generator=g).to(dev)
    ops = {
        'conv3x3s2_wgrad_small': lambda: hip.conv3x3s2_wgrad_small(x, dy),
        'gated_norm fwd+bwd (bn)': lambda: _gated(hip, yg, dout, gam),
        'torch matmul (control)': lambda: a @ a,
        'torch sum(dim) (control)': lambda: x.sum((0, 2, 3)),
    }
    for name, fn in ops.items():
        dist.barrier()
        ref = fn()
        torch.cuda.synchronize()
        bad = 0
        worst = 0.0
        for i in range(reps):
            out = fn()
            if (i & 15) == 15:
                dist.barrier()               # keep the two processes in step (both busy at the same time)
            if not torch.equal(out, ref):
                bad += 1
                worst = max(worst, float((out.double() - ref.double()).norm() / ref.double().norm()))
        torch.cuda.synchronize()
        print('rank %d  %-28s %d of %d repetitions differ from the first%s' % (rank, name, bad, reps, ' (worst rel %.2e)' % worst if bad else ''), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def _gated(hip, y, dout, gam):
    out, stats = hip.gated_norm_fwd(y, None, 'bn', gam[0], gam[1], gam[2], gam[3])
    r = hip.gated_norm_bwd(y, None, 'bn', gam[0], gam[1], gam[2], gam[3], stats, dout)
    return torch.cat([out.flatten(), r[0].flatten(), r[1], r[2], r[3], r[4]])


if __name__ == '__main__':
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    mp.spawn(worker, args=(2, 29613, reps), nprocs=2, join=True)
