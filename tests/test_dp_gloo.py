"""N>1 path on CPU: world_size-2 `gloo` run of the data-parallel bucket (genesis_amd/dp.py).  Each rank
computes the oracle's gradients on its shard of the batch; after ONE all-reduce of the flat bucket every
rank must hold (a) the full-batch gradient and (b) the global batch-mean err / KL in the bucket tail --
what the GECO + Adam update consumes -- identical on both ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import v2_oracle as O
from genesis_amd import testing as T
from genesis_amd.dp import FlatBucket

CFG = O.make_cfg(K_steps=3, img_size=32, feat_dim=8)
B = 4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _local_grads(x, rp, eps):
    sd = T.formula_state_dict(O.template_state_dict(CFG))
    params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
    bucket = FlatBucket(params, n_tail=2)
    p = dict(zip(sd.keys(), params))
    bucket.zero_grad()
    _, losses, _, _, _ = O.v2_forward(p, x, CFG, rp, eps, reference_form=False)
    err, kl, _ = O.aggregate_losses(losses)
    (err + kl).backward()
    assert bucket.grads_in_bucket()
    bucket.set_tail(err, kl)
    return bucket


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    x = T.make_input(3, B, 32)
    rp, eps = T.draw_noise(4, B, 32, 8, 3)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    bucket = _local_grads(x[sl], rp[sl], [e[sl] for e in eps])
    scale = bucket.all_reduce()
    torch.save({'g': bucket.flat_g * scale, 'g64': bucket.flat_g64 * scale}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_bucket_equals_full_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), 'r0.pt'))
    r1 = torch.load(os.path.join(str(tmp_path), 'r1.pt'))
    assert torch.equal(r0['g'], r1['g']) and torch.equal(r0['g64'], r1['g64'])   # identical update on all ranks
    x = T.make_input(3, B, 32)
    rp, eps = T.draw_noise(4, B, 32, 8, 3)
    full = _local_grads(x, rp, eps)
    ref = full.flat_g.detach()
    got = r0['g']
    n32 = full.n32
    rel = float((got[:n32] - ref[:n32]).norm() / ref[:n32].norm())
    assert rel < 1e-4, rel
    # global err, kl right behind the parameters' gradients (the fp64 gradient's (hi, mid, lo) triple follows them)
    np.testing.assert_allclose(got[n32:n32 + 2].numpy(), ref[n32:n32 + 2].numpy(), rtol=1e-5)
    np.testing.assert_allclose(r0['g64'].numpy(), full.flat_g64.detach().numpy(), rtol=1e-4, atol=1e-7)


def test_bucket_views_track_parameters():
    params = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)),
              torch.nn.Parameter(torch.randn((), dtype=torch.float64))]
    before = [p.detach().clone() for p in params]
    b = FlatBucket(params, n_tail=2)
    assert b.n32 == 32 and b.n64 == 1 and b.flat_g.numel() == 37    # 12 -> 16, 5 -> 16 (64-byte aligned slots); tail: err, kl, (hi, mid, lo)
    assert all(p.data_ptr() % 64 == 0 for p in params[:2])
    for p, q in zip(params, before):
        assert torch.equal(p.detach(), q)
    (params[0].sum() * 2 + params[1].sum() * 3 + params[2] * 4).backward()
    assert b.grads_in_bucket()
    assert torch.equal(b.flat_g[:12], torch.full((12,), 2.0)) and torch.equal(b.flat_g[16:21], torch.full((5,), 3.0))
    assert float(b.flat_g[12:16].abs().sum()) == 0.0 and float(b.flat_g[21:32].abs().sum()) == 0.0
    assert float(b.flat_g64[0]) == 4.0
    b.flat_p.mul_(0)          # the optimiser writes the flat buffer; parameters are views of it
    assert float(params[0].abs().sum()) == 0.0


def test_fp64_gradient_rides_the_fp32_tail_exactly_for_one_rank():
    """ONE collective per step: the fp64 gradient (att_process.log_sigma) travels as a (hi, mid, lo) float triple in the fp32
    bucket's tail; for a single contribution the round trip is exact."""
    params = [torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn((), dtype=torch.float64))]
    buf = torch.arange(5, dtype=torch.float32)
    b = FlatBucket(params, n_tail=2, mean_buffers=[buf, torch.zeros((), dtype=torch.int64)])
    assert b.n_buf == 5 and b.flat_g.numel() == b.n32 + 2 + 3 + 5
    g = torch.tensor([0.1234567890123456789], dtype=torch.float64) * 3.0
    b.flat_g64[:1].copy_(g)
    b.pack64()
    b.flat_g64.zero_(); buf.mul_(0)
    b.unpack64(1.0)
    assert torch.equal(b.flat_g64[:1], g)
    assert torch.equal(buf, torch.arange(5, dtype=torch.float32))


def _worker_pieces(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(10 + rank)
    params = [torch.nn.Parameter(torch.randn(*sh)) for sh in ((5, 3), (7,), (4, 4, 3, 3), (9,), (2, 33))]
    params.append(torch.nn.Parameter(torch.randn((), dtype=torch.float64)))
    bucket = FlatBucket(params, n_tail=2)
    g = torch.Generator().manual_seed(20 + rank)
    bucket.flat_g.copy_(torch.randn(bucket.flat_g.shape, generator=g))
    bucket.flat_g64.copy_(torch.randn(bucket.flat_g64.shape, generator=g, dtype=torch.float64))
    whole = FlatBucket([torch.nn.Parameter(p.detach().clone()) for p in params], n_tail=2)
    whole.flat_g.copy_(bucket.flat_g)
    whole.flat_g64.copy_(bucket.flat_g64)
    rng = bucket.param_range(params[2:4])                # two neighbours: one contiguous slice
    assert rng is not None and rng[1] - rng[0] == bucket._pad(144) + bucket._pad(9)
    assert bucket.param_range([params[0], params[2]]) is None            # not back to back
    assert bucket.param_range([params[5]]) is None                       # fp64: rides the tail
    bucket.all_reduce_range(rng[0], rng[1])              # the early piece ...
    s1 = bucket.all_reduce(done=rng)                     # ... and the rest
    s2 = whole.all_reduce()
    assert s1 == s2 == 1.0 / world
    torch.save({'pieces': bucket.flat_g.clone(), 'whole': whole.flat_g.clone(), 'p64': bucket.flat_g64.clone(),
                'w64': whole.flat_g64.clone()}, os.path.join(out_dir, 'p%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucket_reduced_in_pieces_equals_one_collective(tmp_path):
    """TrainStep's early flush (GENESIS_WGQ_EARLY_FLUSH=1) sends the decoder's slice of the bucket first and the two remaining
    slices at the end of the backward: FlatBucket.param_range / all_reduce_range / all_reduce(done=...) on two gloo ranks must
    leave exactly what the single collective leaves (the same element-wise sums), fp64 triples and tail included."""
    port = _free_port()
    mp.spawn(_worker_pieces, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'p%d.pt' % r)) for r in (0, 1))
    for r in (r0, r1):
        assert torch.equal(r['pieces'], r['whole']) and torch.equal(r['p64'], r['w64'])
    assert torch.equal(r0['pieces'], r1['pieces'])


def _worker8(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1000 + rank)
    params = [torch.nn.Parameter(torch.zeros(33, 7)), torch.nn.Parameter(torch.zeros(5)),
              torch.nn.Parameter(torch.zeros((), dtype=torch.float64)), torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))]
    bn_mean, bn_var = torch.randn(6, generator=g), torch.rand(6, generator=g) + 0.5
    tracked = torch.tensor(rank, dtype=torch.int64)           # integer buffer: not averaged
    b = FlatBucket(params, n_tail=2, mean_buffers=[bn_mean, bn_var, tracked])
    b.zero_grad()
    b.flat_g[:b.n32].copy_(torch.randn(b.n32, generator=g))
    g64 = torch.randn(4, dtype=torch.float64, generator=g) * 1e3 + 0.123456789012345
    b.flat_g64[:4].copy_(g64)
    err, kl = torch.tensor(8000.0 + rank), torch.tensor(20.0 + 0.5 * rank)
    b.set_tail(err, kl)
    local = {'g': b.flat_g[:b.n32].clone(), 'g64': g64.clone(), 'mean': bn_mean.clone(), 'var': bn_var.clone()}
    # rank 0's state wins at start-up (TrainStep.sync_from_rank0)
    state = torch.full((3,), float(rank))
    b.flat_p.fill_(float(rank))
    b.broadcast_state([state])
    assert float(b.flat_p.abs().max()) == 0.0 and float(state.abs().max()) == 0.0
    if rank % 2:          # the two forms of the exchange: inside all_reduce, or packed by the caller (the graph-replay path)
        scale = b.all_reduce()
    else:
        b.pack64()
        scale = b.all_reduce(packed=True)
        b.unpack64(scale)
    torch.save({'local': local, 'g': b.flat_g[:b.n32] * scale, 'tail': b.tail(scale), 'g64': b.flat_g64[:4] * scale,
                'mean': bn_mean, 'var': bn_var, 'tracked': tracked, 'scale': scale}, os.path.join(out_dir, 'w%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_bucket_exchange(tmp_path):
    """The 8-GPU launch's exchange step, rehearsed over gloo with 8 processes: ONE all-reduce of the flat bucket delivers to
    every rank the same mean gradient, the global batch-mean err / KL (GECO's input: identical multiplier on all ranks), the
    fp64 gradients (as (hi, mid, lo) float triples: 1e-7 relative after an 8-way fp32 sum of the hi parts) and the
    rank-averaged BatchNorm running statistics; integer buffers stay local; rank 0's state wins the start-up broadcast."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'w%d.pt' % k)) for k in range(world)]
    for k in range(1, world):
        for key in ('g', 'tail', 'g64', 'mean', 'var'):
            assert torch.equal(r[0][key], r[k][key]), (key, k)       # bit-identical on every rank
        assert int(r[k]['tracked']) == k and r[k]['scale'] == 1.0 / world
    mean = lambda key: sum(x['local'][key].double() for x in r) / world   # noqa: E731
    np.testing.assert_allclose(r[0]['g'].double().numpy(), mean('g').numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r[0]['g64'].numpy(), mean('g64').numpy(), rtol=2e-7)
    np.testing.assert_allclose(r[0]['mean'].double().numpy(), mean('mean').numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r[0]['var'].double().numpy(), mean('var').numpy(), rtol=1e-5)
    np.testing.assert_allclose(r[0]['tail'].numpy(), [8000.0 + 3.5, 20.0 + 0.5 * 3.5], rtol=1e-6)


def test_bench_self_launch_command_line(monkeypatch):
    """`python bench.py --gpus N` without a launcher execs the driver's own N > 1 form (one rank per GPU, loopback
    rendezvous); inside a launch whose WORLD_SIZE disagrees with --gpus it refuses instead of timing the wrong thing."""
    import importlib.util
    import os
    import sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        monkeypatch.delenv(k, raising=False)
    argv = bench.self_launch_argv(4, ['--gpus', '4', '--steps', '20', '--warmup', '5'])
    assert argv[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in argv and argv[argv.index('--nproc-per-node') + 1] == '4'
    assert argv[argv.index('--master-addr') + 1] == '127.0.0.1' and int(argv[argv.index('--master-port') + 1]) > 0
    i = argv.index(os.path.join(root, 'bench.py'))
    assert argv[i + 1:] == ['--gpus', '4', '--steps', '20', '--warmup', '5']
    seen = {}

    def fake_execv(exe, args):
        seen['argv'] = args
        raise SystemExit(0)
    monkeypatch.setattr(bench.os, 'execv', fake_execv)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '3', '--warmup', '1'])
    fd1, fd2 = os.dup(1), os.dup(2)
    try:
        with pytest.raises(SystemExit):
            bench.main()
    finally:
        os.dup2(fd1, 1); os.dup2(fd2, 2); os.close(fd1); os.close(fd2)
    assert seen['argv'][seen['argv'].index('--nproc-per-node') + 1] == '8'
    monkeypatch.setenv('WORLD_SIZE', '2')
    fd1 = os.dup(1)
    try:
        with pytest.raises(SystemExit) as e:
            bench.main()
    finally:
        os.dup2(fd1, 1); os.close(fd1)
    assert 'WORLD_SIZE' in str(e.value)
