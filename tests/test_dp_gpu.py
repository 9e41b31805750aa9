"""N>1 training step on real HIP kernels: two ranks (two processes sharing the one GPU of the test box, `gloo` carrying
the device bucket -- RCCL refuses two ranks on one device) each run TrainStep in its multi-rank launch mode
(forward+backward graph | all-reduce of the flat gradient bucket | GECO+Adam graph) on half of the batch.  After every
step both ranks must hold bit-identical parameters and GECO state, and these must agree with a single process
stepping on the full batch (the gradient of a batch mean is the mean of the shard gradients; fp32 summation order
differs, hence a tolerance)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(case='tiny'):
    from tests.common import Golden
    from tests.test_model_gpu import build
    gold = Golden(case)
    x, _, _ = gold.inputs()
    noise = [gold.noise(1 + it) for it in range(STEPS)]
    return gold, build(gold), x, noise


def _run(ts, x, noise, sl, graph_inputs=False):
    out = []
    for it in range(STEPS):
        rp, eps = noise[it]
        o = ts.step(x[sl].cuda(), rand_pixel=rp[sl].cuda(), eps=torch.stack(eps)[:, sl].contiguous().cuda())
        out.append(o.clone())
    return torch.stack(out)


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from genesis_amd.trainer import TrainStep
    gold, model, x, noise = _make()
    B = x.shape[0]
    sl = slice(rank * B // world, (rank + 1) * B // world)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    assert ts.world == world
    hist = _run(ts, x, noise, sl)
    # the graph-replay launch mode on the same ranks: a few more steps with internally drawn noise must keep the ranks
    # in lock-step as well (identical parameters after identical all-reduced gradients)
    ts.use_graph = True
    torch.manual_seed(100 + rank)
    for _ in range(3):
        ts.step(x[sl].cuda())
    assert ts._split and ts.graph2 is not None
    torch.save({'hist': hist.cpu(), 'p': ts.flat_p.cpu(), 'p64': ts.flat_p64.cpu(), 'geco': ts.geco.state.cpu(),
                'p_after_eager': None}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def _worker_eager_only(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from genesis_amd.trainer import TrainStep
    gold, model, x, noise = _make()
    B = x.shape[0]
    sl = slice(rank * B // world, (rank + 1) * B // world)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    hist = _run(ts, x, noise, sl)
    torch.save({'hist': hist.cpu(), 'p': ts.flat_p.cpu(), 'p64': ts.flat_p64.cpu(), 'geco': ts.geco.state.cpu()},
               os.path.join(out_dir, 'e%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_match_full_batch_and_each_other(tmp_path):
    world = 2
    mp.spawn(_worker_eager_only, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    e0 = torch.load(os.path.join(str(tmp_path), 'e0.pt'))
    e1 = torch.load(os.path.join(str(tmp_path), 'e1.pt'))
    for k in ('hist', 'p', 'p64', 'geco'):
        assert torch.equal(e0[k], e1[k]), k                       # identical update on every rank
    # single process, full batch
    from genesis_amd.trainer import TrainStep
    gold, model, x, noise = _make()
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    ref = _run(ts, x, noise, slice(0, x.shape[0])).cpu()
    # [elbo, err, kl, beta] per step: batch means agree to fp32 round-off; Adam's first steps amplify gradient noise
    # into lr-sized parameter differences (see test_train_gpu), hence the absolute floor on the small KL term
    assert torch.allclose(e0['hist'][:, :2], ref[:, :2], rtol=2e-4), (e0['hist'], ref)
    assert torch.allclose(e0['hist'][:, 3], ref[:, 3], rtol=1e-5)
    assert torch.allclose(e0['hist'][:, 2], ref[:, 2], rtol=5e-3, atol=2e-5 * float(ref[0, 0].abs()))
    rel = float((e0['p'] - ts.flat_p.cpu()).norm() / ts.flat_p.cpu().norm())
    assert rel < 1e-4, rel


@pytest.mark.timeout(600)
def test_two_ranks_graph_mode_stays_in_lock_step(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), 'r0.pt'))
    r1 = torch.load(os.path.join(str(tmp_path), 'r1.pt'))
    for k in ('p', 'p64', 'geco'):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.isfinite(r0['p']).all()


def _worker_diverged_start(rank, world, port, out_dir):
    """Ranks build their models from DIFFERENT seeds (and rank 1 perturbs its GECO state): TrainStep must start every
    rank from rank 0's parameters, optimiser and GECO state (what DistributedDataParallel's construction broadcast does)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    from genesis_amd.trainer import TrainStep
    from genesis_amd.geco import make_geco
    from oracle import v2_oracle as O
    cfg = AttrDict(dict(O.make_cfg(K_steps=3, img_size=32, feat_dim=8), debug=False, multi_gpu=False))
    torch.manual_seed(1000 + rank)                     # different initial weights per rank
    model = G.load(cfg).cuda().train()
    geco = make_geco(32)
    if rank == 1:
        geco.beta = 3.0
    before = torch.cat([p.detach().flatten().float() for p in model.parameters()]).cpu()
    ts = TrainStep(model, 32, geco=geco, graph=False)
    torch.manual_seed(7 + rank)
    x = torch.rand(2, 3, 32, 32, device='cuda')
    ts.step(x)
    torch.save({'before': before, 'p': ts.flat_p.cpu(), 'p64': ts.flat_p64.cpu(), 'geco': ts.geco.state.cpu(),
                'm': ts.m32.cpu()}, os.path.join(out_dir, 'd%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ranks_with_different_seeds_start_from_rank0(tmp_path):
    world = 2
    mp.spawn(_worker_diverged_start, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d0 = torch.load(os.path.join(str(tmp_path), 'd0.pt'))
    d1 = torch.load(os.path.join(str(tmp_path), 'd1.pt'))
    assert not torch.equal(d0['before'], d1['before'])            # the ranks really did start apart
    for k in ('p', 'p64', 'geco', 'm'):
        assert torch.equal(d0[k], d1[k]), k                        # ... and are in lock-step after the first step


def _worker_monet(rank, world, port, out_dir):
    """BASELINE config 4 is MONet on 2 GPUs: the same lock-step property for MONet (recurrent shared-weight UNet:
    weight gradients accumulated over K-1 passes go through the same bucket)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from genesis_amd.trainer import TrainStep
    from tests.test_monet_oracle import MonetGolden
    from tests.test_monet_gpu import build
    gold = MonetGolden('tiny')
    model = build(gold)
    x, _ = gold.inputs()
    B = x.shape[0]
    sl = slice(rank * B // world, (rank + 1) * B // world)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    hist = []
    K = gold.cfg['K_steps']
    for it in range(2):
        _, eps = gold.inputs(1 + it)
        e = eps.view(K, B, -1)[:, sl].reshape(-1, eps.shape[-1]).contiguous()
        hist.append(ts.step(x[sl].cuda(), eps=e.cuda()).cpu())
    torch.save({'hist': torch.stack(hist), 'p': ts.flat_p.cpu(), 'geco': ts.geco.state.cpu()},
               os.path.join(out_dir, 'm%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_monet_matches_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker_monet, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m0 = torch.load(os.path.join(str(tmp_path), 'm0.pt'))
    m1 = torch.load(os.path.join(str(tmp_path), 'm1.pt'))
    for k in ('hist', 'p', 'geco'):
        assert torch.equal(m0[k], m1[k]), k
    from genesis_amd.trainer import TrainStep
    from tests.test_monet_oracle import MonetGolden
    from tests.test_monet_gpu import build
    gold = MonetGolden('tiny')
    model = build(gold)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    x, _ = gold.inputs()
    ref = []
    for it in range(2):
        _, eps = gold.inputs(1 + it)
        ref.append(ts.step(x.cuda(), eps=eps.cuda()).cpu())
    ref = torch.stack(ref)
    assert torch.allclose(m0['hist'][:, :2], ref[:, :2], rtol=1e-3), (m0['hist'], ref)
    rel = float((m0['p'] - ts.flat_p.cpu()).norm() / ts.flat_p.cpu().norm())
    assert rel < 1e-3, rel


def _worker_log_mse(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from genesis_amd.trainer import TrainStep
    gold, model, x, noise = _make()
    B = x.shape[0]
    sl = slice(rank * B // world, (rank + 1) * B // world)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False, log_mse=True)
    hist = _run(ts, x, noise, sl)
    torch.save({'hist': hist.cpu()}, os.path.join(out_dir, 'm%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_log_the_whole_batchs_mse(tmp_path):
    """train.py:244-246 logs mse / rmse over the whole batch (nn.DataParallel gathers the reconstructions): with one process
    per GPU the ranks' shard means ride the step's one all-reduce next to err / kl, so every rank reports the same, global
    numbers (advisor finding, round 4: they used to be rank-local while elbo / err / kl in the same vector were global)."""
    world = 2
    mp.spawn(_worker_log_mse, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m0 = torch.load(os.path.join(str(tmp_path), 'm0.pt'))['hist']
    m1 = torch.load(os.path.join(str(tmp_path), 'm1.pt'))['hist']
    assert m0.shape[1] == 6 and torch.equal(m0, m1)
    from genesis_amd.trainer import TrainStep
    gold, model, x, noise = _make()
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False, log_mse=True)
    ref = _run(ts, x, noise, slice(0, x.shape[0])).cpu()
    assert torch.allclose(m0[:, 4:6], ref[:, 4:6], rtol=2e-4), (m0[:, 4:6], ref[:, 4:6])
    assert float(ref[0, 4]) > 0 and float(ref[0, 5]) > float(ref[0, 4])      # rmse > mse for errors below 1


def _worker_genesis_syncbn(rank, world, port, out_dir):
    """GENESIS (BASELINE config 3) with its batch sharded over two ranks and GENESIS_SYNC_BN=1: BatchNorm statistics and
    their backward sums over BOTH shards (genesis_amd/sylvester.sync_bn) -- the reference's single-device run at the global batch
    (models/genesis_config.py:39-40, third_party/sylvester/layers.py:26-27), i.e. tests/golden/full_genesis_cfg3_b32.npz."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['GENESIS_SYNC_BN'] = '1'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import numpy as np
    from genesis_amd import sylvester
    from genesis_amd.trainer import TrainStep
    from tests.test_fullbatch_gpu import Full, FWD_TOL, check_gradients
    gold = Full('genesis_cfg3_b32')
    K, B = gold.K, gold.B
    sl = slice(rank * B // world, (rank + 1) * B // world)

    def shard(nz):
        return dict(eps_m=[n[sl].contiguous().cuda() for n in nz[:K]],
                    eps_c=nz[K].view(K, B, -1)[:, sl].reshape(-1, nz[K].shape[-1]).contiguous().cuda())

    x = gold.x()
    # --- one forward + backward on the shards, batch statistics over both ranks
    model = gold.build()
    sylvester.sync_bn(None, True)
    try:
        recon, losses, stats, att, comp = model(x[sl].cuda(), **shard(gold.noise()))
        err, kl = gold.aggregate(losses)
        (err + kl).backward()
    finally:
        sylvester.sync_bn(None, False)
    # per-image loss terms of this rank's images against the reference's full-batch run
    rt, at = FWD_TOL['genesis']['err']
    np.testing.assert_allclose(losses['err'].detach().cpu().numpy(), gold.g['out/err'][sl], rtol=rt, atol=at)
    rt, at = FWD_TOL['genesis']['kl_l_k']
    np.testing.assert_allclose(torch.stack(list(losses['kl_l_k'])).detach().cpu().numpy(), gold.g['out/kl_l_k'][:, sl], rtol=rt, atol=at)
    np.testing.assert_allclose(torch.stack(list(losses['kl_m_k'])).detach().cpu().numpy(), gold.g['out/kl_m_k'][:, sl], rtol=rt, atol=at)
    tot = torch.stack([err.detach(), kl.detach()]).cpu()
    dist.all_reduce(tot)
    tot /= world
    elbo_ref = float(gold.g['loss/err']) + float(gold.g['loss/kl'])
    assert abs(float(tot.sum()) - elbo_ref) <= 1e-4 * abs(elbo_ref), (tot, elbo_ref)
    grads = {}
    for n, p in model.named_parameters():
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu()
        dist.all_reduce(g)
        grads[n] = (g / world).to(p.device)
    # (GENESIS has no ReLU: the strict bar.  Until the build's pk_peephole pass this check failed in some launches: two processes
    #  sharing one GPU is exactly the condition under which the packed-fp32 operand selection of DESIGN.md finding 48 misreads a
    #  register -- 1e-3 .. 1e-2 in a handful of gradients.)
    if rank == 0:
        check_gradients(gold, grads, [], 'genesis_cfg3_b32 on two ranks, cross-replica BatchNorm')
    del model
    # --- three training steps through TrainStep (GENESIS_SYNC_BN=1 arms the same switch per iteration)
    ts = TrainStep(gold.build(), gold.S, lr=1e-4)
    assert ts._sync_bn and ts.world == world
    hist = gold.g['train_hist']
    xs = x[sl].cuda()
    for it in range(3):
        out = ts.step(xs, **shard(gold.noise(1 + it))).cpu().numpy()
        elbo, err_, kl_, beta = [float(v) for v in out]
        tol = 1e-4 if it == 0 else 5e-4
        assert abs(elbo - hist[it, 0]) <= tol * abs(hist[it, 0]), (it, out, hist[it])
        np.testing.assert_allclose([err_, beta], hist[it, [1, 3]], rtol=5e-4)
    assert abs(float(ts.geco.beta) - float(gold.g['train_beta_final'])) <= 1e-5 * float(gold.g['train_beta_final'])
    bufs = torch.cat([b.detach().double().flatten().cpu() for b in ts.model.buffers()])
    torch.save({'p': ts.flat_p.cpu(), 'bufs': bufs, 'geco': ts.geco.state.cpu()}, os.path.join(out_dir, 'g%d.pt' % rank))
    ts.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_genesis_with_cross_replica_batchnorm_equals_the_reference_at_global_batch(tmp_path):
    world = 2
    mp.spawn(_worker_genesis_syncbn, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(str(tmp_path), 'g0.pt'))
    g1 = torch.load(os.path.join(str(tmp_path), 'g1.pt'))
    for k in ('p', 'bufs', 'geco'):
        assert torch.equal(g0[k], g1[k]), k          # lock-step, running statistics included
