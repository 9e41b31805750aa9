"""Training-step parity on the GPU: GECO + Adam + HIP-graph replay against the reference's own three
training steps stored in the golden fixtures (train.py:223-263 semantics), ELBO within 1e-3 relative."""
import os

import numpy as np
import pytest
import torch

from tests.common import Golden
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('case', ['tiny', 'tiny_klm', 'metric'])
def test_three_steps_vs_reference(case):
    from genesis_amd.trainer import TrainStep
    gold = Golden(case)
    model = build(gold)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    assert model.att_process.log_sigma.dtype == torch.float64
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    hist = gold.g['train_hist']
    for it in range(3):
        rp, eps = gold.noise(1 + it)
        out = ts.step(xd, rand_pixel=rp.to(DEV), eps=torch.stack(eps).to(DEV)).cpu().numpy()
        elbo, err, kl, beta = [float(v) for v in out]
        assert abs(elbo - hist[it, 0]) <= 1e-3 * abs(hist[it, 0]), (it, elbo, hist[it])   # north_star bound
        np.testing.assert_allclose([elbo, err, beta], hist[it, [0, 1, 3]], rtol=2e-4)
        # Adam's first steps move every parameter by ~lr whatever the gradient magnitude, so parameters whose
        # gradient is pure round-off noise drift differently between two fp32 implementations: the small KL
        # term is compared with an absolute floor tied to the ELBO scale
        np.testing.assert_allclose(kl, hist[it, 2], rtol=5e-3, atol=2e-5 * abs(hist[it, 0]))
    assert abs(float(ts.geco.beta) - float(gold.g['train_beta_final'])) <= 1e-5
    assert abs(float(ts.geco.err_ema) - hist[2, 4]) <= 1e-4 * abs(hist[2, 4])
    assert int(ts.step_t) == 3


def test_adam_kernel_matches_torch():
    import ctypes
    from genesis_amd import _lib
    torch.manual_seed(0)
    for dtype, is64 in ((torch.float32, 0), (torch.float64, 1)):
        p = torch.randn(10007, dtype=dtype)
        ref = p.clone().requires_grad_()
        opt = torch.optim.Adam([ref], 1e-3)
        pd = p.to(DEV)
        m, v = torch.zeros_like(pd), torch.zeros_like(pd)
        step = torch.zeros((), dtype=torch.int64, device=DEV)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for it in range(5):
            g = torch.randn(10007, dtype=dtype) * (0.1 + it)
            ref.grad = g.clone()
            opt.step()
            gd = g.to(DEV)
            _lib.call('gx_step_increment', ctypes.c_void_p(step.data_ptr()), st)
            _lib.call('gx_adam_step', ctypes.c_void_p(pd.data_ptr()), ctypes.c_void_p(gd.data_ptr()),
                      ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(v.data_ptr()), pd.numel(), is64,
                      ctypes.c_void_p(step.data_ptr()), 1e-3, 0.9, 0.999, 1e-8, 1.0, st)
        np.testing.assert_allclose(pd.cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-7)


def test_graph_replay_matches_eager():
    """The HIP-graph-captured step and the eager step run the same kernels: with identical on-device RNG
    state they must produce the same losses and parameters."""
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    outs = []
    for graph in (False, True):
        model = build(gold)
        ts = TrainStep(model, gold.S, graph=graph)
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        vals = [ts.step(xd).clone() for _ in range(6)]
        torch.cuda.synchronize()
        outs.append((torch.stack(vals).cpu(), ts.flat_p.clone().cpu(), float(ts.geco.beta)))
    (a, pa, ba), (b, pb, bb) = outs
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    # the step's noise is one counter-based launch keyed by (torch's seed, rank, the step counter) -- the same values whether
    # the launch is issued eagerly or replayed from the graph -- and eager and replay run the same kernels in the same order:
    # losses and parameters agree to fp32 round-off (measured: bit-identical)
    print('graph vs eager: max |d out| %.3e, max |d param| %.3e' % (float((a - b).abs().max()), float((pa - pb).abs().max())))
    assert torch.equal(a, b) and ba == bb and torch.equal(pa, pb)


def test_prepare_captures_without_advancing_state():
    """prepare() is set-up (graph capture): parameters, Adam moments, step counter and GECO state stay untouched."""
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    model = build(gold)
    ts = TrainStep(model, gold.S, graph=True)
    before = [t.clone() for t in (ts.flat_p, ts.flat_p64, ts.m32, ts.v32, ts.step_t, ts.geco.state)]
    ts.prepare(xd)
    assert ts.graph is not None and ts.iters == 0
    after = (ts.flat_p, ts.flat_p64, ts.m32, ts.v32, ts.step_t, ts.geco.state)
    assert all(torch.equal(a, b) for a, b in zip(before, after))
    out = ts.step(xd)
    assert torch.isfinite(out).all() and int(ts.step_t) == 1 and ts.iters == 1
    assert not torch.equal(before[0], ts.flat_p)


def test_profile_collect():
    from genesis_amd import profiling
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    model = build(gold)
    ts = TrainStep(model, gold.S)
    x, _, _ = gold.inputs()
    ts.step(x.to(DEV))
    profiling.enable(True, ts._ctx)              # profiling records belong to the loop's library context
    ts.step(x.to(DEV))
    rows = profiling.collect(ts._ctx)
    profiling.enable(False, ts._ctx)
    assert profiling.collect() == []             # nothing leaked into the default context
    names = {r['name'] for r in rows}
    assert {'tapconv_kernel<0>', 'wgrad_kernel<0>', 'gn_relu_fwd_kernel', 'icsbp_fwd_kernel', 'adam_kernel'} <= names
    assert all(r['ms'] > 0 for r in rows)


def test_weight_cache_matches_per_call_packing():
    """The packed-weight cache (one batched re-pack per iteration) must not change a single bit of the training
    trajectory relative to per-call packing.  Deferred, batched parameter-gradient reductions launch every layer's weight
    gradient in ONE stream-K grid, which cuts the pixel range of a layer at other places than a launch of that layer
    alone does: the fp32 summation order differs, so that comparison is to round-off (and each mode is bit-reproducible)."""
    from genesis_amd import _lib
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    outs = {}
    for cache, defer in ((False, True), (True, True), (True, True), (False, False)):
        model = build(gold)
        ts = TrainStep(model, gold.S, lr=1e-4, graph=False, weight_cache=cache, defer_reduces=defer)
        res = []
        for it in range(4):
            rp, eps = gold.noise(1 + it % 3)
            res.append(ts.step(xd, rand_pixel=rp.to(DEV), eps=torch.stack(eps).to(DEV)).clone())
        out = (torch.stack(res), ts.flat_p.clone())
        if (cache, defer) in outs:                      # the same mode twice: bit-reproducible
            assert torch.equal(outs[(cache, defer)][0], out[0]) and torch.equal(outs[(cache, defer)][1], out[1])
        outs[(cache, defer)] = out
        if cache:
            assert _lib.query('gx_weight_cache_size', ts._wcache) > 10
    assert torch.equal(outs[(False, True)][0], outs[(True, True)][0])
    assert torch.equal(outs[(False, True)][1], outs[(True, True)][1])
    a, b = outs[(True, True)], outs[(False, False)]
    assert torch.allclose(a[0][:, :2], b[0][:, :2], rtol=2e-5), (a[0], b[0])
    assert float((a[1] - b[1]).norm() / b[1].norm()) < 2e-5


def test_rccl_allreduce_in_the_step_matches_single_graph(monkeypatch):
    """Multi-rank launch modes on one GPU (world_size 1 process group, collective forced): (a) the RCCL all-reduce of the
    bucket captured INSIDE the step's one HIP graph (first choice), (b) forward+backward graph | RCCL all-reduce | GECO+Adam
    graph (GENESIS_GRAPH_ALLREDUCE=0, and the fallback when the capture is refused).  Both must reproduce the
    single-graph trajectory bit for bit."""
    import torch.distributed as dist
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)

    def run(ts):
        torch.manual_seed(7)                      # the model draws rand_pixel / eps from the device RNG
        torch.cuda.manual_seed(7)
        return torch.stack([ts.step(xd).clone() for _ in range(4)]), ts.flat_p.clone()

    ref = run(TrainStep(build(gold), gold.S, lr=1e-4, graph=True))
    monkeypatch.setenv('GENESIS_FORCE_ALLREDUCE', '1')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1)
    try:
        ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=True)
        got_in = run(ts)
        in_graph = ts.collective_in_graph
        print('collective captured inside the graph:', in_graph, getattr(ts, 'capture_fallback_reason', ''))
        assert in_graph or (ts._split and ts.graph2 is not None)
        monkeypatch.setenv('GENESIS_GRAPH_ALLREDUCE', '0')
        ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=True)
        got_split = run(ts)
        assert ts._split and ts.graph2 is not None and not ts.collective_in_graph
    finally:
        dist.destroy_process_group()
    for got in (got_in, got_split):
        assert torch.equal(ref[0], got[0])
        assert torch.equal(ref[1], got[1])


def test_cabi_allreduce_in_the_step_matches_single_graph(monkeypatch):
    """The same check with the collective going through the library's own entry points (gx_allreduce_unique_id / _init / _run /
    _destroy: RCCL resolved by libgenesis_hip.so, no torch.distributed in the data path; the process group only carries the
    128-byte id): a world-1 communicator, the all-reduce inside the step, bit-identical to the single-graph trajectory; and
    the raw entry points on a plain buffer."""
    import ctypes
    import torch.distributed as dist
    from genesis_amd import _lib
    from genesis_amd.dp import CabiAllReduce
    from genesis_amd.trainer import TrainStep
    comm = CabiAllReduce()
    buf = torch.arange(1000, dtype=torch.float32, device=DEV) * 0.25
    want = buf.clone()
    comm.run(buf)
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    comm.close()
    with pytest.raises(_lib.GenesisHipError):
        _lib.call('gx_allreduce_init', ctypes.create_string_buffer(128), 128, 3, 2, ctypes.byref(ctypes.c_void_p()))
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)

    def run(ts):
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        return torch.stack([ts.step(xd).clone() for _ in range(4)]), ts.flat_p.clone()

    ref = run(TrainStep(build(gold), gold.S, lr=1e-4, graph=True))
    monkeypatch.setenv('GENESIS_FORCE_ALLREDUCE', '1')
    monkeypatch.setenv('GENESIS_CABI_ALLREDUCE', '1')
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29534', rank=0, world_size=1)
    try:
        ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=True)
        got = run(ts)
        assert ts.bucket._cabi is not None and ts.bucket._cabi.world == 1
        print('collective captured inside the graph:', ts.collective_in_graph, getattr(ts, 'capture_fallback_reason', ''))
        ts.bucket._cabi.close()
    finally:
        dist.destroy_process_group()
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])


def _early_flush_runs(monkeypatch, in_graph=False):
    import torch.distributed as dist
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)

    def run(ts, n=4):
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        return torch.stack([ts.step(xd).clone() for _ in range(n)]), ts.flat_p.clone()

    ref = run(TrainStep(build(gold), gold.S, lr=1e-4, graph=True))
    monkeypatch.setenv('GENESIS_WGQ_EARLY_FLUSH', '1')
    ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=True)
    assert ts._early_range is not None and 0 < ts._early_range[1] - ts._early_range[0] < ts.n32
    early = run(ts)
    # (the KL of this tiny model is ~2 on an ELBO of 2000: its fifth digit moves with the gradients' summation order)
    # (... and after Adam's first, sign-like steps the fourth: an absolute floor tied to the ELBO's scale, as in
    #  test_three_steps_vs_reference)
    assert torch.allclose(early[0], ref[0], rtol=2e-4, atol=2e-6 * float(ref[0][0, 0])), (early[0], ref[0])
    # (parameters: four Adam steps of lr 1e-4 -- a parameter whose gradient is round-off noise steps by +-lr whatever its size,
    #  and the two-flush partition sums that noise in another order: measured 6.5e-5 of the parameter norm = ~1000 such parameters)
    assert float((early[1] - ref[1]).norm() / ref[1].norm()) < 2e-4
    monkeypatch.setenv('GENESIS_FORCE_ALLREDUCE', '1')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % (29536 if in_graph else 29535), rank=0, world_size=1)
    try:
        ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=in_graph)
        got = run(ts)
        assert ts._early_side is not None            # the decoder range travelled on the second stream
        assert bool(getattr(ts, 'collective_in_graph', False)) == in_graph, getattr(ts, 'capture_fallback_reason', '')
    finally:
        dist.destroy_process_group()
    return early, got


def test_early_decoder_flush_and_its_collective(monkeypatch):
    """GENESIS_WGQ_EARLY_FLUSH=1: the decoder's queued weight gradients get a stream-K launch of their own at the end of the
    decoder's backward, and with a collective in the (eager) step that range of the bucket is all-reduced on a second stream while
    the encoder's backward runs.  The two-flush trajectory equals the single-flush one up to fp32 summation order (a stream-K
    launch cuts its tile line by what else is in the launch); with the (world-1) RCCL collective split in three it must equal the
    two-flush trajectory WITHOUT a collective."""
    early, got = _early_flush_runs(monkeypatch, in_graph=False)
    assert torch.allclose(early[0], got[0], rtol=2e-4, atol=2e-6 * float(early[0][0, 0])), (early[0], got[0])
    assert float((early[1] - got[1]).norm() / early[1].norm()) < 1e-5


def test_checkpoint_is_the_reference_wire_format(tmp_path):
    """TrainStep.state_dict() is the dict train.py:405-420 saves: its optimiser_state_dict loads into a genuine
    torch.optim.Adam (the reference's resume path, train.py:179-207), one torch-Adam step from there equals the next
    TrainStep step, and TrainStep.load_state_dict() resumes bit-exactly."""
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    noise = [gold.noise(1 + it) for it in range(3)]

    def kw(it):
        rp, eps = noise[it]
        return dict(rand_pixel=rp.to(DEV), eps=torch.stack(eps).to(DEV))

    ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=False)
    for it in range(2):
        ts.step(xd, **kw(it))
    ckpt = ts.state_dict(1)
    f = tmp_path / 'model.ckpt-1'
    torch.save(ckpt, f)
    ckpt = torch.load(f, map_location='cuda', weights_only=False)
    assert set(ckpt) == {'model_state_dict', 'optimiser_state_dict', 'beta', 'err_ema', 'iter_idx'}
    third = ts.step(xd, **kw(2)).clone()
    p_after = [p.detach().clone() for p in ts.model.parameters()]

    # (a) the reference's resume path: plain model + torch.optim.Adam
    model2 = build(gold)
    opt = torch.optim.Adam(model2.parameters(), 1e-4)
    model2.load_state_dict(ckpt['model_state_dict'])
    import copy
    opt.load_state_dict(copy.deepcopy(ckpt['optimiser_state_dict']))   # (torch adopts the tensors it is given)
    _, losses, _, _, _ = model2(xd, **kw(2))
    err = losses.err.mean(0)
    kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
    opt.zero_grad()
    (err + float(ckpt['beta']) * kl).backward()
    opt.step()
    assert abs(float((err + kl).detach()) - float(third[0])) <= 1e-5 * abs(float(third[0]))
    for a, b in zip(model2.parameters(), p_after):
        assert torch.allclose(a.detach(), b, rtol=1e-4, atol=2e-6)

    # (b) round trip through TrainStep.load_state_dict: bit-exact continuation
    ts2 = TrainStep(build(gold), gold.S, lr=1e-4, graph=False)
    assert ts2.load_state_dict(ckpt) == 2
    third2 = ts2.step(xd, **kw(2))
    assert torch.equal(third, third2)
    for a, b in zip(ts2.model.parameters(), p_after):
        assert torch.equal(a.detach(), b)


def test_two_training_loops_coexist():
    """Each TrainStep owns a library context (gx_ctx_*): its deferred-reduction queues, queued weight-gradient jobs,
    packed-weight cache bookkeeping and step switches.  Two loops stepped alternately -- one of them in graph-replay
    mode -- must produce exactly the trajectories they produce alone."""
    from genesis_amd.trainer import TrainStep
    from tests.common import Golden
    from tests.test_model_gpu import build

    def traj(alone):
        ga, gb = Golden('tiny'), Golden('tiny_b3k3')
        tsa = TrainStep(build(ga), ga.S, lr=1e-4, graph=False)
        tsb = TrainStep(build(gb), gb.S, lr=1e-4, graph=True)
        assert tsa._ctx != tsb._ctx and tsa._ctx > 0 and tsb._ctx > 0
        xa, xb = ga.inputs()[0].cuda(), gb.inputs()[0].cuda()
        outs_a, outs_b = [], []

        def step_a(it):
            rp, eps = ga.noise(1 + it)
            outs_a.append(tsa.step(xa, rand_pixel=rp.cuda(), eps=torch.stack(eps).cuda()).clone())

        def step_b(it):
            torch.manual_seed(50 + it); torch.cuda.manual_seed(50 + it)
            outs_b.append(tsb.step(xb).clone())
        if alone:
            for it in range(3):
                step_a(it)
            for it in range(3):
                step_b(it)
        else:
            for it in range(3):
                step_a(it); step_b(it)
        return torch.stack(outs_a).cpu(), torch.stack(outs_b).cpu(), tsa.flat_p.clone().cpu(), tsb.flat_p.clone().cpu()
    ref = traj(True)
    got = traj(False)
    for r, g in zip(ref, got):
        assert torch.equal(r, g)


def test_replay_after_an_aborted_eager_iteration():
    """The captured graphs hold no bucket fill (Adam zeroes the gradients it consumes).  An eager iteration that dies
    between its backward and its Adam launch leaves partial gradients in the bucket: the next replay must start from a
    clean bucket, i.e. give exactly the step a loop without the aborted iteration gives."""
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)

    def run(abort):
        ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=True)
        ts.prepare(xd)
        if abort:
            rp, eps = gold.noise(1)
            real_update = ts._update

            def dying_update(*a, **k):
                raise RuntimeError('simulated failure after the backward pass')
            ts._update = dying_update
            with pytest.raises(RuntimeError):
                ts.step(xd, rand_pixel=rp.to(DEV), eps=torch.stack(eps).to(DEV))
            ts._update = real_update
            torch.cuda.synchronize()
            assert float(ts.flat_g[:ts.n32].abs().max()) > 0        # the stale gradients are really there
        torch.manual_seed(11); torch.cuda.manual_seed(11)
        out = ts.step(xd).clone()
        torch.cuda.synchronize()
        return out.cpu(), ts.flat_p.clone().cpu(), ts.m32.clone().cpu()
    clean, dirty = run(False), run(True)
    for a, b in zip(clean, dirty):
        assert torch.equal(a, b)


def test_noise_hook_lives_only_inside_an_iteration():
    """TrainStep's Philox source is on the model only while one of its iterations runs: forwards between two steps draw
    fresh torch.rand / randn like the reference's (models/genesisv2_config.py:157, modules/attention.py:177-178) -- two
    validation forwards differ from each other and do not replay the next training step's noise --, the model pickles, and
    closing an older loop does not take a newer loop's source away."""
    import io
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    model = build(gold)
    ts = TrainStep(model, gold.S)
    assert model.noise is None and 'noise' not in model.__dict__
    ts.step(xd)
    assert model.noise is None and 'noise' not in model.__dict__
    with torch.no_grad():
        z1 = torch.stack(list(model(xd)[4]['z_k']))
        z2 = torch.stack(list(model(xd)[4]['z_k']))
    assert not torch.equal(z1, z2)                           # fresh draws per call
    torch.save(model, io.BytesIO())                          # no lambda attribute in the way
    # the step's noise is still the keyed one: two loops from the same state take the same step
    ts2 = TrainStep(model, gold.S)
    ts.close()                                               # the older loop goes away ...
    out = ts2.step(xd)                                       # ... the newer one still draws from its Philox source
    assert torch.isfinite(out).all() and model.noise is None
    ts2.close()


@pytest.mark.parametrize('kind', ['rmsprop', 'sgd'])
def test_rmsprop_and_sgd_steps_match_torch(kind, tmp_path):
    """train.py:170-176: config.optimiser 'rmsprop' -> optim.RMSprop(params, lr), 'sgd' -> optim.SGD(params, lr, 0.9).  Three
    TrainStep iterations with that optimiser against the plain loop model -> loss -> backward -> torch optimiser on the same
    kernels' gradients; the checkpoint's optimiser_state_dict loads into the genuine torch optimiser and TrainStep resumes
    from it."""
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    noise = [gold.noise(1 + it) for it in range(4)]

    def kw(it):
        rp, eps = noise[it]
        return dict(rand_pixel=rp.to(DEV), eps=torch.stack(eps).to(DEV))

    lr = 1e-3 if kind == 'rmsprop' else 1e-5
    ts = TrainStep(build(gold), gold.S, lr=lr, optimiser=kind, use_geco=False, beta_fixed=0.5)
    ref = build(gold)
    opt = torch.optim.RMSprop(ref.parameters(), lr) if kind == 'rmsprop' else torch.optim.SGD(ref.parameters(), lr, 0.9)
    for it in range(3):
        out = ts.step(xd, **kw(it))
        _, losses, _, _, _ = ref(xd, **kw(it))
        err, kl = losses.err.mean(0), torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
        opt.zero_grad()
        (err + 0.5 * kl).backward()
        opt.step()
        assert abs(float(out[0]) - float((err + kl).detach())) <= (2e-5 if it == 0 else 2e-3) * abs(float(out[0])) and float(out[3]) == 0.5
        if kind == 'sgd' and it == 0:
            # one step from identical parameters: p0 - lr g, linear in the gradient (later steps may pick other seed pixels in
            # the two loops -- the argmax is discontinuous -- and RMSprop's first steps are ~ 10 lr sign(g), where round-off
            # decides on analytically-zero gradients: the optimiser ARITHMETIC is pinned on identical gradients by
            # test_rmsprop_and_sgd_kernels_match_torch)
            init = [q.detach().clone() for q in build(gold).parameters()]
            for a, b, p0 in zip(ts.model.parameters(), ref.parameters(), init):
                upd = float((b.detach() - p0).norm())
                assert float((a.detach() - b.detach()).norm()) <= 5e-3 * upd + 4e-7 * float(p0.norm()) + 1e-7
    # wire format: the reference's resume path (train.py:179-207) with the matching torch optimiser, and ours
    ckpt = ts.state_dict(2)
    torch.save(ckpt, tmp_path / 'c')
    ckpt = torch.load(tmp_path / 'c', map_location='cuda', weights_only=False)
    opt2 = torch.optim.RMSprop(ref.parameters(), lr) if kind == 'rmsprop' else torch.optim.SGD(ref.parameters(), lr, 0.9)
    import copy
    opt2.load_state_dict(copy.deepcopy(ckpt['optimiser_state_dict']))
    key = 'square_avg' if kind == 'rmsprop' else 'momentum_buffer'
    for p_ref, p_ts in zip(ref.parameters(), ts.model.parameters()):
        assert opt2.state[p_ref][key].shape == p_ref.shape

    ts2 = TrainStep(build(gold), gold.S, lr=lr, optimiser=kind, use_geco=False, beta_fixed=0.5)
    assert ts2.load_state_dict(ckpt) == 3 and int(ts2.step_t) == 3
    a, b = ts.step(xd, **kw(3)), ts2.step(xd, **kw(3))
    assert torch.equal(a, b) and torch.equal(ts.flat_p, ts2.flat_p)
    with pytest.raises(ValueError):
        TrainStep(build(gold), gold.S, optimiser='adagrad')
    ts.close(); ts2.close()


@pytest.mark.parametrize('kind', ['rmsprop', 'sgd'])
def test_rmsprop_and_sgd_kernels_match_torch(kind):
    """gx_optimiser_step_pair against torch.optim.RMSprop(lr) / torch.optim.SGD(lr, 0.9) on identical gradients: an fp32 and an
    fp64 group in one launch, five steps, gradients zeroed as they are consumed."""
    import ctypes
    from genesis_amd import _lib
    torch.manual_seed(0)
    p32, p64 = torch.randn(10007), torch.randn(33, dtype=torch.float64)
    refs = [p32.clone().requires_grad_(), p64.clone().requires_grad_()]
    opt = torch.optim.RMSprop(refs, 1e-3) if kind == 'rmsprop' else torch.optim.SGD(refs, 1e-3, 0.9)
    d32, d64 = p32.to(DEV), p64.to(DEV)
    m32, m64 = torch.zeros_like(d32), torch.zeros_like(d64)
    step = torch.zeros((), dtype=torch.int64, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    for it in range(5):
        g32, g64 = torch.randn(10007) * (0.1 + it), torch.randn(33, dtype=torch.float64) * (0.1 + it)
        refs[0].grad, refs[1].grad = g32.clone(), g64.clone()
        opt.step()
        gd32, gd64 = g32.to(DEV), g64.to(DEV)
        kind_id, hp, eps = (1, 0.99, 1e-8) if kind == 'rmsprop' else (2, 0.9, 0.0)
        _lib.call('gx_optimiser_step_pair', kind_id, P(d32), P(gd32), P(m32), d32.numel(), P(d64), P(gd64), P(m64), d64.numel(),
                  P(step), 1e-3, hp, eps, 1.0, 1, st)
        assert float(gd32.abs().max()) == 0.0 and float(gd64.abs().max()) == 0.0
    np.testing.assert_allclose(d32.cpu().numpy(), refs[0].detach().numpy(), rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(d64.cpu().numpy(), refs[1].detach().numpy(), rtol=1e-12, atol=1e-14)


def test_beta_warmup_and_mse_logging():
    """train.py:249-259: without GECO and with config.beta_warmup, beta = clamp(beta * iter / (0.2 * train_iter), 0, beta) --
    0 at the first iteration; train.py:244-246: mse / rmse of the reconstruction, here from one launch next to the step."""
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    for graph in (False, True):
        model = build(gold)
        ts = TrainStep(model, gold.S, use_geco=False, beta_fixed=0.5, beta_warmup=True, train_iter=20, log_mse=True, graph=graph)
        torch.manual_seed(3)
        betas, outs = [], []
        for it in range(7):
            out = ts.step(xd).clone()
            betas.append(float(out[3])); outs.append(out)
        np.testing.assert_allclose(betas, [min(0.5 * it / 4.0, 0.5) for it in range(7)], rtol=1e-6)     # 0.2 * 20 = 4 iterations
        assert outs[0].shape == (6,)
        with torch.no_grad():
            recon = model(xd)[0]
        # mse / rmse are those of the step's own forward (its noise); a fresh forward reconstructs almost the same image
        mse = ((xd - recon) ** 2).mean((1, 2, 3))
        assert abs(float(outs[-1][4]) - float(mse.mean())) <= 0.05 * float(mse.mean())
        assert abs(float(outs[-1][5]) - float(mse.sqrt().mean())) <= 0.05 * float(mse.sqrt().mean())
        ts.close()
    with pytest.raises(ValueError):
        TrainStep(build(gold), gold.S, beta_warmup=True, train_iter=10)          # GECO on: no warm-up branch (train.py:249)


def test_mse_rmse_kernel_exact():
    import ctypes
    from genesis_amd import _lib
    g = torch.Generator().manual_seed(5)
    x, r = torch.rand(5, 3, 32, 32, generator=g), torch.rand(5, 3, 32, 32, generator=g)
    mse = ((x.double() - r.double()) ** 2).mean((1, 2, 3))
    xd, rd = x.to(DEV), r.to(DEV)
    out, ws = torch.zeros(2, device=DEV), torch.zeros(5 + 4, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):        # (twice: the kernel re-arms its completion counter)
        _lib.call('gx_mse_rmse', ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(rd.data_ptr()), 5, 3 * 32 * 32,
                  ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, st)
    np.testing.assert_allclose(out.cpu().numpy(), [float(mse.mean()), float(mse.sqrt().mean())], rtol=2e-6)


def _describe_ckpt(obj, path, out):
    """The same flattening as tests/golden/make_golden_checkpoint.py:describe (structure only)."""
    if torch.is_tensor(obj):
        out[path] = ('tensor', str(obj.dtype).replace('torch.', ''), list(obj.shape), obj)
    elif isinstance(obj, dict):
        out[path] = (type(obj).__name__, '', [len(obj)], None)
        for k, v in obj.items():
            _describe_ckpt(v, '%s/%s%s' % (path, 'i:' if isinstance(k, int) else '', k), out)
    elif isinstance(obj, (list, tuple)):
        out[path] = (type(obj).__name__, '', [len(obj)], None)
        for i, v in enumerate(obj):
            _describe_ckpt(v, '%s/#%d' % (path, i), out)
    else:
        out[path] = (type(obj).__name__, repr(obj), [], None)


def test_checkpoint_written_by_the_reference(tmp_path):
    """tests/golden/ckpt_v2_tiny.npz is the manifest (key paths, Python types, dtypes, shapes, tensor summaries) of a file the
    reference's OWN save_checkpoint (train.py:410-420) wrote after two of its training iterations on the `tiny` fixture, plus
    the third iteration the reference computed after restoring it (train.py:179-207).  TrainStep after the same two steps must
    hold that checkpoint: the same entries of the same types with the same values; a dict of exactly the manifest's structure
    loads through TrainStep.load_state_dict, and the next step is the reference's third iteration."""
    import json
    import os.path as osp
    import numpy as np
    from collections import OrderedDict
    from genesis_amd import testing as T
    from genesis_amd.trainer import TrainStep
    from tests.common import GOLDEN
    g = np.load(osp.join(GOLDEN, 'ckpt_v2_tiny.npz'), allow_pickle=False)
    manifest = json.loads(str(g['manifest_json']))
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    noise = [gold.noise(1 + it) for it in range(3)]

    def kw(it):
        rp, eps = noise[it]
        return dict(rand_pixel=rp.to(DEV), eps=torch.stack(eps).to(DEV))

    ts = TrainStep(build(gold), gold.S, lr=1e-4, graph=False)
    hist = [ts.step(xd, **kw(it)).cpu().double().numpy() for it in range(2)]
    for it in range(2):
        ref = g['hist'][it]
        assert abs(hist[it][0] - ref[0]) <= 2e-4 * abs(ref[0]) and abs(hist[it][1] - ref[1]) <= 2e-4 * abs(ref[1])
        assert abs(hist[it][3] - ref[3]) <= 1e-5 * abs(ref[3])        # beta used in the step
    f = tmp_path / 'model.ckpt-1'
    torch.save(ts.state_dict(1), f)
    mine = {}
    _describe_ckpt(torch.load(f, map_location='cpu', weights_only=False), '', mine)

    # --- structure: the same paths, container types, scalar types and values, tensor dtypes and shapes
    ref_paths = [m[0] for m in manifest]
    assert sorted(mine) == sorted(ref_paths), sorted(set(mine) ^ set(ref_paths))[:10]
    # (entry order inside the dicts torch.load hands to load_state_dict: OrderedDict of the model, parameter indices)
    assert [p for p in mine if p.startswith('/model_state_dict/')] == [p for p in ref_paths if p.startswith('/model_state_dict/')]
    # moments: relative L2 per tensor on the strided samples, with a floor relative to the LARGEST moment tensor (the gradients of
    # conv biases in front of a norm are analytically zero: their moments are round-off of round-off in every implementation)
    def mean_abs(path):
        return float(g['t/' + path + '/asum']) / max(int(g['t/' + path + '/n']), 1)
    big_m = max(mean_abs(m[0]) for m in manifest if m[1] == 'tensor' and m[0].endswith('/exp_avg'))
    big_v = max(mean_abs(m[0]) for m in manifest if m[1] == 'tensor' and m[0].endswith('/exp_avg_sq'))
    for path, kind, desc, shape in manifest:
        k2, d2, s2, t = mine[path]
        if kind == 'tensor':
            assert (k2, d2, s2) == (kind, desc, shape), (path, (k2, d2, s2), (kind, desc, shape))
            tt = t.double() if t.dtype == torch.float64 else t.float()
            if path.endswith('/exp_avg') or path.endswith('/exp_avg_sq'):
                sm = T.summarize(tt)
                ref = g['t/' + path + '/samples'].astype(np.float64)
                assert int(g['t/' + path + '/n']) == int(sm['n']), path
                second = path.endswith('_sq')
                floor = (1e-5 * big_v if second else 1e-4 * big_m) * np.sqrt(len(ref))
                diff = float(np.linalg.norm(sm['samples'].astype(np.float64) - ref))
                assert diff <= (1e-2 if second else 5e-3) * float(np.linalg.norm(ref)) + floor, (path, diff, float(np.linalg.norm(ref)), floor)
                continue
            scalarish = path.endswith('/step') or path in ('/beta', '/err_ema')
            # parameters after two Adam steps of 1e-4: sign-like updates, so 1e-6 absolute on O(0.1) weights is round-off of
            # round-off
            scale = mean_abs(path)
            rtol = 1e-5 if scalarish else 2e-3
            T.check_summary('t/' + path, tt, g, rtol, rtol * scale + 1e-12, path)
        elif kind in ('dict', 'OrderedDict', 'list', 'tuple'):
            assert (k2, s2) == (kind, shape), (path, k2, kind, s2, shape)
        else:
            assert k2 == kind, (path, k2, kind)
            if kind == 'float':
                assert abs(float(d2) - float(desc)) <= 1e-12 * abs(float(desc)), (path, d2, desc)
            else:
                assert d2 == desc, (path, d2, desc)

    # --- a dict of EXACTLY the manifest's structure (what torch.load returns for the reference's file), filled from this
    #     loop's state, restores through load_state_dict; the third step is the reference's third iteration
    def build_from_manifest():
        root = {}
        nodes = {'': root}
        for path, kind, desc, shape in manifest[1:]:
            parent, key = path.rsplit('/', 1)
            if kind == 'tensor':
                val = mine[path][3].clone()
            elif kind == 'OrderedDict':
                val = OrderedDict()
            elif kind == 'dict':
                val = {}
            elif kind == 'list':
                val = []
            elif kind == 'tuple':
                val = None          # filled below from its children
            else:
                val = {'int': int, 'float': float, 'bool': lambda s: s == 'True', 'NoneType': lambda s: None, 'str': lambda s: s[1:-1]}[kind](desc)
            nodes[path] = val
            cont = nodes[parent]
            if key.startswith('#'):
                cont.append(val)
            else:
                cont[int(key[2:]) if key.startswith('i:') else key] = val
        return root

    tuples = [m[0] for m in manifest if m[1] == 'tuple']
    if tuples:
        # tuples (Adam's betas) are rebuilt as lists first, then frozen
        for m in manifest:
            if m[1] == 'tuple':
                m[1] = 'list'
        ck = build_from_manifest()
        for path in tuples:
            parent, key = path.rsplit('/', 1)
            node = ck
            for part in parent.strip('/').split('/'):
                node = node[int(part[1:])] if part.startswith('#') else node[int(part[2:]) if part.startswith('i:') else part]
            node[key] = tuple(node[key])
    else:
        ck = build_from_manifest()
    assert set(ck) == {'model_state_dict', 'optimiser_state_dict', 'beta', 'err_ema', 'iter_idx'}
    ts2 = TrainStep(build(gold), gold.S, lr=1e-4, graph=False)
    assert ts2.load_state_dict(ck) == int(g['start_iter'])
    third = ts2.step(xd, **kw(2)).cpu().double().numpy()
    ref = g['hist'][2]
    assert abs(third[0] - ref[0]) <= 2e-4 * abs(ref[0]), (third, ref)
    assert abs(third[1] - ref[1]) <= 2e-4 * abs(ref[1]), (third, ref)
    assert abs(third[3] - ref[3]) <= 1e-5 * abs(ref[3]), (third, ref)          # GECO's beta restored from the file
    assert abs(float(ts2.geco.err_ema) - ref[4]) <= 1e-5 * abs(ref[4])
    after = torch.cat([p.detach().double().flatten().float().cpu() for p in ts2.model.parameters()])
    T.check_summary('after3/params', after, g, 1e-3, 2e-6, 'parameters after the third step')
