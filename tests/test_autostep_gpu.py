"""The reference's loop UNCHANGED (train.py:223-263) picks up TrainStep's launch structure by itself (genesis_amd/autostep.py):
one packed-weight refresh per iteration, one stream-K weight-gradient launch + batched reductions per backward pass, gradients
written straight into (zeroed) .grad views -- and leaves exactly the state the plain autograd path leaves."""
import pytest
import torch

from tests.common import Golden
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _loop(model, x, noise, steps, autostep_on, zero=True, lr=1e-4):
    from genesis_amd import autostep
    prev = autostep.ENABLED
    autostep.ENABLED = autostep_on
    try:
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        hist, grads = [], None
        for it in range(steps):
            if zero:
                opt.zero_grad()
            rp, eps = noise[it]
            recon, losses, stats, att, comp = model(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
            err = losses.err.mean(0)
            kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
            (err + kl).backward()
            if it == 0:
                grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
            opt.step()
            hist.append(float(err + kl))
        return hist, grads, {n: p.detach().clone() for n, p in model.named_parameters()}
    finally:
        autostep.ENABLED = prev


def _state_is_clean():
    from genesis_amd import _lib, autostep, functions as fn, hip_ops
    st = autostep._STATE
    assert not st.in_pass and not st.direct and st.cache_on is None and not st.models
    assert not fn.step_state().direct_param_grads and not hip_ops.defer_state().on
    assert _lib.query('gx_defer_pending') == 0


@pytest.mark.parametrize('case', ['tiny', 'metric'])
def test_unchanged_loop_on_the_step_machinery_equals_the_plain_path(case):
    gold = Golden(case)
    x, _, _ = gold.inputs()
    noise = [gold.noise(1 + it) for it in range(3)]
    h0, g0, p0 = _loop(build(gold), x, noise, 3, False)
    h1, g1, p1 = _loop(build(gold), x, noise, 3, True)
    _state_is_clean()
    # the first iteration's forward is the same launches; its gradients differ by the split-K slabs' summation order only
    assert abs(h0[0] - h1[0]) <= 1e-6 * abs(h0[0])
    worst = 0.0
    for n in g0:
        den = float(g0[n].double().norm()) + 1e-6 * max(float(v.double().norm()) for v in g0.values())
        worst = max(worst, float((g0[n].double() - g1[n].double()).norm()) / den)
    assert worst <= 2e-5, worst
    for a, b in zip(h0, h1):
        assert abs(a - b) <= 2e-4 * abs(a), (h0, h1)
    rel = max(float((p0[n].double() - p1[n].double()).norm() / (p0[n].double().norm() + 1e-12)) for n in p0)
    assert rel <= 1e-3, rel          # (Adam's first steps are sign-like: lr-sized differences from round-off-sized gradient ones)


def test_launch_structure_of_the_unchanged_loop():
    """Second iteration on: ONE batched weight packing instead of one per conv call, the weight gradients as ONE stream-K launch."""
    from genesis_amd import profiling
    gold = Golden('metric')
    x, _, _ = gold.inputs()
    model = build(gold)
    noise = [gold.noise(1 + it) for it in range(3)]
    _loop(model, x, noise, 1, True)                       # records the packed-weight cache
    profiling.enable(True)
    try:
        _loop(model, x, noise, 1, True)
        rows = {r['name']: r['launches'] for r in profiling.collect()}
    finally:
        profiling.enable(False)
    assert rows.get('wgq_stream_kernel', 0) == 1, rows
    assert rows.get('pack_weights_kernel', 0) <= 4, rows          # (the cache serves the conv layers; a few per-call packings remain)
    _state_is_clean()


def test_gradient_accumulation_and_interrupted_passes_fall_back_to_the_plain_path():
    from genesis_amd import autostep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    noise = [gold.noise(1)] * 2
    # two backward passes without zero_grad: the second finds gradients in place -> plain accumulation, twice the gradient
    model = build(gold)
    prev, autostep.ENABLED = autostep.ENABLED, True
    try:
        def fb():
            rp, eps = noise[0]
            recon, losses, *_ = model(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
            (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()).backward()
        fb()
        g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        fb()
        for n, p in model.named_parameters():
            assert torch.allclose(p.grad, 2 * g1[n], rtol=2e-4, atol=1e-6 * float(g1[n].abs().max() + 1e-12)), n
        _state_is_clean()
        # a forward that is never followed by a backward, then a normal iteration
        model.zero_grad(set_to_none=True)
        rp, eps = noise[0]
        model(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
        fb()
        for n, p in model.named_parameters():
            assert torch.allclose(p.grad, g1[n], rtol=2e-4, atol=1e-6 * float(g1[n].abs().max() + 1e-12)), n
        _state_is_clean()
        # evaluation forwards do not arm anything
        with torch.no_grad():
            model(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
        model.eval()
        model(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
        _state_is_clean()
    finally:
        autostep.ENABLED = prev


def test_trainstep_and_the_unchanged_loop_coexist():
    """A TrainStep re-homes the parameters into its flat bucket: the loop's packed-weight cache notices the moved pointers, and a
    TrainStep iteration is never touched by the mechanism (it runs in its own library context)."""
    from genesis_amd.trainer import TrainStep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    noise = [gold.noise(1 + it) for it in range(2)]
    model = build(gold)
    _loop(model, x, noise, 1, True)
    ts = TrainStep(model, gold.S, lr=1e-4)
    rp, eps = noise[1]
    out = ts.step(x.to(DEV), rand_pixel=rp.to(DEV), eps=torch.stack(eps).to(DEV))
    assert torch.isfinite(out).all()
    ts.close()
    h, _, _ = _loop(model, x, noise, 1, True)
    assert h[0] == h[0] and abs(h[0]) < 1e9
    _state_is_clean()


def _one_iteration_grads(model, x, noise):
    rp, eps = noise
    model.zero_grad(set_to_none=True)
    recon, losses, *_ = model(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
    (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()).backward()
    return losses


def test_distributed_data_parallel_wrapper_takes_the_plain_path():
    """DDP copies p.grad into its bucket from AccumulateGrad post-hooks, i.e. before the mechanism's end-of-backward flush would
    have written the conv-weight gradients: under a DDP forward the mechanism must stay off, and the reduced gradients must be
    the plain path's (world 1: the bucket round trip still happens)."""
    import os
    import socket
    import torch.distributed as dist
    from genesis_amd import autostep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    noise = gold.noise(1)
    prev, autostep.ENABLED = autostep.ENABLED, False
    try:
        ref = build(gold)
        _one_iteration_grads(ref, x, noise)
        g0 = {n: p.grad.detach().clone() for n, p in ref.named_parameters()}
        autostep.ENABLED = True
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        own_group = not dist.is_initialized()
        if own_group:
            dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0)
        try:
            model = build(gold)
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[torch.cuda.current_device()])
            rp, eps = noise
            recon, losses, *_ = ddp(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
            st = autostep._STATE
            assert st.cache_on is None and not st.models          # the DDP forward armed nothing
            (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()).backward()
            for n, p in model.named_parameters():
                assert p.grad is not None, n
                den = float(g0[n].double().norm()) + 1e-6 * max(float(v.double().norm()) for v in g0.values())
                assert float((p.grad.double() - g0[n].double()).norm()) / den <= 2e-5, n
            _state_is_clean()
        finally:
            if own_group:
                dist.destroy_process_group()
    finally:
        autostep.ENABLED = prev


def test_data_parallel_replicas_and_worker_threads_take_the_plain_path():
    """train.py --multi_gpu (train.py:153-155) wraps the model in nn.DataParallel: replicas run on worker threads and have no
    parameters(); they must not touch the mechanism's (process-wide) state or the library's packed-weight cache tables."""
    import threading
    from genesis_amd import autostep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    rp, eps = gold.noise(1)
    prev, autostep.ENABLED = autostep.ENABLED, True
    try:
        model = build(gold)
        dp = torch.nn.DataParallel(model, device_ids=[torch.cuda.current_device()])
        # one device: DataParallel calls the module itself on the calling thread -- exercise the replica path explicitly
        replica = torch.nn.parallel.replicate(model, [torch.cuda.current_device()])[0]
        assert getattr(replica, '_is_replica', False)
        seen = {}

        def worker():
            with torch.cuda.device(torch.cuda.current_device()):
                out = replica(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
                seen['err'] = out[1].err.detach().clone()
                seen['state'] = (autostep._STATE.cache_on, list(autostep._STATE.models))
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert seen['state'] == (None, []), seen['state']
        recon, losses, *_ = dp(x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
        assert torch.allclose(losses.err, seen['err'], rtol=1e-6, atol=1e-6)
        (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()).backward()
        _state_is_clean()
    finally:
        autostep.ENABLED = prev


def test_evaluation_forward_after_an_abandoned_training_forward_sees_the_new_weights():
    """A training forward that never gets its backward leaves the packed-weight cache serving; an in-place weight change followed
    by an EVALUATION forward (which does not arm) must not be answered from it."""
    from genesis_amd import autostep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    rp, eps = gold.noise(1)
    args = (x.to(DEV), rp.to(DEV), torch.stack(eps).to(DEV))
    prev, autostep.ENABLED = autostep.ENABLED, True
    try:
        model = build(gold)
        _one_iteration_grads(model, x, (rp, eps))         # records the cache
        model.zero_grad(set_to_none=True)
        model(*args)                                      # armed, cache serving, no backward follows
        with torch.no_grad():
            for p in model.parameters():
                if p.dim() == 4:
                    p.mul_(1.25)
        model.eval()
        with torch.no_grad():
            got = model(*args)[1].err.clone()
        _state_is_clean()
        autostep.ENABLED = False
        with torch.no_grad():
            want = model(*args)[1].err.clone()
        assert torch.equal(got, want), (got, want)
    finally:
        autostep.ENABLED = prev


def test_copies_of_a_model_do_not_share_the_native_cache():
    import copy
    from genesis_amd import autostep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    noise = gold.noise(1)
    prev, autostep.ENABLED = autostep.ENABLED, True
    try:
        model = build(gold)
        _one_iteration_grads(model, x, noise)
        _one_iteration_grads(model, x, noise)
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        twin = copy.deepcopy(model)
        assert not any(k.startswith('_gx_autostep') for k in twin.__dict__)
        assert autostep._BOOK.get(twin) is None
        _one_iteration_grads(twin, x, noise)
        _one_iteration_grads(twin, x, noise)
        cid = autostep._BOOK[model]['cache'][0]
        assert autostep._BOOK[twin]['cache'][0] != cid
        del twin
        import gc
        gc.collect()
        _one_iteration_grads(model, x, noise)             # the original's cache is still alive
        for n, p in model.named_parameters():
            assert torch.equal(p.grad, g[n]), n
        # a replaced parameter is noticed (the cached parameter list is validated by identity)
        w = model.seg_head.params()[0]
        owner = [(m, k) for m in model.modules() for k, v in m._parameters.items() if v is w][0]
        setattr(owner[0], owner[1], torch.nn.Parameter(w.detach().clone()))
        _one_iteration_grads(model, x, noise)
        assert getattr(owner[0], owner[1]).grad is not None
        _state_is_clean()
    finally:
        autostep.ENABLED = prev


# ------------------------------------------------------------------------------------------------ the loop as two replayed graphs
def _plain_loop(model, x, steps, seeds, opt=None, lr=1e-4, zero=True, hook=None):
    """train.py:223-263's statements with the model's OWN noise (torch.rand / randn inside forward), torch seeded per iteration."""
    opt = opt or torch.optim.Adam(model.parameters(), lr=lr)
    hist = []
    for it in range(steps):
        if zero:
            opt.zero_grad()
        torch.manual_seed(seeds[it])
        recon, losses, stats, att, comp = model(x)
        err = losses.err.mean(0)
        kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
        loss = err + 0.7 * kl
        if hook is not None:
            loss = hook(loss, recon, stats)
        loss.backward()
        opt.step()
        hist.append((float(err.detach()), float(kl.detach())))
    return hist, opt


@pytest.mark.parametrize('case', ['tiny', 'metric'])
def test_unchanged_loop_as_two_replayed_graphs_equals_the_eager_loop(case):
    """From the third iteration on the loop's forward and backward passes are two replayed HIP graphs (autostep.graph_forward):
    the same trajectory as the eager path (same torch seeds: the captured torch.rand / randn draw what the eager ones draw),
    every parameter with its .grad after backward(), the replays counted, the state clean."""
    from genesis_amd import autostep
    gold = Golden(case)
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    seeds = [100 + i for i in range(7)]
    prev = (autostep.ENABLED, autostep.GRAPH)
    try:
        autostep.ENABLED, autostep.GRAPH = True, False
        h0, _ = _plain_loop(build(gold), xd, 7, seeds)
        autostep.GRAPH = True
        model = build(gold)
        h1, _ = _plain_loop(model, xd, 7, seeds)
        fwd, bwd, fb = autostep.graph_stats(model)
        assert fwd == 5 and bwd == 5 and fb == 0, (fwd, bwd, fb)
        assert all(p.grad is not None for p in model.parameters())
        for a, b in zip(h0, h1):
            assert abs(a[0] - b[0]) <= 2e-4 * abs(a[0]) and abs(a[1] - b[1]) <= 2e-3 * abs(a[1]) + 1e-4, (h0, h1)
        _state_is_clean()
    finally:
        autostep.ENABLED, autostep.GRAPH = prev


def test_graph_loop_falls_back_and_stays_correct():
    """Gradient accumulation (no zero_grad between two backward passes), a loss that also reads `recon`, an evaluation forward
    and a changed batch size in the middle of a captured loop: each takes the ordinary autograd path and gives the gradients
    the eager loop gives."""
    from genesis_amd import autostep
    gold = Golden('tiny')
    x, _, _ = gold.inputs()
    xd = x.to(DEV)
    prev = (autostep.ENABLED, autostep.GRAPH)

    def grads_after(model, graph_on, scenario):
        autostep.GRAPH = graph_on
        opt = torch.optim.SGD(model.parameters(), lr=0.0)          # (lr 0: the iterations differ by their noise only)
        _plain_loop(model, xd, 4, [1, 2, 3, 4], opt=opt)           # by now the graph stage has captured (if on)
        opt.zero_grad()
        if scenario == 'accumulate':
            for s in (5, 6):
                torch.manual_seed(s)
                _, losses, *_ = model(xd)
                (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()).backward()
        elif scenario == 'recon_loss':
            torch.manual_seed(5)
            recon, losses, *_ = model(xd)
            (losses.err.mean(0) + recon.pow(2).mean()).backward()
        elif scenario == 'eval_between':
            model.eval()
            with torch.no_grad():
                model(xd)
            model.train()
            torch.manual_seed(5)
            _, losses, *_ = model(xd)
            (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()).backward()
        else:
            torch.manual_seed(5)
            _, losses, *_ = model(xd[:1])
            (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()).backward()
        # (a parameter the backward pass never reaches: None on the plain autograd path, zeros under the mechanism)
        return {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}

    try:
        autostep.ENABLED = True
        for scenario in ('accumulate', 'recon_loss', 'eval_between', 'other_batch'):
            g0 = grads_after(build(gold), False, scenario)
            model = build(gold)
            g1 = grads_after(model, True, scenario)
            big = max(float(v.double().norm()) for v in g0.values())
            for n in g0:
                den = float(g0[n].double().norm()) + 1e-6 * big
                assert float((g0[n].double() - g1[n].double()).norm()) / den <= 5e-5, (scenario, n)
            fwd, bwd, fb = autostep.graph_stats(model)
            assert fwd >= 2, (scenario, fwd)
            if scenario in ('accumulate', 'recon_loss'):
                assert fb >= 1, (scenario, fb)
            _state_is_clean()
    finally:
        autostep.ENABLED, autostep.GRAPH = prev
