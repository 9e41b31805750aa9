"""fp32 error budget against the fp64 oracle -- the rigorous form of the gradient comparison.

Ground truth = the oracle run in fp64 on the same weights, inputs and noise.  Two fp32 implementations are measured
against it: the oracle in fp32 on the host (stock ATen ops: what the reference computes) and the HIP path.  Bar:

    HIP error  <=  4 x (CPU-fp32 error)  +  floor,     per forward tensor and per parameter gradient (relative L2)

The floor: 2e-6 relative, and for analytically-zero gradients 2e-6 x the largest gradient norm.  (The golden-vector
tests compare against the reference's OWN fp32 results on closed-form weights, where a ReLU pre-activation within ~1e-7
of zero can flip sign between two correct fp32 implementations and move every upstream gradient by 1e-4 .. 1e-3
(tools/diag_model_decoder.py); here, against fp64 on random-init weights, no such allowance is needed: RELU_FLIP = 0.)
Shapes: the configuration of the round-1 test, the metric configuration (K=7, 64x64, feat 64) and BASELINE config 5
(K=11, 128x128) for GENESIS-V2; MONet (config 4) and GENESIS (config 3) at tiny and BASELINE shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
RELU_FLIP = 0.0      # measured: no allowance needed (worst HIP gradient error 5.6e-4 where CPU-fp32 has 7.6e-4)
# The golden fixtures' closed-form weights are another matter: pre-activations sit within 1e-7 of zero there, and WHICH
# fp32 implementation flips a ReLU is a coin toss per summation order.  Measured on golden case cfg2: CPU-fp32 itself is
# 2.7e-3 off fp64 on decoder_module.1.weight where the HIP path is 6e-6; on cfg5 the HIP path is 2.4e-5 off on
# decoder_module.10.weight where CPU-fp32 is 1.1e-6 (after the decoder's first layer became a matrix product over
# tap-summed weights: same mathematics, other rounding).  A per-parameter bar of 4 x the CPU error cannot hold in both
# directions, so that test alone grants a flip 1e-4 per parameter (a flip moves a gradient by 1e-5 .. 1e-3).
GOLDEN_RELU_FLIP = 1e-4


def relerr(a, ref):
    ref = ref.detach().double().cpu()
    return float((a.detach().double().cpu() - ref).norm()) / (float(ref.norm()) + 1e-30)


def judge(rows_fwd, grads_hip, g32, g64, what, relu_flip=None):
    """rows_fwd: (name, hip tensor, cpu32 tensor, fp64 tensor).  Prints the table, returns the offenders."""
    bad, table = [], []
    relu_flip = RELU_FLIP if relu_flip is None else relu_flip
    for name, got, r32, r64 in rows_fwd:
        e_gpu, e_cpu = relerr(got, r64), relerr(r32, r64)
        table.append((name, e_gpu, e_cpu))
        if e_gpu > 4 * e_cpu + 2e-6:
            bad.append(name)
    gmax = max(float(v.norm()) for v in g64.values())
    worst = 0.0
    for n, got in grads_hip:
        ref = g64[n]
        floor = 2e-6 * gmax / (float(ref.norm()) + 1e-30)
        e_gpu, e_cpu = relerr(got, ref), relerr(g32[n], ref)
        table.append(('grad ' + n, e_gpu, e_cpu))
        worst = max(worst, e_gpu - floor)
        if e_gpu > max(4 * e_cpu + 2e-6, relu_flip) + floor:
            bad.append(n)
    print('\n[%s] %-46s %12s %12s' % (what, 'tensor', 'hip-vs-f64', 'cpu32-vs-f64'))
    for r in table:
        print('%-56s %12.3e %12.3e' % r)
    print('[%s] worst gradient error above its floor: %.3e' % (what, worst))
    return bad


def to_dtype(sd, dtype):
    return {k: v.clone().to(dtype if v.dtype == torch.float32 else v.dtype).requires_grad_(v.is_floating_point())
            for k, v in sd.items()}


def grads_of(p):
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)).double() for k, v in p.items() if v.requires_grad}


def hip_grads(model):
    return [(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()]


@pytest.mark.parametrize('name,K,S,D,B', [('k5', 5, 64, 32, 4), ('metric', 7, 64, 64, 2), ('cfg5', 11, 128, 64, 1),
                                          ('metric_b32', 7, 64, 64, 32)])
def test_genesis_v2(name, K, S, D, B):
    """('metric_b32': the metric configuration at its FULL per-GPU batch -- the B = 32 dispatch: Winograd convs, row-ring
    weight gradients on the bf16 pipe, bf16-pipe transposed convs -- against the fp64 oracle.)"""
    from oracle import v2_oracle as O
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    from genesis_amd import testing as T
    cfg = O.make_cfg(K_steps=K, img_size=S, feat_dim=D)
    torch.manual_seed(7)
    model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False)))
    with torch.no_grad():
        model.att_process.colour_head.gate.gate.fill_(0.2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = T.make_input(99, B, S)
    rp, eps_k = T.draw_noise(123, B, S, D, K)

    def oracle(dtype, seed_idx=None):
        p = to_dtype(sd, dtype)
        out = O.v2_forward(p, x.to(dtype), cfg, rp.to(dtype), [e.to(dtype) for e in eps_k], seed_idx=seed_idx,
                           reference_form=False)
        e, kl, _ = O.aggregate_losses(out[1])
        (e + kl).backward()
        return out, grads_of(p)
    o64, g64 = oracle(torch.float64)
    seeds64 = torch.stack(o64[3]['seed_idx'])
    model = model.to(DEV)
    if B <= 4:
        o32, g32 = oracle(torch.float32)
        recon, losses, stats, att, _ = model(x.to(DEV), rp.to(DEV), torch.stack(eps_k).to(DEV))
        assert torch.equal(torch.stack(list(att['seed_idx'])).cpu(), seeds64)   # no fallback here
    else:
        # 32 x (K - 1) argmax decisions: a near-tie may legitimately fall the other way in fp32; the budget is about the
        # arithmetic, so BOTH fp32 runs (host and HIP) take the fp64 run's seed pixels -- after counting the free HIP run's
        o32, g32 = oracle(torch.float32, list(seeds64.unbind(0)))
        with torch.no_grad():
            free = torch.stack(list(model(x.to(DEV), rp.to(DEV), torch.stack(eps_k).to(DEV))[3]['seed_idx'])).cpu()
        agree = float((free == seeds64).float().mean())
        print('seed pixels of the free HIP run equal to the fp64 run: %.4f' % agree)
        assert agree >= 0.95
        recon, losses, stats, att, _ = model(x.to(DEV), rp.to(DEV), torch.stack(eps_k).to(DEV), seeds64.to(DEV))
    (losses.err.mean(0) + torch.stack(losses.kl_l_k, 1).mean(0).sum()).backward()
    st = lambda l: torch.stack(list(l))   # noqa: E731
    fwd = [('recon', recon, o32[0], o64[0]), ('err', losses.err, o32[1]['err'], o64[1]['err']),
           ('log_m', st(stats.log_m_k), st(o32[2]['log_m_k']), st(o64[2]['log_m_k'])),
           ('kl', st(losses.kl_l_k), st(o32[1]['kl_l_k']), st(o64[1]['kl_l_k']))]
    bad = judge(fwd, hip_grads(model), g32, g64, 'GENESIS-V2 ' + name)
    assert not bad, bad


@pytest.mark.parametrize('case', ['metric', 'cfg2', 'cfg5'])
def test_genesis_v2_on_the_golden_cases_weights_and_inputs(case):
    """The golden fixtures hold the reference's fp32 gradients on closed-form weights; against those the HIP gradients
    differ by 4e-3 .. 8e-3 at 64x64 / 128x128.  Same weights, inputs and noise against fp64: which side carries that
    error?  (Printed: both columns; bar as everywhere, HIP <= 4 x CPU-fp32 + floor.)"""
    from oracle import v2_oracle as O
    from tests.common import Golden
    from tests.test_model_gpu import build
    gold = Golden(case)
    model = build(gold)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, rp, eps_k = gold.inputs()
    cfg = gold.cfg

    def oracle(dtype):
        p = to_dtype(sd, dtype)
        out = O.v2_forward(p, x.to(dtype), cfg, rp.to(dtype), [e.to(dtype) for e in eps_k], reference_form=False)
        e, kl, _ = O.aggregate_losses(out[1])
        (e + kl).backward()
        return out, grads_of(p)
    o64, g64 = oracle(torch.float64)
    o32, g32 = oracle(torch.float32)
    seeds64 = torch.stack(o64[3]['seed_idx'])
    recon, losses, stats, att, _ = model(x.to(DEV), rp.to(DEV), torch.stack(eps_k).to(DEV), seeds64.to(DEV))
    (losses.err.mean(0) + torch.stack(losses.kl_l_k, 1).mean(0).sum()).backward()
    st = lambda l: torch.stack(list(l))   # noqa: E731
    fwd = [('recon', recon, o32[0], o64[0]), ('err', losses.err, o32[1]['err'], o64[1]['err']),
           ('log_m', st(stats.log_m_k), st(o32[2]['log_m_k']), st(o64[2]['log_m_k']))]
    bad = judge(fwd, hip_grads(model), g32, g64, 'GENESIS-V2 golden case ' + case, relu_flip=GOLDEN_RELU_FLIP)
    assert not bad, bad


@pytest.mark.parametrize('name,K,S,B', [('tiny', 4, 32, 2), ('cfg4', 7, 64, 2)])
def test_monet(name, K, S, B):
    from oracle import monet_oracle as O
    import genesis_amd.monet_config as G
    from genesis_amd.compat.attrdict import AttrDict
    from genesis_amd import testing as T
    cfg = O.make_cfg(K_steps=K, img_size=S)
    torch.manual_seed(11)
    model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False)))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = T.make_input(98, B, S)
    eps = torch.randn(K * B, cfg['comp_ldim'], generator=torch.Generator().manual_seed(5))

    def oracle(dtype):
        p = to_dtype(sd, dtype)
        out = O.monet_forward(p, x.to(dtype), cfg, eps.to(dtype))
        e, kl_l, kl_m = O.aggregate_losses(out[1])
        (e + kl_l + kl_m).backward()
        return out, grads_of(p)
    o64, g64 = oracle(torch.float64)
    o32, g32 = oracle(torch.float32)
    model = model.to(DEV)
    recon, losses, stats, _, _ = model(x.to(DEV), eps.to(DEV))
    (losses.err.mean(0) + torch.stack(losses.kl_l_k, 1).mean(0).sum() + losses.kl_m.mean(0)).backward()
    st = lambda l: torch.stack(list(l))   # noqa: E731
    fwd = [('recon', recon, o32[0], o64[0]), ('err', losses.err, o32[1]['err'], o64[1]['err']),
           ('kl_m', losses.kl_m, o32[1]['kl_m'], o64[1]['kl_m']),
           ('log_m', st(stats.log_m_k), st(o32[2]['log_m_k']), st(o64[2]['log_m_k']))]
    bad = judge(fwd, hip_grads(model), g32, g64, 'MONet ' + name)
    assert not bad, bad


@pytest.mark.parametrize('name,K,S,B', [('tiny', 3, 32, 2), ('cfg3', 7, 64, 2)])
def test_genesis(name, K, S, B):
    from oracle import genesis_oracle as O
    import genesis_amd.genesis_config as G
    from genesis_amd.compat.attrdict import AttrDict
    from genesis_amd import testing as T
    cfg = O.make_cfg(K_steps=K, img_size=S)
    torch.manual_seed(13)
    model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False)))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = T.make_input(97, B, S)
    gen = torch.Generator().manual_seed(6)
    eps_m = [torch.randn(B, cfg['attention_latents'], generator=gen) for _ in range(K)]
    eps_c = torch.randn(K * B, cfg['comp_ldim'], generator=gen)

    def oracle(dtype):
        p = to_dtype(sd, dtype)
        out = O.genesis_forward(p, x.to(dtype), cfg, [e.to(dtype) for e in eps_m], eps_c.to(dtype))
        e, kl_l, kl_m = O.aggregate_losses(out[1])
        (e + kl_l + kl_m).backward()
        return out, grads_of(p)
    o64, g64 = oracle(torch.float64)
    o32, g32 = oracle(torch.float32)
    model = model.to(DEV).train()
    recon, losses, stats, _, _ = model(x.to(DEV), [e.to(DEV) for e in eps_m], eps_c.to(DEV))
    (losses.err.mean(0) + torch.stack(losses.kl_m_k, 1).mean(0).sum() + torch.stack(losses.kl_l_k, 1).mean(0).sum()).backward()
    st = lambda l: torch.stack(list(l))   # noqa: E731
    fwd = [('recon', recon, o32[0], o64[0]), ('err', losses.err, o32[1]['err'], o64[1]['err']),
           ('log_m', st(stats.log_m_k), st(o32[2]['log_m_k']), st(o64[2]['log_m_k']))]
    bad = judge(fwd, hip_grads(model), g32, g64, 'GENESIS ' + name)
    assert not bad, bad


@pytest.mark.parametrize('B', [32, 64])
def test_unet_gradients_equal_fp64_on_the_same_relu_pattern(B):
    """The chip-filling dispatch of the UNet encoder (Winograd convs, stream-K weight gradients, single-slab data
    gradients) at the benchmark's batch sizes, without any allowance for ReLU decisions: a pre-activation within fp32
    round-off of zero may legitimately fall on either side, and ONE such decision moves every upstream gradient by ~1e-3
    (tests/test_fullbatch_gpu.py) -- so the fp64 oracle is evaluated ON THE HIP FORWARD'S OWN ReLU PATTERN (read from the
    autograd node's saved block outputs), which leaves pure arithmetic: every parameter gradient within 5e-6 of fp64 in
    relative L2; and the decisions that differ from fp64's own are counted and must sit at |pre-activation| < 1e-5."""
    import torch.nn.functional as F
    from oracle import v2_oracle as VO
    from genesis_amd import functions as fn
    import genesis_amd.genesisv2_config as G
    from genesis_amd.compat.attrdict import AttrDict
    cfg = VO.make_cfg(K_steps=5, img_size=64, feat_dim=64)
    torch.manual_seed(0)
    model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False, dynamic_K=False)))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith('encoder.')}
    model = model.to(DEV)
    nb = model.encoder.num_blocks
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, 64, 64, generator=g)
    dy = torch.randn(B, 64, 64, 64, generator=g)
    enc = fn.UNetEncoderFn.apply(x.to(DEV), nb, 8, *model.encoder.flat_params())
    node = enc.grad_fn
    masks = []                      # in the oracle's F.relu call order: down blocks, the three MLP layers, up blocks, F.relu(out)
    for i in range(nb):
        j = nb - 1 - i
        C = node.saved_down[i][1].shape[1]
        masks.append((node.cats[j][:, node.cats[j].shape[1] - C:] > 0).cpu())
    for q in range(3):
        masks.append((node.mlp[1][q][1] > 0).cpu())
    for j in range(nb):
        C = node.saved_up[j][0].shape[1]
        masks.append(((node.cats[j + 1][:, :C, ::2, ::2] if j < nb - 1 else enc) > 0).cpu())
    masks.append(torch.ones_like(masks[-1]))
    enc.backward(dy.to(DEV))
    real_relu, calls = F.relu, []

    def relu(t, inplace=False):
        m = masks[len(calls)].view(t.shape)
        calls.append(t.detach())
        return t * m.to(t.dtype)
    F.relu = relu
    try:
        p = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
        F.relu(VO.unet_forward(p, x.double(), nb)).backward(dy.double())
    finally:
        F.relu = real_relu
    differing = 0
    for pre, m in zip(calls[:-1], masks[:-1]):
        d = (pre > 0) != m.view(pre.shape)
        differing += int(d.sum())
        assert not d.any() or float(pre[d].abs().max()) < 1e-5     # only decisions at round-off distance from zero
    worst = max((relerr(prm.grad, p[k].grad), k) for k, prm in model.named_parameters() if k.startswith('encoder.'))
    print('UNet B=%d: %d ReLU decisions differ from fp64 (of %d); worst gradient error on the same pattern %.2e (%s)'
          % (B, differing, sum(m.numel() for m in masks[:-1]), worst[0], worst[1]))
    assert worst[0] <= 5e-6, worst
