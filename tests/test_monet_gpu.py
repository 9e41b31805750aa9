"""MONet (BASELINE config 4) on the HIP path vs golden vectors captured from the real reference: forward tensors,
parameter gradients, three GECO + Adam steps, output contract."""
import numpy as np
import pytest
import torch

from tests.test_monet_oracle import CASES, MonetGolden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build(gold):
    import genesis_amd.monet_config as G
    from genesis_amd.compat.attrdict import AttrDict
    cfg = AttrDict(dict(gold.cfg, debug=False, multi_gpu=False))
    torch.manual_seed(0)
    model = G.load(cfg)
    model.load_state_dict(gold.weights(model.state_dict()))
    return model.to(DEV).train()


@pytest.mark.parametrize('case', CASES)
def test_forward_and_grads_vs_golden(case):
    gold = MonetGolden(case)
    model = build(gold)
    x, eps = gold.inputs()
    recon, losses, stats, _, comp = model(x.to(DEV), eps.to(DEV))
    gold.check_forward(recon, losses, stats, comp, rtol=1e-4, atol=2e-5, mask_atol=1e-3)
    err = losses.err.mean(0)
    kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum() + losses.kl_m.mean(0)
    elbo_ref = float(gold.g['loss/err']) + float(gold.g['loss/kl'])
    assert abs(float(err + kl) - elbo_ref) <= 1e-3 * abs(elbo_ref)
    assert abs(float(err + kl) - elbo_ref) <= 5e-5 * abs(elbo_ref)
    (err + kl).backward()
    # per-parameter tolerance from the fp32 error budget of THIS case (tests.common.fp32_budget: the CPU oracle in fp32 against
    # fp64 on the golden weights and inputs): 5 x that error (HIP's 4 x bar + the reference's own) + 1e-3.  The recurrent
    # UNet(IN) attention (K-1 shared-weight passes) is ill-conditioned in fp32 -- 0.3-3 % for ANY fp32 implementation at
    # 64x64 / K=7 --, the ComponentVAE gradients agree to 1e-7..1e-5: the budget gives each its own bar
    from tests.common import budget_tolerances, fp32_budget
    from oracle import monet_oracle as MO
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    def loss_fn(p, dtype):
        out = MO.monet_forward(p, x.to(dtype), gold.cfg, eps.to(dtype))
        e, kl_l, kl_m = MO.aggregate_losses(out[1])
        return e + kl_l + kl_m
    e_cpu = fp32_budget(loss_fn, sd, is_param=lambda k: k != 'std')
    tol = budget_tolerances(e_cpu, floor=1e-3, cap=6e-2)      # (cap: the round-2 constant; floor: ReLU flips, UNet + decoder)
    print('MONet %s: fp32 budget per parameter: max %.3e, median %.3e' % (case, max(e_cpu.values()), sorted(e_cpu.values())[len(e_cpu) // 2]))
    gold.check_grads([(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()],
                     per_param=tol)
    for key in ('log_m_k', 'log_m_r_k'):
        assert float((torch.stack(stats[key], 4).exp().sum(4) - 1).abs().max()) < 1e-3


@pytest.mark.parametrize('case', ['tiny', 'cfg4'])
def test_three_training_steps(case):
    from genesis_amd.trainer import TrainStep
    gold = MonetGolden(case)
    model = build(gold)
    ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
    x, _ = gold.inputs()
    hist = gold.g['train_hist']
    for it in range(3):
        _, eps = gold.inputs(1 + it)
        out = ts.step(x.to(DEV), eps=eps.to(DEV)).cpu().numpy()
        elbo, err, kl, beta = [float(v) for v in out]
        # step 0 is a pure forward comparison; later steps inherit the fp32 conditioning of the attention UNet's
        # gradients (see above) through Adam's sign-like first updates
        tol = 1e-4 if it == 0 else 1e-3
        assert abs(elbo - hist[it, 0]) <= tol * abs(hist[it, 0]), (it, elbo, hist[it])
        np.testing.assert_allclose([err, beta], hist[it, [1, 3]], rtol=5e-4)
    assert abs(float(ts.geco.beta) - float(gold.g['train_beta_final'])) <= 1e-5


def test_output_contract_and_sample():
    gold = MonetGolden('tiny')
    model = build(gold)
    B, K, S, L = gold.B, gold.K, gold.S, gold.L
    x, _ = gold.inputs()
    recon, losses, stats, att_stats, comp_stats = model(x.to(DEV))
    assert recon.shape == (B, 3, S, S) and losses.err.shape == (B,) and losses.kl_m.shape == (B,)
    assert len(losses.kl_l_k) == K and torch.stack(losses.kl_l_k, dim=1).shape == (B, K)
    assert torch.cat(stats.log_m_k, 1).shape == (B, K, S, S) and len(stats.log_s_k) == K
    assert len(comp_stats.z_k) == K and comp_stats.z_k[0].shape == (B, L)
    assert model.get_features(x.to(DEV)).shape == (B, K * L)
    img, st = model.sample(3)
    assert img.shape == (3, 3, S, S) and len(st.x_k) == K
    assert float((torch.stack(st.log_m_k, 4).exp().sum(4) - 1).abs().max()) < 1e-3
    assert list(model.state_dict().keys())[0] == 'std'


def test_shared_weights_accumulate_through_the_deferred_reductions():
    """MONet's attention UNet is ONE set of weights used K-1 times per iteration.  Inside TrainStep every use's conv weight
    gradient and GroupNorm affine gradient goes through the deferred reductions into the flat bucket (the queued reduces ADD;
    records that share a destination are split over consecutive launches): the bucket must equal plain autograd's
    accumulated gradient, bit-identically from run to run."""
    from genesis_amd.trainer import TrainStep
    gold = MonetGolden('tiny_k4')
    x, eps = gold.inputs()
    xd, ed = x.to(DEV), eps.to(DEV)
    ref_model = build(gold)
    recon, losses, _, _, _ = ref_model(xd, ed)
    (losses.err.mean(0) + torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum() + losses.kl_m.mean(0)).backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in ref_model.parameters()])
    runs = []
    for _ in range(4):
        model = build(gold)
        ts = TrainStep(model, gold.S, lr=1e-4, graph=False)
        ts._zero_grads()
        ts._enter()
        try:
            ts._forward_backward(xd, eps=ed)          # beta = 1 at the first iteration: the same objective
        finally:
            ts._leave()
        torch.cuda.synchronize()
        runs.append(torch.cat([p.grad.flatten() for p in model.parameters()]).clone())
        ts.close()
    for r in runs[1:]:
        assert torch.equal(runs[0], r)
    rel = float((runs[0] - ref).norm() / ref.norm())
    assert rel < 2e-5, rel
    # per parameter: nothing lost, nothing counted twice
    off = 0
    for n, p in ref_model.named_parameters():
        a, b = runs[0][off:off + p.numel()], ref[off:off + p.numel()]
        off += p.numel()
        if float(b.norm()) > 1e-6 * float(ref.norm()):
            assert float((a - b).norm()) <= 2e-4 * float(b.norm()), n
