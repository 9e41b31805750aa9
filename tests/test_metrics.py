"""Segmentation metrics: the oracle restatement against golden values produced by the reference's own
utils/misc.py functions (CPU), and the device implementation (HIP contingency kernel) against both (GPU)."""
import glob
import os.path as osp

import numpy as np
import pytest
import torch

HERE = osp.dirname(osp.abspath(__file__))
CASES = sorted(osp.basename(p)[8:-4] for p in glob.glob(osp.join(HERE, 'golden', 'metrics_*.npz')))


def load(case):
    g = np.load(osp.join(HERE, 'golden', 'metrics_%s.npz' % case))
    log_m_k = [torch.from_numpy(g['log_m'][k]) for k in range(g['log_m'].shape[0])]
    return g, log_m_k, torch.from_numpy(g['inst'])


@pytest.mark.parametrize('case', CASES)
def test_oracle_matches_reference_golden(case):
    from oracle import metrics_oracle as O
    g, log_m_k, inst = load(case)
    for fg in (0, 1):
        if 'ari_mean_fg%d' % fg in g:
            mean, lst = O.average_ari([m.numpy() for m in log_m_k], inst.numpy(), bool(fg))
            np.testing.assert_allclose(lst, g['ari_list_fg%d' % fg], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(mean, g['ari_mean_fg%d' % fg], rtol=1e-12)
    ins_seg = torch.argmax(torch.cat(log_m_k, 1), 1, True)
    for bg in (0, 1):
        m, s = O.average_segcover(inst.numpy(), ins_seg.numpy(), bool(bg))
        assert m == g['sc_mean_bg%d' % bg] and s == g['sc_scaled_bg%d' % bg]      # float32, bit for bit


def test_contingency_oracle_is_exact_on_a_hand_case():
    from oracle import metrics_oracle as O
    # two clusterings of 6 points: ARI known from the textbook pair-counting definition
    assert abs(O.adjusted_rand_score([0, 0, 1, 1, 2, 2], [0, 0, 1, 2, 2, 2]) - 0.4444444444444444) < 1e-15
    assert O.adjusted_rand_score([0, 0, 1, 1], [1, 1, 0, 0]) == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_device_metrics_match_reference_golden(case):
    from genesis_amd import metrics as M
    g, log_m_k, inst = load(case)
    dm = [m.cuda() for m in log_m_k]
    for fg in (0, 1):
        if 'ari_mean_fg%d' % fg in g:
            mean, lst = M.average_ari(dm, inst.cuda(), bool(fg))
            np.testing.assert_allclose(lst, g['ari_list_fg%d' % fg], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(mean, g['ari_mean_fg%d' % fg], rtol=1e-12)
    ins_seg = torch.argmax(torch.cat(dm, 1), 1, True)
    for bg in (0, 1):
        m, s = M.average_segcover(inst.cuda(), ins_seg, bool(bg))
        # integer tables are exact; the last float32 steps (per-image quotient, batch mean) may round one ulp apart
        # between the host's and the device's reduction order: 2e-7 relative
        np.testing.assert_allclose(float(m), float(g['sc_mean_bg%d' % bg]), rtol=2e-7)
        np.testing.assert_allclose(float(s), float(g['sc_scaled_bg%d' % bg]), rtol=2e-7)


@pytest.mark.gpu
def test_contingency_kernel_bit_exact_and_edge_cases():
    from genesis_amd import metrics as M
    g = torch.Generator().manual_seed(3)
    B, HW, KA, KB = 5, 64 * 64, 6, 9
    a = torch.randint(-2, KA + 1, (B, HW), generator=g)          # includes ignore labels (< 0) and out-of-range ones
    b = torch.randint(-1, KB + 2, (B, HW), generator=g)
    c = M.contingency(a.cuda(), b.cuda(), KA, KB).cpu()
    ref = torch.zeros(B, KA, KB + 1, dtype=torch.int32)
    for i in range(KA):
        for j in range(KB):
            ref[:, i, j] = ((a == i) & (b == j)).sum(1)
        ref[:, i, KB] = ((a == i) & ((b < 0) | (b >= KB))).sum(1)
    assert torch.equal(c, ref)
    # a single-pixel image and a single label
    one = torch.zeros(1, 1, dtype=torch.int64).cuda()
    assert M.contingency(one, one, 1, 1).cpu().tolist() == [[[1, 0]]]
    with pytest.raises(Exception):
        M.contingency(a, b, KA, KB)                              # host tensors: no CPU path
