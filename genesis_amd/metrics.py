"""Segmentation metrics of the reference's validation loop on the device: `average_ari` and `average_segcover` with the
signatures and return values of utils/misc.py:101-114 and :173-235 (callers: train.py:542-546,
scripts/compute_seg_metrics.py:113-117, utils/misc.py:135-136).

Both are functions of the per-image contingency table of the two label maps, which one HIP launch produces
(gx_label_contingency, integer atomics in LDS: bit-exact); the remaining arithmetic runs on [B, K, K]-sized device
tensors.  The reference moves every image to the host and loops in Python (numpy argmax + sklearn per image; a
boolean-mask pass over the batch per label pair)."""
import ctypes

import torch

from . import _lib
from ._lib import GenesisHipError


def contingency(segA, segB, KA, KB):
    """int32 [B, KA, KB+1]: counts[b,i,j] = #{p: segA[b,p] == i, segB[b,p] == j}; column KB = segB outside [0,KB);
    pixels with segA outside [0,KA) are skipped."""
    if not (segA.is_cuda and segB.is_cuda):
        raise GenesisHipError('metrics: label maps must live on the HIP device; there is no CPU path')
    a = segA.reshape(segA.shape[0], -1).to(torch.int64).contiguous()
    b = segB.reshape(segB.shape[0], -1).to(torch.int64).contiguous()
    if a.shape != b.shape:
        raise GenesisHipError('metrics: label maps differ in shape: %s vs %s' % (tuple(segA.shape), tuple(segB.shape)))
    B, HW = a.shape
    counts = torch.empty(B, KA, KB + 1, dtype=torch.int32, device=a.device)
    _lib.call('gx_label_contingency', ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), B, HW, KA, KB,
              ctypes.c_void_p(counts.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return counts


def _num_labels(t):
    return int(t.max().item()) + 1 if t.numel() else 1


def adjusted_rand_from_contingency(c):
    """sklearn.metrics.adjusted_rand_score restated on contingency tables c [B, R, C] (int64):
    pair confusion (tn, fp, fn, tp) from sum n_ij^2 and the squared marginals; 1.0 when fn == fp == 0."""
    c = c.to(torch.int64)
    n = c.sum((1, 2))
    sum_sq = (c * c).sum((1, 2))
    tp = sum_sq - n
    fp = (c.sum(1) ** 2).sum(1) - sum_sq          # column marginals (labels_pred in sklearn's convention)
    fn = (c.sum(2) ** 2).sum(1) - sum_sq
    tn = n * n - fp - fn - sum_sq
    tp, fp, fn, tn = [t.to(torch.float64) for t in (tp, fp, fn, tn)]
    den = (tp + fn) * (fn + tn) + (tp + fp) * (fp + tn)
    ari = 2.0 * (tp * tn - fn * fp) / den
    return torch.where((fn == 0) & (fp == 0), torch.ones_like(ari), ari)


def average_ari(log_m_k, instances, foreground_only=False):
    """utils/misc.py:101-114.  log_m_k: K x [B,1,H,W] log-masks; instances: [B,1,H,W] (or [B,H,W]) integer ground truth.
    Returns (mean ARI as a Python float, list of per-image ARI floats)."""
    masks = torch.cat(list(log_m_k), 1)                           # argmax(exp(.)) == argmax(.)
    pred = torch.argmax(masks, dim=1)
    gt = instances.to(masks.device).reshape(pred.shape[0], -1).to(torch.int64)
    K, G = masks.shape[1], _num_labels(gt)
    # rows = ground truth (row 0 = background), columns = prediction
    c = contingency(gt, pred, G, K)[:, :, :K].to(torch.int64)
    if foreground_only:
        c = c[:, 1:, :]
    ari = adjusted_rand_from_contingency(c.transpose(1, 2))       # sklearn: (labels_true=pred, labels_pred=gt)
    lst = [float(v) for v in ari.cpu()]
    return sum(lst) / len(lst), lst


def average_segcover(segA, segB, ignore_background=False):
    """utils/misc.py:173-235: covering of segA by segB, both [B,1,H,W] integer maps; negative labels in segA are
    ignore regions.  Returns (mean_sc.mean(0), scaled_sc.mean(0)) as 0-dim float32 tensors (on the device)."""
    assert segA.shape == segB.shape, '%s - %s' % (tuple(segA.shape), tuple(segB.shape))
    assert segA.shape[1] == 1 and segB.shape[1] == 1
    dev = segB.device if segB.is_cuda else segA.device
    segA, segB = segA.to(dev), segB.to(dev)
    bsz = segA.shape[0]
    KA, KB = _num_labels(segA), _num_labels(segB)
    c = contingency(segA, segB, max(KA, 1), max(KB, 1)).to(torch.int64)       # [B, KA, KB+1]
    a_i = c.sum(2)                                                            # |A == i| per image
    n_ij = c[:, :, :KB]
    b_j = n_ij.sum(1)                                                         # |(B == j) & (A >= 0)|
    mean_scores = torch.zeros(bsz, device=dev)
    N = torch.zeros(bsz, dtype=torch.int64, device=dev)
    scaled_scores = torch.zeros(bsz, device=dev)
    scaling_sum = torch.zeros(bsz, dtype=torch.int64, device=dev)
    present_a = (a_i.sum(0) > 0).cpu().tolist()                               # labels torch.unique would return
    present_b = (b_j.sum(0) > 0).cpu().tolist() if KB else []
    neg = torch.tensor(-100.0, device=dev)
    for i in range(1 if ignore_background else 0, KA):
        if not present_a[i]:
            continue
        max_iou = torch.zeros(bsz, device=dev)
        for j in range(KB):
            if not present_b[j]:
                continue
            inter = n_ij[:, i, j]
            union = a_i[:, i] + b_j[:, j] - inter
            iou = torch.where(union == 0, neg, inter.float() / union.float())
            max_iou = torch.where(iou > max_iou, iou, max_iou)
        mean_scores = mean_scores + max_iou
        N = torch.where(a_i[:, i] > 0, N + 1, N)
        scaled_scores = scaled_scores + a_i[:, i].float() * max_iou
        scaling_sum = scaling_sum + a_i[:, i]
    mean_sc = mean_scores / torch.clamp(N, min=1).float()
    scaled_sc = scaled_scores / torch.clamp(scaling_sum, min=1).float()
    return mean_sc.mean(0), scaled_sc.mean(0)
