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
    ignore regions.  Returns (mean_sc.mean(0), scaled_sc.mean(0)) as 0-dim float32 tensors (on the device).

    One tensor expression over the contingency table c[b,i,j] (no loop over label pairs, no host round trip):
        |A_i| = sum_j c[b,i,:]  (column KB holds segB labels < 0),   |B_j, not ignored| = sum_i c[b,i,j],
        IoU[b,i,j] = c / (|A_i| + |B_j| - c)  (0 where the union is empty -- the reference's -100 never wins its running
        maximum, which starts at 0),   best[b,i] = max_j IoU,
        mean covering = sum_i best / #{i: |A_i| > 0},   scaled covering = sum_i |A_i| best / sum_i |A_i|.
    A label that no image of the batch holds contributes 0 to every sum, so "the labels torch.unique returns" needs no
    separate bookkeeping; `ignore_background` drops row 0 from the covered labels (its pixels still count in |B_j|)."""
    assert segA.shape == segB.shape, '%s - %s' % (tuple(segA.shape), tuple(segB.shape))
    assert segA.shape[1] == 1 and segB.shape[1] == 1
    dev = segB.device if segB.is_cuda else segA.device
    segA, segB = segA.to(dev), segB.to(dev)
    labels = torch.stack((segA.max(), segB.max())).clamp_min(0).tolist()      # the table's size: the one host read
    KA, KB = int(labels[0]) + 1, int(labels[1]) + 1
    c = contingency(segA, segB, KA, KB).to(torch.int64)                       # [B, KA, KB+1]
    b_area = c[:, :, :KB].sum(1, keepdim=True)                                # [B, 1, KB]  (background pixels of A count)
    if ignore_background:
        c = c[:, 1:]                                                          # (only A's label 0 is not covered)
    area = c.sum(2)                                                           # [B, KA']
    inter = c[:, :, :KB]
    union = area.unsqueeze(2) + b_area - inter
    iou = torch.where(union == 0, torch.zeros((), device=dev), inter.float() / union.float())
    best = iou.max(2).values if iou.shape[1] else iou.new_zeros(iou.shape[:2])
    mean_sc = best.sum(1) / (area > 0).sum(1).clamp(min=1).float()
    scaled_sc = (area.float() * best).sum(1) / area.sum(1).clamp(min=1).float()
    return mean_sc.mean(0), scaled_sc.mean(0)
