"""CPU restatement of the reference's segmentation metrics -- TEST INFRASTRUCTURE ONLY (tests/ and the golden
generator import it; the product path is genesis_amd/metrics.py on the HIP contingency kernel).

average_ari      utils/misc.py:101-114 (numpy argmax over the stacked masks, sklearn adjusted_rand_score per image)
average_segcover utils/misc.py:173-235 (+ iou_binary :162-170)

Third-party arithmetic: scikit-learn's adjusted_rand_score (reference pins scikit-learn==0.22, environment.yml:118;
1.7.2 here).  Its published algorithm is restated below on the contingency matrix (pair-confusion form used since
0.24; algebraically identical to the 0.22 comb(n,2) form).  Pinned by tests/golden/metrics_*.npz, generated from the
real reference functions (tests/golden/make_golden_metrics.py)."""
import numpy as np


def adjusted_rand_score(labels_true, labels_pred):
    lt, lp = np.asarray(labels_true).ravel(), np.asarray(labels_pred).ravel()
    _, ti = np.unique(lt, return_inverse=True)
    _, pi = np.unique(lp, return_inverse=True)
    c = np.zeros((ti.max() + 1, pi.max() + 1), dtype=np.int64)
    np.add.at(c, (ti, pi), 1)
    n = np.int64(lt.size)
    sum_sq = (c * c).sum()
    tp = sum_sq - n
    fp = (c.sum(0) ** 2).sum() - sum_sq
    fn = (c.sum(1) ** 2).sum() - sum_sq
    tn = n * n - fp - fn - sum_sq
    if fn == 0 and fp == 0:
        return 1.0
    tp, fp, fn, tn = float(tp), float(fp), float(fn), float(tn)
    return 2.0 * (tp * tn - fn * fp) / ((tp + fn) * (fn + tn) + (tp + fp) * (fp + tn))


def average_ari(log_m_k, instances, foreground_only=False):
    """log_m_k: list of K arrays [B,1,H,W]; instances [B,1,H,W] ints."""
    masks = np.exp(np.stack([np.asarray(m) for m in log_m_k], axis=4))
    ari = []
    for i in range(masks.shape[0]):
        pred = np.argmax(masks[i:i + 1], axis=-1).flatten()
        gt = np.asarray(instances[i]).flatten()
        if foreground_only:
            pred = pred[np.where(gt > 0)]
            gt = gt[np.where(gt > 0)]
        ari.append(adjusted_rand_score(pred, gt))
    return sum(ari) / len(ari), ari


def average_segcover(segA, segB, ignore_background=False):
    """Returns (mean_sc.mean(), scaled_sc.mean()) as float32 numpy scalars, following the reference's float32 steps."""
    segA, segB = np.asarray(segA), np.asarray(segB)
    assert segA.shape == segB.shape and segA.shape[1] == 1
    bsz = segA.shape[0]
    nonignore = segA >= 0
    mean_scores = np.zeros(bsz, np.float32)
    N = np.zeros(bsz, np.int64)
    scaled_scores = np.zeros(bsz, np.float32)
    scaling_sum = np.zeros(bsz, np.int64)
    iter_a = np.unique(segA[segA > 0] if ignore_background else segA[segA >= 0]).tolist()
    iter_b = np.unique(segB[segB >= 0]).tolist()
    for i in iter_a:
        binA = segA == i
        if not binA.any():
            continue
        max_iou = np.zeros(bsz, np.float32)
        for j in iter_b:
            binB = (segB == j) & nonignore
            if not binB.any():
                continue
            inter = (binA & binB).sum((1, 2, 3))
            union = (binA | binB).sum((1, 2, 3))
            with np.errstate(divide='ignore', invalid='ignore'):
                iou = np.where(union == 0, np.float32(-100.0), inter.astype(np.float32) / union.astype(np.float32))
            max_iou = np.where(iou > max_iou, iou, max_iou).astype(np.float32)
        mean_scores += max_iou
        cnt = binA.sum((1, 2, 3))
        N = np.where(cnt > 0, N + 1, N)
        scaled_scores += cnt.astype(np.float32) * max_iou
        scaling_sum += cnt
    mean_sc = mean_scores / np.maximum(N, 1).astype(np.float32)
    scaled_sc = scaled_scores / np.maximum(scaling_sum, 1).astype(np.float32)
    return np.float32(mean_sc.mean(dtype=np.float32)), np.float32(scaled_sc.mean(dtype=np.float32))
