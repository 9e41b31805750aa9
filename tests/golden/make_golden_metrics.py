"""Generates tests/golden/metrics_*.npz from the REAL reference functions utils/misc.py:average_ari and
average_segcover (imported from /root/reference in the build container; sklearn supplies adjusted_rand_score).
Inputs: seeded random log-masks and instance maps (blocky segments, with background 0, some labels absent per
image, and -1 ignore regions in one case)."""
import os.path as osp
import sys

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import ref_import as R  # noqa: E402

CASES = {  # name: (B, K, S, n_gt_labels, ignore_regions, seed)
    'small': (3, 4, 16, 3, False, 11),
    'k7': (4, 7, 64, 5, False, 12),
    'ignore': (3, 5, 32, 4, True, 13),
    'perfect': (2, 3, 16, 3, False, 14),
}


def make_case(B, K, S, G, ignore, seed):
    g = torch.Generator().manual_seed(seed)
    # blocky ground truth: upsampled low-res random labels
    low = torch.randint(0, G, (B, 1, S // 4, S // 4), generator=g)
    inst = low.repeat_interleave(4, 2).repeat_interleave(4, 3)
    inst[0][inst[0] == G - 1] = 0                       # a label absent from image 0
    logits = torch.randn(B, K, S, S, generator=g)
    # make predictions correlated with the ground truth
    for k in range(min(K, G)):
        logits[:, k:k + 1] += 2.0 * (inst == k).float()
    if ignore:
        inst[:, :, :4, :] = -1
    log_m = torch.log_softmax(logits, 1)
    return [log_m[:, k:k + 1].contiguous() for k in range(K)], inst


def main():
    R.import_reference()
    import utils.misc as misc
    for name, (B, K, S, G, ignore, seed) in CASES.items():
        log_m_k, inst = make_case(B, K, S, G, ignore, seed)
        if name == 'perfect':
            log_m_k = [torch.log((inst == k).float().clamp_min(1e-6)) for k in range(K)]
        out = {'log_m': torch.stack(log_m_k).numpy(), 'inst': inst.numpy()}
        if not ignore:                                   # sklearn on label -1 is fine too, but train.py never does it
            for fg in (False, True):
                mean, lst = misc.average_ari(log_m_k, inst, fg)
                out['ari_mean_fg%d' % fg] = np.float64(mean)
                out['ari_list_fg%d' % fg] = np.array(lst, np.float64)
        ins_seg = torch.argmax(torch.cat(log_m_k, 1), 1, True)
        for bg in (False, True):
            m, s = misc.average_segcover(inst, ins_seg, bg)
            out['sc_mean_bg%d' % bg] = np.float32(m)
            out['sc_scaled_bg%d' % bg] = np.float32(s)
        np.savez_compressed(osp.join(HERE, 'metrics_%s.npz' % name), **out)
        print(name, {k: (v if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


if __name__ == '__main__':
    main()
