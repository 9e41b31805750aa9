"""conv3x3 forward + data gradient at the small UNet levels: direct tap-conv kernels (policy 0) vs the Winograd kernel (policy 2)."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip, _lib, profiling
torch.manual_seed(0)
for spec in sys.argv[1:]:
    N, Ci, Co, S = [int(v) for v in spec.split(':')]
    x = torch.randn(N, Ci, S, S, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05; dy = torch.randn(N, Co, S, S, device='cuda')
    for pol in (0, 2):
        _lib.call('gx_conv3x3_wino_policy', pol)
        for _ in range(3): hip.conv3x3_fwd(x, w); hip.conv3x3_dgrad(dy, w)
        torch.cuda.synchronize()
        profiling.enable(True)
        for _ in range(20): hip.conv3x3_fwd(x, w); hip.conv3x3_dgrad(dy, w)
        torch.cuda.synchronize()
        rows = {r['name']: r for r in profiling.collect()}
        profiling.enable(False)
        t = sum(r['ms'] for n, r in rows.items() if 'pack' not in n) / 40 * 1e3
        print('%-18s policy %d: %6.1f us per launch (fwd / dgrad average; %s)' % (spec, pol, t, ', '.join('%s x%d' % (n, r['launches']) for n, r in rows.items() if 'pack' not in n)), flush=True)
_lib.call('gx_conv3x3_wino_policy', 1)
