"""A few launches of the big transposed-conv layers (k-quad kernels) for PMC passes / timing.  argv: fwd|dgrad|c3 N S"""
import os, sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
from genesis_amd import hip_ops as hip, _lib
which = sys.argv[1] if len(sys.argv) > 1 else 'dgrad'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 224
s = int(sys.argv[3]) if len(sys.argv) > 3 else 32
pol = int(sys.argv[4]) if len(sys.argv) > 4 else 1
_lib.call('gx_kq_policy', pol)
torch.manual_seed(0)
x = torch.randn(N, 64, s, s, device='cuda')
w = torch.randn(64, 64, 5, 5, device='cuda') * 0.05
b = torch.randn(64, device='cuda')
dy = torch.randn(N, 64, 2 * s, 2 * s, device='cuda')
w3 = torch.randn(64, 64, 3, 3, device='cuda') * 0.05
fn = {'fwd': lambda: hip.deconv5x5s2_fwd(x, w, b), 'dgrad': lambda: hip.deconv5x5s2_dgrad(dy, w),
      'c3': lambda: hip.conv3x3_fwd(x, w3), 'wgrad': lambda: hip.deconv5x5s2_wgrad(x, dy),
      'wgrad3': lambda: hip.conv3x3_wgrad(x, x)}[which]
for _ in range(6):
    fn()
torch.cuda.synchronize()
