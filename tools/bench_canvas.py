"""The BroadcastDecoder's canvas layer in isolation: conv3x3 32 -> 32 (+ bias + ELU) on [224, 32, 72, 72], its data gradient
and its weight gradient; prints the HIP-event time of each (tools/pmc_kernel.sh runs this under PMC passes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from genesis_amd import hip_ops as hip  # noqa: E402

N, C, S = 224, 32, 72
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = torch.randn(N, C, S, S, device='cuda')
dy = torch.randn(N, C, S, S, device='cuda')
w = torch.randn(C, C, 3, 3, device='cuda') * 0.06
b = torch.randn(C, device='cuda')


def timed(fn, name, flops):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print('%-28s %8.1f us  %6.1f TF/s' % (name, us, flops / us * 1e-6))


fl = 2.0 * N * C * C * 9 * S * S
timed(lambda: hip.conv3x3_bias_act_fwd(x, w, b, 'elu'), 'forward (bias + ELU)', fl)
timed(lambda: hip.conv3x3_dgrad(dy, w), 'data gradient', fl)
if hip.conv3x3_wgrad_quad_supported(N, C, S, S):
    timed(lambda: hip.conv3x3_wgrad_quad(x, dy), 'weight gradient (quad)', fl)
