"""The canvas conv (32 -> 32 on [N, 32, 72, 72]) forward (+ bias + ReLU / ELU), data gradient and data gradient with the previous
layer's activation backward: time per call and a checksum file, so that two runs (GENESIS_KQ_C3P=0 / 1) can be compared bit for bit:
    GENESIS_KQ_C3P=0 python tools/c3p_check.py /tmp/a.pt; python tools/c3p_check.py /tmp/b.pt /tmp/a.pt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from genesis_amd import hip_ops as hip  # noqa: E402

out_path = sys.argv[1]
ref_path = sys.argv[2] if len(sys.argv) > 2 else None
res = {}
for (N, C, S) in ((224, 32, 72), (5, 32, 72), (224, 16, 72), (37, 32, 40)):
    g = torch.Generator(device='cuda').manual_seed(N + C)
    x = torch.randn(N, C, S, S, device='cuda', generator=g)
    dy = torch.randn(N, C, S, S, device='cuda', generator=g)
    w = torch.randn(C, C, 3, 3, device='cuda', generator=g) * 0.06
    b = torch.randn(C, device='cuda', generator=g)

    def timed(fn, name):
        y = fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        fl = 2.0 * N * C * C * 9 * S * S
        print('%-14s %-34s %8.1f us  %6.1f TF/s' % ((N, C, S), name, us, fl / us * 1e-6), flush=True)
        return y
    res[(N, C, S, 'fwd_relu')] = timed(lambda: hip.conv3x3_bias_act_fwd(x, w, b, 'relu'), 'forward (bias + ReLU)').cpu()
    res[(N, C, S, 'fwd_elu')] = timed(lambda: hip.conv3x3_bias_act_fwd(x, w, b, 'elu'), 'forward (bias + ELU)').cpu()
    res[(N, C, S, 'dgrad')] = timed(lambda: hip.conv3x3_dgrad(dy, w), 'data gradient').cpu()
    if hip.conv3x3_dgrad_act_supported(N, C, C, S, S):
        xo = torch.relu(x)
        r = timed(lambda: hip.conv3x3_dgrad_act(dy, w, xo, 'relu'), 'data gradient x act\'(mask)')
        res[(N, C, S, 'dgrad_act')] = r[0].cpu()
    # against torch on a small case
    if N <= 40:
        ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1).relu()
        print('   forward vs fp64: %.2e' % float((res[(N, C, S, 'fwd_relu')].double().cuda() - ref).norm() / ref.norm()))
torch.save(res, out_path)
if ref_path:
    ref = torch.load(ref_path)
    for k in res:
        same = torch.equal(res[k], ref[k])
        print(k, 'bit-identical' if same else 'DIFFERS: max abs %.3e' % float((res[k] - ref[k]).abs().max()))
