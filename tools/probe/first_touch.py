"""What does a reader pay for being the FIRST reader of a tensor another kernel has just written?  (DESIGN.md finding 49 (c): the
canvas convs got 8 - 14 us slower per launch when the amax pass in front of them disappeared.)  Producer: a kernel that writes the
148 MB canvas tensor; consumer: the canvas conv (gx_conv3x3_bias_act_fwd) or a plain streaming read; in between nothing / a full read
pass / a read of one cache line per 4 KB / per 64 KB.  Times by HIP events, medians of 30 repetitions.
usage: python tools/probe/first_touch.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from genesis_amd import hip_ops as hip  # noqa: E402

dev = 'cuda'
N, C, S = 224, 32, 72
torch.manual_seed(0)
src = torch.randn(N, C, S, S, device=dev)
x = torch.empty_like(src)
w = torch.randn(C, C, 3, 3, device=dev) * 0.06
b = torch.zeros(C, device=dev)
flat = x.view(-1)
am = hip.amax_of(src)          # the input's maxima handed in: the conv makes no pass of its own
other = torch.randn(64 * 1024 * 1024, device=dev)      # 256 MB of unrelated traffic


def med(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2]


def run(consumer, between, flush):
    out = []
    for _ in range(30):
        if flush:
            other.mul_(1.0)                      # push everything else out of the caches
        x.copy_(src)                             # the producer: writes the tensor
        if between == 'full':
            flat.sum()
        elif between == 'amax':
            hip.amax_of(x)
        elif between == 'line4k':
            flat.view(-1, 1024)[:, 0].sum()      # one element per 4 KB
        elif between == 'line64k':
            flat.view(-1, 16384)[:, 0].sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if consumer == 'conv':
            hip.conv3x3_bias_act_fwd(x, w, b, 'relu', amax_in=am)
        else:
            flat.sum()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return med(out)


for consumer in ('conv', 'read'):
    for flush in (False, True):
        row = []
        for between in ('none', 'full', 'amax', 'line4k', 'line64k'):
            row.append('%s %.1f' % (between, run(consumer, between, flush)))
        print('consumer %-4s  other traffic before the producer: %-5s  us: %s' % (consumer, flush, '  '.join(row)), flush=True)
