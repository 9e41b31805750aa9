"""Is the UNet backward's ~1e-3 distance from fp64 at B >= 32 arithmetic, or ReLU decisions on pre-activations within fp32
round-off of zero?  Runs the fp64 oracle UNet twice: with its own ReLUs, and with the HIP forward's ReLU pattern imposed."""
import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import v2_oracle as VO
from genesis_amd import functions as fn
from genesis_amd.compat.attrdict import AttrDict
import genesis_amd.genesisv2_config as G
cfg = VO.make_cfg(K_steps=5, img_size=64, feat_dim=64)
torch.manual_seed(0)
model = G.load(AttrDict(dict(cfg, debug=False, multi_gpu=False, dynamic_K=False)))
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.cuda()
nb = 5
for B in [int(a) for a in sys.argv[1:]] or [32]:
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, 64, 64, generator=g)
    dy = torch.randn(B, 64, 64, 64, generator=g)
    model.zero_grad()
    enc = fn.UNetEncoderFn.apply(x.cuda(), model.encoder.num_blocks, 8, *model.encoder.flat_params())
    node = enc.grad_fn
    # the HIP forward's ReLU pattern, in the oracle's F.relu call order: down 0..4, mlp x3, up 0..4, the final F.relu
    masks = []
    for i in range(nb):
        j = nb - 1 - i
        C = node.saved_down[i][1].shape[1]
        cx = node.cats[j].shape[1] - C
        masks.append((node.cats[j][:, cx:] > 0).cpu())
    for q in range(3):
        masks.append((node.mlp[1][q][1] > 0).cpu())
    for j in range(nb):
        if j < nb - 1:
            C = node.saved_up[j][0].shape[1]
            masks.append((node.cats[j + 1][:, :C, ::2, ::2] > 0).cpu())
        else:
            masks.append((enc > 0).cpu())
    masks.append(torch.ones_like(masks[-1]))          # (relu of the relu'd output: identity)
    enc.backward(dy.cuda())
    real_relu = F.relu
    results = {}
    for tag in ('own', 'hip'):
        calls = []
        def relu(t, inplace=False):
            i = len(calls); calls.append(t)
            if tag == 'own':
                return real_relu(t)
            m = masks[i].view(t.shape)
            return t * m.to(t.dtype)
        F.relu = relu
        try:
            p = {k: v.clone().double().requires_grad_(True) for k, v in sd.items() if k.startswith('encoder.')}
            y = F.relu(VO.unet_forward(p, x.double(), nb))
            y.backward(dy.double())
        finally:
            F.relu = real_relu
        results[tag] = {k: v.grad for k, v in p.items()}
        if tag == 'own':
            pre = [c.detach() for c in calls]
    flips = [(int(((pre[i] > 0) != masks[i].view(pre[i].shape)).sum()), float(pre[i][(pre[i] > 0) != masks[i].view(pre[i].shape)].abs().max()) if ((pre[i] > 0) != masks[i].view(pre[i].shape)).any() else 0.0) for i in range(len(pre) - 1)]
    print('B=%d: ReLU decisions differing from fp64 per layer (count, largest |pre-activation| among them): %s' % (B, flips))
    worst = {'own': (0, ''), 'hip': (0, '')}
    for k, prm in model.named_parameters():
        if not k.startswith('encoder.'):
            continue
        for tag in ('own', 'hip'):
            g64 = results[tag][k]
            e = float((prm.grad.double().cpu() - g64).norm() / g64.norm())
            worst[tag] = max(worst[tag], (e, k))
    print('   worst gradient error vs fp64 with its own ReLUs: %.3e (%s); vs fp64 on the HIP ReLU pattern: %.3e (%s)' % (worst['own'] + worst['hip']))
