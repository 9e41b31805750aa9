"""The reference's training loop UNCHANGED (train.py:223-263) on the HIP model, eager: wall time per iteration, and -- under
`rocprofv3 --kernel-trace --stats` -- the kernel time behind it (is the loop bound by the GPU or by the host's launch rate?).
    python tools/ref_loop_time.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import genesis_amd.genesisv2_config as G  # noqa: E402
from genesis_amd.compat.attrdict import AttrDict  # noqa: E402
from genesis_amd.geco import make_geco  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = AttrDict(K_steps=7, img_size=64, feat_dim=64, kernel='gaussian', semiconv=True, dynamic_K=False, klm_loss=False,
               detach_mr_in_klm=True, pixel_bound=True, autoreg_prior=True, pixel_std1=0.7, pixel_std2=0.7, debug=False, multi_gpu=False)
torch.manual_seed(0)
model = G.load(cfg).cuda().train()
optimiser = torch.optim.Adam(model.parameters(), lr=1e-4)
geco = make_geco(64)
g = torch.Generator().manual_seed(1234)
batches = [torch.rand(32, 3, 64, 64, generator=g).cuda() for _ in range(4)]
sync = os.environ.get('REF_LOOP_SYNC', '1') != '0'


T = {'zero_grad': 0.0, 'forward': 0.0, 'losses': 0.0, 'backward': 0.0, 'step': 0.0}
SECT = bool(os.environ.get('REF_LOOP_SECTIONS'))         # host-side time per section (the GPU runs behind, asynchronously)


def iteration(x):
    if SECT:
        return iteration_sections(x)
    optimiser.zero_grad()
    output, losses, stats, att_stats, comp_stats = model(x)
    err = losses.err.mean(0)
    kl_l = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
    elbo = (err + kl_l).detach()
    mse_batched = ((x - output) ** 2).mean((1, 2, 3)).detach()
    mse, rmse = mse_batched.mean(0), mse_batched.sqrt().mean(0)
    loss = geco.loss(err, kl_l)
    if sync:
        float(geco.state[1])            # utils/geco.py:45
    loss.backward()
    optimiser.step()
    return elbo


def iteration_sections(x):
    t = [time.perf_counter()]
    optimiser.zero_grad(); t.append(time.perf_counter())
    output, losses, stats, att_stats, comp_stats = model(x); t.append(time.perf_counter())
    err = losses.err.mean(0)
    kl_l = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
    elbo = (err + kl_l).detach()
    mse_batched = ((x - output) ** 2).mean((1, 2, 3)).detach()
    mse, rmse = mse_batched.mean(0), mse_batched.sqrt().mean(0)
    loss = geco.loss(err, kl_l)
    if sync:
        float(geco.state[1])
    t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    optimiser.step(); t.append(time.perf_counter())
    for k, a, b in zip(T, t[:-1], t[1:]):
        T[k] += b - a
    return elbo


for i in range(5):
    iteration(batches[i % 4])
for k in T:
    T[k] = 0.0
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    e = iteration(batches[i % 4])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if SECT:
    print('host time per section (ms): ' + ', '.join('%s %.3f' % (k, 1e3 * v / steps) for k, v in T.items()))
print('reference loop: %.3f ms / iteration (%.0f img/s), host sync per iteration: %s, final ELBO %.2f' % (1e3 * dt / steps, 32 * steps / dt, sync, float(e)))
