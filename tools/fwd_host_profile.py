"""Host-side cost of one training-mode forward of the metric configuration (the unchanged train.py loop is bound by it): cProfile over
N forwards, the GPU running behind asynchronously.   python tools/fwd_host_profile.py [N]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import genesis_amd.genesisv2_config as G  # noqa: E402
from genesis_amd.compat.attrdict import AttrDict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = AttrDict(K_steps=7, img_size=64, feat_dim=64, kernel='gaussian', semiconv=True, dynamic_K=False, klm_loss=False,
               detach_mr_in_klm=True, pixel_bound=True, autoreg_prior=True, pixel_std1=0.7, pixel_std2=0.7, debug=False, multi_gpu=False)
torch.manual_seed(0)
model = G.load(cfg).cuda().train()
x = torch.rand(32, 3, 64, 64).cuda()
for _ in range(3):
    model(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    model(x)
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
