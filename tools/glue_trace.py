"""Which torch (aten) ops still launch kernels inside a training step, and from which line of genesis_amd/ they come:
one eager step of the metric configuration under torch.profiler (with_stack)."""
import sys, os.path as osp, collections
sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch
import bench
from genesis_amd.trainer import TrainStep
from torch.profiler import profile, ProfilerActivity

sys.argv = [sys.argv[0]] + sys.argv[1:]          # e.g. --model monet
args = bench.parse()
model = bench.build_model(args, 'cuda')
ts = TrainStep(model, args.img, lr=1e-4, graph=False)
x = torch.rand(args.batch, 3, args.img, args.img, device='cuda')
for _ in range(3):
    ts.step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    ts.step(x)
    torch.cuda.synchronize()
seen = collections.OrderedDict()
for ev in prof.events():
    if not ev.name.startswith('aten::') or not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if 'genesis_amd' in f]
    key = (ev.name, (frames[0] if frames else '(autograd engine)') + ' ' + str(ev.input_shapes)[:80])
    seen.setdefault(key, []).append([k.name[:60] for k in ev.kernels])
for (name, where), ks in seen.items():
    print('%-28s x%d  %s   -> %s' % (name, len(ks), where[-150:], ks[0][0]))
