"""Generates tests/golden/ckpt_v2_tiny.npz: the MANIFEST of a checkpoint written by the reference's OWN `save_checkpoint`
(train.py:410-420) after two iterations of the reference's training statements (train.py:223-263: GECO objective + torch.optim.Adam)
on the `tiny` GENESIS-V2 fixture's closed-form weights, inputs and noise -- and what the reference computes in the THIRD iteration
after restoring that file the way train.py:179-207 does.

    python tests/golden/make_golden_checkpoint.py

Committed: for every entry of the file the reference wrote its key path, Python type, dtype and shape, and for tensors a summary
(sum, abs-sum, strided samples) -- not the 30 MB of tensors (model + two Adam moments).  The GPU test
(tests/test_train_gpu.py::test_checkpoint_written_by_the_reference) takes the same two steps with TrainStep, requires its
state_dict() to have exactly this structure and these values, fills a dict OF THE MANIFEST'S STRUCTURE, loads it through
TrainStep.load_state_dict and reproduces the reference's third iteration.

train.py itself is imported (for `save_checkpoint`); its imports that are absent here and carry nothing of the checkpoint path
(torchvision.utils.make_grid, tensorboardX.SummaryWriter, scripts.compute_fid) are satisfied by empty stand-in modules created
below, in memory."""
import json
import os
import os.path as osp
import sys
import tempfile
import types

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, REPO)

from genesis_amd import testing as T  # noqa: E402
from oracle import ref_import as R  # noqa: E402
from oracle import v2_oracle as VO  # noqa: E402


def _standin(name, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_train():
    tv = _standin('torchvision')
    tv.utils = _standin('torchvision.utils', make_grid=None)
    _standin('tensorboardX', SummaryWriter=object)
    import scripts  # noqa: F401  (the reference's package)
    _standin('scripts.compute_fid', fid_from_model=None)
    import train
    return train


def describe(obj, path, out):
    """Flattens the checkpoint into {path: description}; tensors are summarised."""
    if torch.is_tensor(obj):
        out['manifest'].append([path, 'tensor', str(obj.dtype).replace('torch.', ''), list(obj.shape)])
        T.pack_summary('t/' + path, obj.double() if obj.dtype == torch.float64 else obj.float(), out['arrays'])
    elif isinstance(obj, dict):
        out['manifest'].append([path, type(obj).__name__, '', [len(obj)]])
        for k, v in obj.items():
            describe(v, '%s/%s%s' % (path, 'i:' if isinstance(k, int) else '', k), out)
    elif isinstance(obj, (list, tuple)):
        out['manifest'].append([path, type(obj).__name__, '', [len(obj)]])
        for i, v in enumerate(obj):
            describe(v, '%s/#%d' % (path, i), out)
    else:
        out['manifest'].append([path, type(obj).__name__, repr(obj), []])


def main():
    mods = R.import_reference()
    train = import_train()
    train.fprint = lambda *a, **k: None
    g = np.load(osp.join(HERE, 'v2_tiny.npz'), allow_pickle=False)
    cfgd = json.loads(str(g['cfg_json']))
    cfgd['pixel_std2'] = cfgd['pixel_std1']
    B, K, S, D = int(g['B']), cfgd['K_steps'], cfgd['img_size'], cfgd['feat_dim']
    cfg = R.reference_cfg(**cfgd)
    torch.manual_seed(0)
    model = mods['genesisv2_config'].load(cfg)
    model.load_state_dict(T.formula_state_dict(model.state_dict()))
    model.train()
    x = T.make_input(int(g['x_seed']), B, S)
    nseed = int(g['noise_seed'])
    GECO = mods['geco'].GECO

    def make_geco():
        return GECO(0.5655 * 3 * S * S, 1e-5 * (64 ** 2 / S ** 2), 0.99, 1.0, 1e-10, 10)     # train.py:159-167

    def iteration(model, optimiser, geco, it):
        """train.py:223-263 with the noise replayed: torch.manual_seed(s) then the forward's own draws."""
        optimiser.zero_grad()
        torch.manual_seed(nseed + 1 + it)
        _, losses, _, _, _ = model(x)
        err = losses.err.mean(0)
        kl = torch.stack(losses.kl_l_k, dim=1).mean(dim=0).sum()
        beta = float(geco.beta)
        loss = geco.loss(err, kl)
        loss.backward()
        optimiser.step()
        return [float(err + kl), float(err), float(kl), beta, float(geco.err_ema)]

    geco = make_geco()
    optimiser = torch.optim.Adam(model.parameters(), 1e-4)                                      # train.py:174-175
    hist = [iteration(model, optimiser, geco, it) for it in range(2)]
    with tempfile.TemporaryDirectory() as d:
        f = osp.join(d, 'model.ckpt-1')
        train.save_checkpoint(f, model, optimiser, geco.beta, geco.err_ema, 1, verbose=False)   # train.py:410-420
        size = osp.getsize(f)
        ckpt = torch.load(f, map_location='cpu', weights_only=False)
        out = {'manifest': [], 'arrays': {}}
        describe(ckpt, '', out)
        # the reference's resume path (train.py:179-207) into FRESH objects, then the third iteration
        torch.manual_seed(0)
        model2 = mods['genesisv2_config'].load(cfg)
        model2.train()
        optimiser2 = torch.optim.Adam(model2.parameters(), 1e-4)
        geco2 = make_geco()
        ck = torch.load(f, map_location='cpu', weights_only=False)
        msd = ck['model_state_dict']
        msd.pop('comp_vae.decoder_module.seq.0.pixel_coords.g_1', None)
        msd.pop('comp_vae.decoder_module.seq.0.pixel_coords.g_2', None)
        model2.load_state_dict(msd)
        optimiser2.load_state_dict(ck['optimiser_state_dict'])
        geco2.beta = ck['beta']
        geco2.err_ema = ck['err_ema']
        start = ck['iter_idx'] + 1
    third = iteration(model2, optimiser2, geco2, 2)
    third_direct = iteration(model, optimiser, geco, 2)
    assert third == third_direct, (third, third_direct)       # (resuming IS continuing, in the reference)
    arrays = out['arrays']
    arrays['manifest_json'] = np.array(json.dumps(out['manifest']))
    arrays['hist'] = np.array(hist + [third])
    arrays['start_iter'] = np.int64(start)
    arrays['file_bytes'] = np.int64(size)
    arrays['torch_version'] = np.array(torch.__version__)
    T.pack_summary('after3/params', torch.cat([p.detach().double().flatten().float() for p in model2.parameters()]), arrays)
    path = osp.join(HERE, 'ckpt_v2_tiny.npz')
    np.savez_compressed(path, **arrays)
    print('checkpoint of %d bytes, %d manifest entries -> %s (%d KiB)' % (size, len(out['manifest']), path, osp.getsize(path) // 1024))
    print('history', hist, 'third', third)


if __name__ == '__main__':
    main()
