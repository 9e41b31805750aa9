"""Stand-in for `forge.experiment_tools`, restated from call sites
(train.py:100,122,142,148,152; scripts/compute_fid.py:55-57,74,82): `load(path, cfg)`
imports a config file and calls its `load(cfg)`; `init_checkpoint` creates the run
directory and returns `(logdir, checkpoint_or_None)`; `fprint` prints (and appends to
`FPRINT_FILE` when set). Behaviour beyond those call sites is parity-unpinned.
"""
import importlib.util
import json
import os
import os.path as osp
import sys

from . import flags as _flags

EXPERIMENT_FOLDER = None
FPRINT_FILE = None


def fprint(msg, flush=False):
    print(msg, flush=bool(flush))
    if FPRINT_FILE is not None:
        with open(FPRINT_FILE, 'a') as f:
            f.write(str(msg) + '\n')


def print_flags():
    fprint(json.dumps(dict(_flags.FLAGS), indent=4, sort_keys=True, default=str))


def json_load(path):
    with open(path) as f:
        return json.load(f)


def load(path, cfg=None, *args, **kwargs):
    """Import the config file at `path` (registering its flags) and call `load(cfg)`."""
    path = osp.abspath(path)
    name = '_forge_cfg_' + osp.splitext(osp.basename(path))[0]
    if name in sys.modules and getattr(sys.modules[name], '__file__', None) == path:
        mod = sys.modules[name]
    else:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    if cfg is None:
        cfg = _flags.FLAGS
    # flags registered by the import become visible on a FLAGS-backed cfg
    if isinstance(cfg, dict):
        for k, v in _flags.FLAGS.items():
            cfg.setdefault(k, v)
    return mod.load(cfg, *args, **kwargs)


def init_checkpoint(logdir, data_config=None, model_config=None, resume=False):
    global EXPERIMENT_FOLDER, FPRINT_FILE
    os.makedirs(logdir, exist_ok=True)
    EXPERIMENT_FOLDER = logdir
    FPRINT_FILE = osp.join(logdir, 'fprint.log')
    ckpt = None
    if resume:
        cands = sorted(f for f in os.listdir(logdir) if f.startswith('model.ckpt'))
        if cands:
            ckpt = osp.join(logdir, cands[-1])
    with open(osp.join(logdir, 'flags.json'), 'w') as f:
        json.dump(dict(_flags.FLAGS), f, indent=2, sort_keys=True, default=str)
    return logdir, ckpt
