"""Import the *real* reference (applied-ai-lab/genesis @ /root/reference) in the build
container, to pin the oracle restatement and to generate golden vectors.

TEST INFRASTRUCTURE ONLY. /root/reference does not exist on the GPU box, nothing on
the product path may import this. The four absent pure-Python deps are satisfied by
stand-ins that carry no arithmetic: `attrdict`, `forge`
(genesis_amd/compat), `simplejson`, `tensorflow` (oracle/stubs).
"""
import os
import os.path as osp
import sys

REFERENCE_ROOT = os.environ.get('GENESIS_REFERENCE_ROOT', '/root/reference')
_HERE = osp.dirname(osp.abspath(__file__))


def reference_available():
    return osp.isdir(osp.join(REFERENCE_ROOT, 'models'))


def import_reference():
    """Returns a dict of the reference's model-config modules."""
    if not reference_available():
        raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
    repo = osp.dirname(_HERE)
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from genesis_amd import compat
    compat.install()
    stubs = osp.join(_HERE, 'stubs')
    for p in (REFERENCE_ROOT, stubs):
        if p not in sys.path:
            sys.path.append(p)
    import importlib
    mods = {}
    for name in ('genesisv2_config', 'genesis_config', 'monet_config', 'vae_config'):
        mods[name] = importlib.import_module('models.' + name)
    mods['geco'] = importlib.import_module('utils.geco')
    mods['blocks'] = importlib.import_module('modules.blocks')
    mods['attention'] = importlib.import_module('modules.attention')
    return mods


def reference_cfg(**overrides):
    """cfg object for the reference's load(cfg): registered flag defaults + overrides."""
    from forge import flags
    cfg = type(flags.FLAGS)(flags.FLAGS)
    cfg.update(dict(debug=False, multi_gpu=False))
    cfg.update(overrides)
    return cfg
