"""Stand-ins for the two pure-Python packages the reference's model-config API
needs but which are absent (`forge`: empty submodule; `attrdict`: not installed).
`install()` registers them in `sys.modules` only if the real ones do not import."""
import importlib
import sys


def install():
    for name, target in (('attrdict', 'genesis_amd.compat.attrdict'),
                         ('forge', 'genesis_amd.compat.forge')):
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except ImportError:
            mod = importlib.import_module(target)
            sys.modules[name] = mod
            if name == 'forge':
                sys.modules['forge.flags'] = mod.flags
                sys.modules['forge.experiment_tools'] = mod.experiment_tools
