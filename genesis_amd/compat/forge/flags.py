"""Flag registry stand-in for `forge.flags` (akosiorek/forge is an empty, un-vendored
submodule in the reference, .gitmodules:1-3). Behaviour is restated from the
reference's call sites only (models/genesisv2_config.py:35-42, train.py:45-91,100):
`DEFINE_{string,integer,float,boolean}(name, default, doc)` at import time, then
`forge.config()` parses `--name value` pairs from argv. Re-definition of a flag is
tolerated (vae_config.py:32 and genesis_config.py:49 both define `pixel_bound`).
"""
import sys


class _Flags(dict):
    """dict with attribute access; mutable, like the object train.py mutates."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    @property
    def __flags(self):
        return dict(self)


FLAGS = _Flags()
_TYPES = {}
_DOCS = {}


def _define(name, default, doc, typ):
    if name not in FLAGS:
        FLAGS[name] = default
    _TYPES[name] = typ
    _DOCS[name] = doc


def DEFINE_string(name, default, doc=''):
    _define(name, default, doc, str)


def DEFINE_integer(name, default, doc=''):
    _define(name, default, doc, int)


def DEFINE_float(name, default, doc=''):
    _define(name, default, doc, float)


def _to_bool(s):
    if isinstance(s, bool):
        return s
    return str(s).lower() in ('1', 'true', 't', 'yes', 'y')


def DEFINE_boolean(name, default, doc=''):
    _define(name, default, doc, _to_bool)


DEFINE_bool = DEFINE_boolean


def parse(argv=None):
    """Parse `--name value`, `--name=value`, `--flag` / `--noflag`."""
    argv = list(sys.argv[1:] if argv is None else argv)
    i = 0
    while i < len(argv):
        tok = argv[i]
        i += 1
        if not tok.startswith('--'):
            continue
        tok = tok[2:]
        if '=' in tok:
            name, val = tok.split('=', 1)
        else:
            name, val = tok, None
        if name not in FLAGS and name.startswith('no') and name[2:] in FLAGS \
                and _TYPES.get(name[2:]) is _to_bool:
            FLAGS[name[2:]] = False
            continue
        typ = _TYPES.get(name, str)
        if val is None:
            if typ is _to_bool and (i >= len(argv) or argv[i].startswith('--')):
                val = True
            elif i < len(argv):
                val = argv[i]
                i += 1
        FLAGS[name] = typ(val) if val is not None else val
    return FLAGS
