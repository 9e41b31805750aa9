"""Minimal stand-in for the `attrdict` package (attrdict==2.0.1 is pinned by the
reference, environment.yml:94, but is not installed here).

Only the behaviour the hot path's callers rely on is provided:
  * `obj.key` and `obj['key']` access, `'k' in obj`, `.items()`, `.update()`;
  * attribute access returns list/tuple values as *tuples* and dict values wrapped
    as AttrDict, item access returns the raw stored object -- this is why
    `comp_stats['mu_k'].append(mu)` followed by `comp_stats.z_k` works in
    models/genesisv2_config.py:145-164.
No arithmetic lives here.
"""


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    @staticmethod
    def _wrap(value):
        if isinstance(value, AttrDict):
            return value
        if isinstance(value, dict):
            return AttrDict(value)
        if isinstance(value, (list, tuple)):
            return tuple(AttrDict._wrap(v) for v in value)
        return value

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        try:
            return self._wrap(self[name])
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def copy(self):
        return AttrDict(self)


class AttrDefault(AttrDict):
    def __init__(self, default_factory=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        object.__setattr__(self, '_default_factory', default_factory)

    def __missing__(self, key):
        factory = object.__getattribute__(self, '_default_factory')
        if factory is None:
            raise KeyError(key)
        self[key] = value = factory()
        return value
