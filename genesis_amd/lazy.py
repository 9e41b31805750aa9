"""Containers for the model outputs: an AttrDict whose derived (visualisation-only) entries are computed on first
access, and a per-slot list that remembers the stacked tensor it was unbound from (so the training step can reduce
the KL terms in one launch instead of re-stacking the list)."""
from genesis_amd import compat as _compat

_compat.install()
from attrdict import AttrDict  # noqa: E402


class Lazy(object):
    """A value computed by `thunk()` the first time its key is read."""
    __slots__ = ('thunk',)

    def __init__(self, thunk):
        self.thunk = thunk


class LazyAttrDict(AttrDict):
    """AttrDict (same attribute / item access as the reference's outputs); Lazy entries resolve on access."""

    def _resolve(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, Lazy):
            v = v.thunk()
            dict.__setitem__(self, key, v)
        return v

    def _resolve_all(self):
        for k in list(dict.keys(self)):
            self._resolve(k)

    def __getitem__(self, key):
        return self._resolve(key)

    def get(self, key, default=None):
        return self._resolve(key) if key in self else default

    def items(self):
        self._resolve_all()
        return dict.items(self)

    def values(self):
        self._resolve_all()
        return dict.values(self)

    def pop(self, key, *default):
        if key in self:
            self._resolve(key)
        return dict.pop(self, key, *default)

    def copy(self):
        self._resolve_all()
        return AttrDict(self)


class SlotList(list):
    """list of per-slot tensors (what the reference returns) + the [K, ...] tensor they are views of."""

    def __init__(self, items, stacked=None):
        super().__init__(items)
        self.stacked = stacked
