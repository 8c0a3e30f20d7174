"""Small host-side helpers (reference: torch_scae/general_utils.py:9-11 and
monty.collections.AttrDict, which the reference imports but this package does
not depend on)."""
import operator
from functools import reduce


class AttrDict(dict):
    """dict whose items are also attributes (get / set / del), the result
    container of every module here -- same behaviour as monty 3.0.2's."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


class LazyAttrDict(AttrDict):
    """AttrDict some of whose entries are computed on first access:
    ``d._lazy = {key: thunk}``; ``thunk(d)`` stores the entry (and possibly
    others).  Lookups by item, attribute, ``get`` and ``in`` see lazy entries;
    plain iteration only lists what has been materialised."""

    def __missing__(self, key):
        lazy = dict.get(self, "_lazy")
        if lazy and key in lazy:
            lazy.pop(key)(self)
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __getattr__(self, key):          # reached only when the entry is absent
        if key.startswith("__"):
            raise AttributeError(key)
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __contains__(self, key):
        lazy = dict.get(self, "_lazy")
        return dict.__contains__(self, key) or bool(lazy and key in lazy)

    def get(self, key, default=None):
        return self[key] if key in self else default


def prod(iterable):
    return reduce(operator.mul, iterable, 1)
