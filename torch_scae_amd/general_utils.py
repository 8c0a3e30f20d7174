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


def prod(iterable):
    return reduce(operator.mul, iterable, 1)
