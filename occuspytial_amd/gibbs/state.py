"""Attribute namespaces of the sampler (reference ``occuspytial/gibbs/state.py``)."""
from types import SimpleNamespace


class _Storage(SimpleNamespace):
    def __getitem__(self, key):
        return self.__dict__[key]


class State(_Storage):
    """Mutable values; iterating yields the attribute names in assignment order."""

    def __iter__(self):
        yield from self.__dict__


class FixedState(_Storage):
    """Write-once values: re-assigning an attribute raises ``KeyError``."""

    def __setattr__(self, name, value):
        if name in self.__dict__:
            raise KeyError('cannot change attributes already set')
        super().__setattr__(name, value)
