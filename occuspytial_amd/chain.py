"""Per-parameter sample store (API of the reference's ``occuspytial/chain.py:4-117``)."""
import numpy as np


class Chain:
    """Pre-allocated chain of named parameters.

    ``params`` maps a parameter name to its dimension; ``size`` is the capacity in draws.  Same
    public surface as the reference class: ``append``, ``expand``, ``full``, item access by name,
    ``len`` and ``repr``; a scalar parameter is stored as a 1-D array, a vector one as
    ``(size, dim)``.
    """

    def __init__(self, params, size):
        self.size = size
        self._names = tuple(params)
        self._index = 0
        self._store = {name: np.zeros((size, dim) if dim > 1 else size) for name, dim in params.items()}

    @classmethod
    def _from_arrays(cls, arrays):
        """Wrap already-filled arrays (one row per kept draw) without copying row by row."""
        first = next(iter(arrays.values()))
        out = cls.__new__(cls)
        out.size = first.shape[0]
        out._names = tuple(arrays)
        out._index = first.shape[0]
        out._store = {k: (v[:, 0] if (v.ndim > 1 and v.shape[1] == 1) else v) for k, v in arrays.items()}
        return out

    @property
    def full(self):
        """All parameters side by side, one row per stored draw."""
        cols = [v if v.ndim > 1 else v[:, None] for v in self._store.values()]
        return np.concatenate(cols, axis=1)[:self._index]

    def append(self, params):
        if self._index >= self.size:
            raise ValueError('Chain is full, cannot append any new values')
        for name, value in params.items():
            self._store[name][self._index] = value
        self._index += 1

    def expand(self, size):
        for name, value in self._store.items():
            extra = np.zeros((size,) + value.shape[1:])
            self._store[name] = np.concatenate([value, extra], axis=0)
        self.size += size

    def __getitem__(self, name):
        return self._store[name][:self._index]

    def __len__(self):
        return self._index

    def __repr__(self):
        return f'Chain(params: {self._names}, size: {self._index})'
