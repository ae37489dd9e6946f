"""``Data``: the dict-backed ragged container of the reference (``occuspytial/data.pyx:34-147``).

Kept for API compatibility (``sampler.W[[1, 3]]``, ``sampler.y.surveyed`` ...).  The sampler itself
never touches it on the per-iteration path: the ragged blocks are flattened once
(:mod:`occuspytial_amd._problem`) and live in HBM.
"""
import numpy as np


class Data:
    def __init__(self, data):
        if not isinstance(data, dict):
            raise TypeError('data must be a dict mapping site number to array')
        self._data = data
        self.surveyed = list(data)

    def visits(self, sites):
        """Number of visits of one site (int) or of several (tuple), as ``data.pyx:92-115``."""
        if isinstance(sites, (list, tuple)):
            return tuple(self._data[s].shape[0] for s in sites)
        return self._data[sites].shape[0]

    def __getitem__(self, sites):
        if isinstance(sites, (list, tuple)):
            return np.concatenate([self._data[s] for s in sites], axis=0)
        return self._data[sites]

    def __len__(self):
        return len(self._data)

    def __reduce__(self):
        return self.__class__, (self._data,)
