"""
Irrep containers with the surface of the reference's lgn.g_lib (g_tensor.py, g_vec.py,
g_scalar.py, g_tau.py) that callers of the hot path rely on: dict-like access keyed by
(k, n), ``.items()/.keys()/.values()``, ``__iter__`` yielding (key, part) pairs,
``.device/.dtype/.tau``, ``__class__(dict)`` construction (used by
lgn/g_lib/rotations.py:36-51 and lgn/models/autotest/utils.py:11-45).

Layouts: GVec part (2, *batch, C, d); GScalar part (2, *batch, C); planar complex on dim 0.
The arithmetic of the reference's g_torch / cplx_lib lives in the HIP kernels, not here.
"""
from typing import Dict, Tuple

import torch

Irrep = Tuple[int, int]


def irrep_dim(key: Irrep) -> int:
    return (key[0] + 1) * (key[1] + 1)


class GTau:
    """Multiplicity bookkeeping {(k,n): channels} (reference: lgn/g_lib/g_tau.py:6-171)."""

    def __init__(self, tau):
        if isinstance(tau, GTau):
            tau = tau._tau
        elif not isinstance(tau, dict):
            tau = tau.tau
            if isinstance(tau, GTau):
                tau = tau._tau
        for k, v in tau.items():
            if not (isinstance(k, tuple) and isinstance(v, int)):
                raise ValueError(f"GTau needs {{(k,n): int}}, got {k!r}: {v!r}")
        self._tau = dict(tau)

    def keys(self):
        return self._tau.keys()

    def values(self):
        return self._tau.values()

    def items(self):
        return self._tau.items()

    def __iter__(self):
        return iter(self._tau.items())

    def __getitem__(self, key):
        return self._tau[key]

    def __setitem__(self, key, val):
        self._tau[key] = val

    def __len__(self):
        return len(self._tau)

    def __eq__(self, other):
        other = other._tau if isinstance(other, GTau) else dict(other)
        return self._tau == other

    @property
    def maxdim(self):
        return max(max(k) for k in self._tau) + 1

    @property
    def channels(self):
        vals = set(v for v in self._tau.values() if v)
        return vals.pop() if len(vals) == 1 else None

    def __repr__(self):
        return str(self._tau)


class GTensor:
    """Ordered mapping irrep -> tensor (reference: lgn/g_lib/g_tensor.py:9-472)."""
    cdim = None
    rdim = None
    zdim = 0

    def __init__(self, data, ignore_check: bool = False):
        if isinstance(data, GTensor):
            data = data._data
        self._data: Dict[Irrep, torch.Tensor] = {
            k: v for k, v in data.items() if isinstance(k, tuple) and torch.is_tensor(v) and v.numel() > 0}
        if not ignore_check:
            self.check_data(self._data)

    def check_data(self, data):
        for key, part in data.items():
            if part.shape[self.zdim] != 2:
                raise ValueError(f"complex dimension of part {key} must have length 2, got {tuple(part.shape)}")

    # dict surface
    def keys(self):
        return self._data.keys()

    def values(self):
        return self._data.values()

    def items(self):
        return self._data.items()

    def pop(self, key):
        return self._data.pop(key)

    def __iter__(self):
        return iter(self._data.items())

    def __len__(self):
        return len(self._data)

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            raise ValueError(f"keys of G tensors are (k, n) tuples, got {key!r}")
        return self._data[key]

    def __setitem__(self, key, val):
        self._data[key] = val

    def __contains__(self, key):
        return key in self._data

    @property
    def data(self):
        return self._data

    @property
    def device(self):
        return next(iter(self._data.values())).device

    @property
    def dtype(self):
        return next(iter(self._data.values())).dtype

    @property
    def shapes(self):
        return {k: v.shape for k, v in self._data.items()}

    @property
    def maxdim(self):
        return max(max(k) for k in self._data) + 1

    @property
    def tau(self):
        return GTau({k: int(v.shape[self.cdim]) for k, v in self._data.items()})

    def truncate(self, maxdim):
        return self.__class__({k: v for k, v in self._data.items() if max(k) < maxdim})

    def to(self, *args, **kwargs):
        return self.__class__({k: v.to(*args, **kwargs) for k, v in self._data.items()})

    def __repr__(self):
        return f"{self.__class__.__name__}({ {k: tuple(v.shape) for k, v in self._data.items()} })"


class GVec(GTensor):
    """(2, *batch, C, d) parts (reference: lgn/g_lib/g_vec.py:11-89)."""
    cdim = -2
    rdim = -1

    def check_data(self, data):
        super().check_data(data)
        for key, part in data.items():
            if part.shape[-1] != irrep_dim(key):
                raise ValueError(f"part {key} must have last dimension {irrep_dim(key)}, got {tuple(part.shape)}")


class GScalar(GTensor):
    """(2, *batch, C) parts (reference: lgn/g_lib/g_scalar.py:7-59)."""
    cdim = -1
    rdim = None


__all__ = ["GTau", "GTensor", "GVec", "GScalar", "irrep_dim"]
