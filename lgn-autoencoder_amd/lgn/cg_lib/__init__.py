"""
Clebsch-Gordan coefficient tables for SL(2,C) irreps (k, n) -- the product's own generator.

Reference: CGDict / _gen_cg_dict / clebschmat / clebschSU2 (lgn/cg_lib/cg_dict.py:11-436).
Same conventions: real tables ``cg[((k1,n1),(k2,n2))][(k,n)]`` of shape (d, d1*d2) ("transposed"
form, cg_dict.py:103-114), Condon-Shortley phases, an irrep (k,n) stored as the concatenation of
its SU(2) components l = |k-n|/2 .. (k+n)/2.

The SU(2) coefficients are evaluated in exact rational arithmetic (Racah's closed form, with the
square root taken once at the end), not in floating point like the reference; the tables agree to
~1e-16 (tests/test_host.py pins them to the fixture dump of the reference's tables).
At maxdim = 2 the only non-trivial block, (1,1)x(1,1)->(0,0) = 1/2 [e00 + e13 - e22 + e31], is baked
into the HIP kernels (csrc/level.hpp); these tables serve the generic path, the ``cg_dict``
attribute of the modules and the equivariance harness (LorentzD needs cg[((k,0),(0,n))][(k,n)]).
"""
from fractions import Fraction
from functools import lru_cache
from math import factorial, isqrt, sqrt
from typing import Dict, Tuple

import numpy as np
import torch

Irrep = Tuple[int, int]


def _sqrt_fraction(q: Fraction) -> float:
    """sqrt of a non-negative rational, exact when it is a perfect square."""
    n, d = q.numerator, q.denominator
    rn, rd = isqrt(n), isqrt(d)
    if rn * rn == n and rd * rd == d:
        return rn / rd
    return sqrt(n) / sqrt(d) if n < 2 ** 1000 else float(Fraction(isqrt(n * 2 ** 200), isqrt(d * 2 ** 200)))


@lru_cache(maxsize=None)
def su2_cg(tj1: int, tm1: int, tj2: int, tm2: int, tj: int, tm: int) -> float:
    """<j1 m1; j2 m2 | j m> with all arguments DOUBLED (integers).  Racah's formula."""
    if tm != tm1 + tm2 or tj < abs(tj1 - tj2) or tj > tj1 + tj2 or (tj1 + tj2 + tj) % 2:
        return 0.0
    if abs(tm1) > tj1 or abs(tm2) > tj2 or abs(tm) > tj or (tj1 + tm1) % 2 or (tj2 + tm2) % 2 or (tj + tm) % 2:
        return 0.0
    h = lambda x: x // 2  # noqa: E731  (all combinations below are even)
    f = factorial
    delta = Fraction((tj + 1) * f(h(tj + tj1 - tj2)) * f(h(tj - tj1 + tj2)) * f(h(tj1 + tj2 - tj)), f(h(tj1 + tj2 + tj) + 1))
    norm = f(h(tj + tm)) * f(h(tj - tm)) * f(h(tj1 - tm1)) * f(h(tj1 + tm1)) * f(h(tj2 - tm2)) * f(h(tj2 + tm2))
    s = Fraction(0)
    kmin = max(0, h(tj2 - tj - tm1), h(tj1 + tm2 - tj))
    kmax = min(h(tj1 + tj2 - tj), h(tj1 - tm1), h(tj2 + tm2))
    for k in range(kmin, kmax + 1):
        den = (f(k) * f(h(tj1 + tj2 - tj) - k) * f(h(tj1 - tm1) - k) * f(h(tj2 + tm2) - k)
               * f(h(tj - tj2 + tm1) + k) * f(h(tj - tj1 - tm2) + k))
        s += Fraction((-1) ** k, den)
    val = _sqrt_fraction(delta * norm * s * s)
    return val if s >= 0 else -val


def _su2_block(tj1: int, tj2: int, tj: int) -> np.ndarray:
    out = np.zeros((tj1 + 1, tj2 + 1, tj + 1))
    for a in range(tj1 + 1):
        for b in range(tj2 + 1):
            tm1, tm2 = 2 * a - tj1, 2 * b - tj2
            if abs(tm1 + tm2) <= tj:
                out[a, b, (tj + tm1 + tm2) // 2] = su2_cg(tj1, tm1, tj2, tm2, tj, tm1 + tm2)
    return out


def _recoupling(k: int, n: int) -> np.ndarray:
    """(k+1, n+1, (k+1)(n+1)): SU(2)_left x SU(2)_right -> diagonal SU(2) components of irrep (k,n)."""
    return np.concatenate([_su2_block(k, n, tj) for tj in range(abs(k - n), k + n + 1, 2)], axis=-1)


def lorentz_cg(r1: Irrep, r2: Irrep, r: Irrep) -> np.ndarray:
    """H[m1, m2, m] (d1, d2, d): couple the k- and n- spins separately, then recouple (cg_dict.py:253-281)."""
    (k1, n1), (k2, n2), (k, n) = r1, r2, r
    return np.einsum("abm,pqa,rsb,prx,qsy->xym", _recoupling(k, n), _su2_block(k1, k2, k), _su2_block(n1, n2, n),
                     _recoupling(k1, n1), _recoupling(k2, n2), optimize=True)


class CGDict:
    """Dictionary of Lorentz-group CG tables, same access pattern as the reference's CGDict."""

    def __init__(self, maxdim=None, transpose=True, dtype=torch.float64, device=None):
        self.dtype = dtype
        self.device = device if device is not None else torch.device("cpu")
        self._transpose = transpose
        self._maxdim = None
        self._cg_dict: Dict[Tuple[Irrep, Irrep], Dict[Irrep, torch.Tensor]] = {}
        if maxdim is not None:
            self.update_maxdim(maxdim)

    @property
    def transpose(self):
        return self._transpose

    @property
    def maxdim(self):
        return self._maxdim

    def update_maxdim(self, new_maxdim: int):
        if self._maxdim is not None and self._maxdim >= new_maxdim:
            return self
        rng = range(new_maxdim)
        for k1 in rng:
            for n1 in rng:
                for k2 in rng:
                    for n2 in rng:
                        pair = ((k1, n1), (k2, n2))
                        if pair in self._cg_dict:
                            continue
                        entry = {}
                        for k in range(abs(k1 - k2), k1 + k2 + 1, 2):
                            for n in range(abs(n1 - n2), n1 + n2 + 1, 2):
                                h = lorentz_cg((k1, n1), (k2, n2), (k, n))
                                mat = torch.from_numpy(np.ascontiguousarray(h.reshape(-1, h.shape[-1])))
                                if self._transpose:
                                    mat = mat.t().contiguous()
                                entry[(k, n)] = mat.to(dtype=self.dtype, device=self.device)
                        self._cg_dict[pair] = entry
        self._maxdim = new_maxdim
        return self

    def to(self, dtype=None, device=None):
        if dtype is not None:
            self.dtype = dtype
        if device is not None:
            self.device = device
        self._cg_dict = {p: {r: m.to(dtype=self.dtype, device=self.device) for r, m in e.items()}
                         for p, e in self._cg_dict.items()}
        return self

    def keys(self):
        return self._cg_dict.keys()

    def values(self):
        return self._cg_dict.values()

    def items(self):
        return self._cg_dict.items()

    def __getitem__(self, pair):
        if self._maxdim is None:
            raise ValueError("CGDict not initialised: set maxdim or call update_maxdim()")
        return self._cg_dict[pair]

    def __bool__(self):
        return self._maxdim is not None


from .zonal_functions import p_to_rep, p_cplx_to_rep, rep_to_p, normsq, normsq4, repdot  # noqa: E402  (cg_lib/__init__.py:13-23)
from .product import CGProduct, cg_product, cg_product_tau  # noqa: E402  (cg_lib/__init__.py:8-11)

__all__ = ["CGDict", "lorentz_cg", "su2_cg", "p_to_rep", "p_cplx_to_rep", "rep_to_p", "normsq", "normsq4", "repdot", "CGProduct",
           "cg_product", "cg_product_tau"]
