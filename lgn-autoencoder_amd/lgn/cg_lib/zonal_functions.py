"""
lgn.cg_lib.zonal_functions -- the basis-change / Minkowski-product helpers the reference's callers import from this
module path (utils/losses/chamfer_loss/distance_sq.py:3: ``p_cplx_to_rep, repdot``; lgn/cg_lib/__init__.py:13-23
re-exports ``p_to_rep, rep_to_p, normsq, normsq4``).

These sit in the reference's *loss* and IO code, not on the message-passing hot path: inside the networks the same
basis changes and norms are recomputed by the HIP kernels (csrc/level_dev.hpp, csrc/net_kernels.hip).  They are
therefore plain PyTorch glue (lgn/ops.py) with the reference's call shapes and return types
(lgn/cg_lib/zonal_functions.py:201-218,251-446).
"""
import numpy as np
import torch

from .. import ops
from ..g_lib import GTensor, GVec


def p_to_rep(p):
    """real Cartesian (...,4) -> GVec {(1,1): (2,...,1,4)} in the canonical basis (zonal_functions.py:251-289)."""
    return GVec({(1, 1): ops.cart_to_canonical_real(p).unsqueeze(-2)})


def p_cplx_to_rep(p):
    """complex Cartesian (2,...,4), tensor or {(1,1): tensor} -> GVec {(1,1): (2,...,4)} (zonal_functions.py:292-341)."""
    if type(p) == dict or isinstance(p, GTensor):
        p = p[(1, 1)]
    assert p.shape[0] == 2, "The first dimension of p must be the complex dimension of size 2"
    return GVec({(1, 1): ops.cart_to_canonical_cplx(p)})


def rep_to_p(rep):
    """canonical (2,...,4), tensor or GVec -> complex Cartesian tensor (2,...,4) (zonal_functions.py:344-381)."""
    if isinstance(rep, GTensor) or type(rep) == dict:
        rep = rep[(1, 1)]
    assert rep.shape[0] == 2, "the first dimension of rep must be the complex dimension"
    return ops.canonical_to_cart(rep)


def normsq4(p):
    """2 E^2 - sum_mu p_mu^2 of real Cartesian 4-vectors (zonal_functions.py:201-218)."""
    return ops.normsq4(p)


def metric(key):
    """Invariant bilinear form of irrep (k,n) in the canonical basis: (-1)^(l+m) delta_{l l'} delta_{m,-m'}
    over the SU(2) components l = |k-n|/2 .. (k+n)/2 (zonal_functions.py:396-407)."""
    k, n = key
    idx = [(l, m) for l in np.arange(abs(k - n) / 2, (k + n) / 2 + 1, 1) for m in np.arange(-l, l + 1, 1)]
    met = torch.zeros(len(idx), len(idx), dtype=torch.float64)
    for a, (l, m) in enumerate(idx):
        for b, (ll, mm) in enumerate(idx):
            if l == ll and m + mm == 0:
                met[a, b] = (-1) ** int(l + m)
    return met


def repdot(rep1, rep2):
    """Lorentz-invariant complex dot product per irrep, {key: (2,...,1)} (zonal_functions.py:410-438)."""
    assert {k: v.shape for k, v in rep1.items()} == {k: v.shape for k, v in rep2.items()}, \
        "rep1 and rep2 must have all the same irreps of the same shapes!"
    out = {}
    for key in rep1.keys():
        a, b = rep1[key], rep2[key]
        met = metric(key).to(device=a.device, dtype=a.dtype)
        mb = torch.stack([b[0] @ met.t(), b[1] @ met.t()], 0)       # (met b)_a = sum_b met[a,b] b_b
        out[key] = torch.stack([(a[0] * mb[0]).sum(-1) - (a[1] * mb[1]).sum(-1),
                                (a[0] * mb[1]).sum(-1) + (a[1] * mb[0]).sum(-1)], 0).unsqueeze(-1)
    return out


def normsq(p):
    """repdot(p, p) (zonal_functions.py:384-393)."""
    if not (type(p) is dict or isinstance(p, GTensor)):
        p = {(1, 1): p}
    return repdot(p, p)


def eps(data):
    """zonal_functions.py:441-446 (None for non-fp64 tensors, like the reference)."""
    if isinstance(data, torch.Tensor):
        if data.dtype in [torch.float64, torch.double]:
            return 1e-16
    else:
        return 1e-12


__all__ = ["p_to_rep", "p_cplx_to_rep", "rep_to_p", "normsq4", "normsq", "metric", "repdot", "eps"]
