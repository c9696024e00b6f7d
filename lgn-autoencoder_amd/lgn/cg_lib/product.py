"""
Clebsch-Gordan product of two G vectors as a module / function of its own (reference: CGProduct and cg_product,
lgn/cg_lib/cg_ops.py:10-218; complex_kron_product :221-297; cg_product_tau, lgn/cg_lib/cg_ops_tau.py:6-44).

Inside LGNEncoder / LGNDecoder the product never runs by itself -- the level kernels fuse it with the edge network and CatMix --
but the reference exports it, so this package does too: one native call per pair of irreps (lgn_cg_product_fwd/bwd_f64,
csrc/cg_product.hip: the pair's stacked Clebsch-Gordan matrix in CSR form), under autograd.  Same conventions as the
reference: parts are (2, *batch, C, d); the channel axis is carried through (no mixing); with ``aggregate`` one operand is
edge-like, (2, B, N, N, C, d), and the Kronecker products are summed over the neighbour index before the CG matrix is
applied; the irreps a pair produces are appended to the output part of their key in the order the pairs are visited
(dict order of rep1, then of rep2), parts of one key are concatenated along the channel axis.
"""
from math import inf

import torch
import torch.nn as nn

from .. import _native as N
from ..g_lib import GTau, GVec


def cg_product_tau(tau1, tau2, maxdim=inf):
    """Multiplicities of the CG product of two G vectors (lgn/cg_lib/cg_ops_tau.py:6-44), in its key order."""
    tau1, tau2 = GTau(tau1), GTau(tau2)
    tau = {}
    for (k1, n1) in tau1.keys():
        for (k2, n2) in tau2.keys():
            if max(k1, n1, k2, n2) >= maxdim:
                continue
            for k in range(abs(k1 - k2), min(k1 + k2, maxdim - 1) + 1, 2):
                for n in range(abs(n1 - n2), min(n1 + n2, maxdim - 1) + 1, 2):
                    tau[(k, n)] = tau.get((k, n), 0) + tau1[(k1, n1)] * tau2[(k2, n2)]
    return GTau(tau)


_TABLES = {}


def _pair_table(cg_dict, r1, r2, keys, device):
    """CSR form of the stacked (transposed-convention) CG matrix of the pair, rows = the concatenated irreps of `keys`."""
    ck = (id(cg_dict), r1, r2, tuple(keys), str(device))
    tab = _TABLES.get(ck)
    if tab is None:
        mat = torch.cat([cg_dict[(r1, r2)][key].detach().to("cpu", torch.float64) for key in keys], -2)      # [DO][d1 * d2]
        row_ptr, col, coef = [0], [], []
        for o in range(mat.shape[0]):
            nz = torch.nonzero(mat[o]).flatten().tolist()
            col += nz
            coef += [float(mat[o, j]) for j in nz]
            row_ptr.append(len(col))
        tab = (torch.tensor(row_ptr, dtype=torch.int32, device=device), torch.tensor(col, dtype=torch.int32, device=device),
               torch.tensor(coef, dtype=torch.float64, device=device), mat.shape[0], len(col), cg_dict)      # (keeps cg_dict alive: id())
        _TABLES[ck] = tab
    return tab


class _PairProduct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tab, mode, x1, x2):
        row_ptr, col, coef, DO, nnz, _ = tab
        x1, x2 = N.f64(x1), N.f64(x2)                       # fp64 only (the reference's precision): any other dtype raises, nothing is cast
        node = x2 if mode == 1 else x1                       # the node-like operand fixes rows and channels
        C_, D1, D2 = node.shape[-2], x1.shape[-1], x2.shape[-1]
        R = node[0].numel() // (C_ * node.shape[-1])
        n = node.shape[-3] if mode else 1
        out = torch.empty((2,) + tuple(node.shape[1:-1]) + (DO,), device=node.device, dtype=node.dtype)
        N._check(N.lib().lgn_cg_product_fwd_f64(R, n, C_, D1, D2, DO, mode, nnz, N.ptr(row_ptr), N.ptr(col), N.ptr(coef), N.ptr(x1), N.ptr(x2),
                                                N.ptr(out), N.stream_ptr()), "lgn_cg_product_fwd_f64")
        ctx.tab, ctx.mode, ctx.dims = tab, mode, (R, n, C_, D1, D2)
        ctx.save_for_backward(x1, x2)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x1, x2 = ctx.saved_tensors
        row_ptr, col, coef, DO, nnz, _ = ctx.tab
        R, n, C_, D1, D2 = ctx.dims
        # only the gradients autograd asks for (a null pointer tells the library to skip that operand: lgn_cg_product_bwd_f64)
        g1 = torch.zeros_like(x1) if ctx.needs_input_grad[2] else None
        g2 = torch.zeros_like(x2) if ctx.needs_input_grad[3] else None
        if g1 is None and g2 is None:
            return None, None, None, None
        N._check(N.lib().lgn_cg_product_bwd_f64(R, n, C_, D1, D2, DO, ctx.mode, nnz, N.ptr(row_ptr), N.ptr(col), N.ptr(coef), N.ptr(x1), N.ptr(x2),
                                                N.ptr(N.f64(g_out)), N.ptr(g1) if g1 is not None else None,
                                                N.ptr(g2) if g2 is not None else None, N.stream_ptr()), "lgn_cg_product_bwd_f64")
        return None, None, g1, g2


def cg_product(cg_dict, rep1, rep2, maxdim=inf, aggregate=False, ignore_check=False):
    """The Clebsch-Gordan product of two G vectors (lgn/cg_lib/cg_ops.py:135-218)."""
    tau1 = GTau({key: int(part.shape[-2]) for key, part in rep1.items()})
    tau2 = GTau({key: int(part.shape[-2]) for key, part in rep2.items()})
    assert tau1.channels and (tau1.channels == tau2.channels), f"The number of fragments must be same for each part! {tau1} {tau2}"
    maxk1, maxn1 = max(k for k, _ in rep1.keys()), max(n for _, n in rep1.keys())
    maxk2, maxn2 = max(k for k, _ in rep2.keys()), max(n for _, n in rep2.keys())
    max_dim = min(max(maxk1 + maxk2, maxn1 + maxn2) + 1, maxdim)
    if (cg_dict.maxdim < max_dim) or (cg_dict.maxdim < max(maxk1, maxn1, maxk2, maxn2)):
        raise ValueError(f"CG Dictionary maxdim ({cg_dict.maxdim}) not sufficiently large for (maxdim, L1, L2) = ({maxdim} {maxk1} {maxk2})")
    assert cg_dict.transpose, "This operation uses transposed CG coefficients!"

    new_rep = {}
    for (k1, n1), part1 in rep1.items():
        for (k2, n2), part2 in rep2.items():
            if max(k1, n1, k2, n2) > max_dim - 1 or part1.shape[-2] == 0 or part2.shape[-2] == 0:
                continue
            keys = [(k, n) for k in range(abs(k1 - k2), min(maxdim, k1 + k2 + 1), 2) for n in range(abs(n1 - n2), min(maxdim, n1 + n2 + 1), 2)]
            if part1.device.type != "cuda":
                raise RuntimeError("lgn (MI355X build): cg_product runs only in the HIP kernels of liblgn_amd.so on a GPU device; got tensors on "
                                   f"'{part1.device}'. There is no CPU fallback.")
            mode = 0
            b1, b2 = part1.shape[1:-2], part2.shape[1:-2]
            if not aggregate:
                assert b1 == b2, f"Batch sizes must be equal! {b1} {b2}"
            elif len(b1) == 3 and len(b2) == 2:
                assert b1[0] == b2[0], f"Batch sizes must be equal! {b1} {b2}"
                assert b1[2] == b2[1], f"Neighborhood sizes must be equal! {b1} {b2}"
                mode = 1
            elif len(b1) == 2 and len(b2) == 3:
                assert b2[0] == b1[0], f"Batch sizes must be equal! {b1} {b2}"
                assert b2[2] == b1[1], f"Neighborhood sizes must be equal! {b1} {b2}"
                mode = 2
            else:
                raise ValueError(f"Batch size error! {b1} {b2}")
            assert part1.shape[-2] == part2.shape[-2], f"Number of channels must match! {part1.shape[-2]} {part2.shape[-2]}"
            tab = _pair_table(cg_dict, (k1, n1), (k2, n2), keys, part1.device)
            prod = _PairProduct.apply(tab, mode, part1.contiguous(), part2.contiguous())
            for key, piece in zip(keys, torch.split(prod, [(k + 1) * (n + 1) for k, n in keys], dim=-1)):
                new_rep.setdefault(key, []).append(piece)
    return GVec({key: torch.cat(val, dim=-2) for key, val in new_rep.items()}, ignore_check=ignore_check)


class CGProduct(nn.Module):
    """Module form of the product (reference: lgn/cg_lib/cg_ops.py:10-133): optional fixed input multiplicities ``tau1`` / ``tau2``
    (checked against the operands at call time), the output multiplicities as ``tau_out`` (alias ``tau``), ``forward(rep1, rep2)``.
    ``maxdim`` defaults to the dictionary's, else to the length of the longer tau, as in the reference."""

    def __init__(self, tau1=None, tau2=None, aggregate=False, maxdim=inf, cg_dict=None, dtype=None, device=None):
        super().__init__()
        from . import CGDict
        if maxdim == inf:
            if cg_dict:
                maxdim = cg_dict.maxdim
            elif tau1 and tau2:
                maxdim = max(len(tau1), len(tau2))
            else:
                raise ValueError("maxdim is not defined, and was unable to retrieve get maxdim from cg_dict or tau1 and tau2")
        self.aggregate, self.maxdim = aggregate, maxdim
        self.device = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.dtype = dtype if dtype is not None else torch.float64
        self.cg_dict = cg_dict if cg_dict else CGDict(maxdim=maxdim, dtype=self.dtype, device=self.device)
        self.set_taus(tau1, tau2)

    def set_taus(self, tau1=None, tau2=None):
        self.tau1 = GTau(tau1) if tau1 else None
        self.tau2 = GTau(tau2) if tau2 else None
        if self.tau1 and self.tau2 and (not self.tau1.channels or self.tau1.channels != self.tau2.channels):
            raise ValueError(f"The number of fragments must be same for each part! {self.tau1} {self.tau2}")

    @property
    def tau_out(self):
        if not (self.tau1 and self.tau2):
            raise ValueError("Module not intialized with input type!")
        present = lambda tau: {key: int(mult > 0) for key, mult in tau.items()}          # noqa: E731
        channels = next(mult for tau in (self.tau1, self.tau2) for mult in tau.values() if mult > 0)
        return {key: channels * t for key, t in cg_product_tau(present(self.tau1), present(self.tau2), maxdim=self.maxdim).items()}

    tau = tau_out

    def forward(self, rep1, rep2):
        for want, rep, which in ((self.tau1, rep1, 1), (self.tau2, rep2, 2)):
            if want and want != rep.tau:
                raise ValueError(f"Input rep{which} does not match predefined tau!")
        return cg_product(self.cg_dict, rep1, rep2, maxdim=self.maxdim, aggregate=self.aggregate)
