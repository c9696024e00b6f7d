"""Pieces shared by LGNEncoder and LGNDecoder: CGModule-like base and the level stack driver."""
from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..cg_lib import CGDict
from ..g_lib import GVec
from ..plan import LevelPlan, check_maxdim2_layout


def adapt_var_list(var, num_cg_levels):
    """lgn/models/utils.py:4-42, including its over-long-list truncation quirk."""
    if type(var) == list:
        if len(var) < num_cg_levels:
            return var + (num_cg_levels - len(var)) * [var[-1]]
        if len(var) == num_cg_levels:
            return var
        return var[: num_cg_levels - 1]
    if type(var) in (float, int):
        return [var] * num_cg_levels
    raise ValueError(f"Incorrect type of variables: {type(var)}. The allowed data types are list, float, or int")


class CGModule(nn.Module):
    """Device / dtype / cg_dict plumbing of the reference's CGModule (lgn/cg_lib/cg_module.py:7-210)."""

    def __init__(self, maxdim, device=None, dtype=None, cg_dict=None):
        super().__init__()
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if dtype is None:
            dtype = torch.float64
        if dtype not in (torch.float64, torch.double):
            # the reference itself only works in fp64 (cg_module.py:62-73, zonal_functions.py:441-446)
            raise ValueError(f"the native LGN path is fp64, like the reference; got dtype {dtype}")
        self._device, self._dtype, self._maxdim = torch.device(device), dtype, maxdim
        if cg_dict is None:
            cg_dict = CGDict(maxdim=maxdim, transpose=True, device=self._device, dtype=dtype)
        elif cg_dict.maxdim is None or cg_dict.maxdim < maxdim:
            cg_dict.update_maxdim(maxdim)
        self._cg_dict = cg_dict

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    @property
    def maxdim(self):
        return self._maxdim

    @property
    def cg_dict(self):
        return self._cg_dict

    def l1_norm(self) -> torch.Tensor:
        return sum(p.abs().sum() for p in self.parameters())

    def l2_norm(self) -> torch.Tensor:
        return sum(torch.pow(p, 2).sum() for p in self.parameters())

    def _require_gpu(self):
        if self._device.type != "cuda":
            raise RuntimeError(
                "lgn (MI355X build): the LGN hot path runs only in the HIP kernels of liblgn_amd.so on a GPU device; "
                f"this module was created on '{self._device}'. There is no CPU fallback.")


def _is_fused_layout(plan: LevelPlan, maxdim: int) -> bool:
    if maxdim != 2:
        return False
    try:
        check_maxdim2_layout(plan)
        return True
    except RuntimeError:
        return False


def run_levels(module, decoder: bool, feats, p, mask):
    """LGNCG.forward (lgn/models/lgn_cg.py:124-180): per level one native level call (edge network + CG aggregate +
    CG power + CatMix) followed by the native CGMLP on the scalars.  ``feats`` = {irrep: (2,B,N,C,d)} in the
    level's GVec order.  Levels whose layout is the maxdim=2 one use the fused closed-form kernels
    (csrc/level_*.hip); any other irrep content goes through the table-driven generic kernels
    (csrc/generic_*.hip).  Returns the list of feature dicts after every level (input first)."""
    lgn_cg, rad_funcs, plans = module.lgn_cg, module.rad_funcs, module.plans
    out = [feats]
    for lvl, plan in enumerate(plans):
        mix = lgn_cg.node_levels[lvl].cat_mix.mix_reps
        radp = rad_funcs.rad_funcs[lvl].flat_params()
        if _is_fused_layout(plan, module.level_maxdim[lvl]):
            s, v = ops.LevelFn.apply(decoder, feats[(0, 0)].squeeze(-1), feats[(1, 1)], p, mask, *radp,
                                     mix.weight((0, 0)), mix.weight((1, 1)))
            new = {(0, 0): s.unsqueeze(-1), (1, 1): v}
        else:
            tables = module.level_tables(lvl)
            X = torch.cat([feats[r] for r in plan.node_order], dim=-1).contiguous()
            wmix = [mix.weight(r) for r in tables.meta["out_irreps"]]
            Y = ops.GenericLevelFn.apply(decoder, tables, plan.channels_out, X, p, mask, *radp, *wmix)
            parts = torch.split(Y, [(r[0] + 1) * (r[1] + 1) for r in tables.meta["out_irreps"]], dim=-1)
            new = dict(zip(tables.meta["out_irreps"], parts))
        if lgn_cg.mlp:
            s = ops.CGMLPFn.apply(new[(0, 0)].squeeze(-1).contiguous(), *lgn_cg.mlp_levels[lvl].flat_params())
            new[(0, 0)] = s.unsqueeze(-1)
        feats = {r: new[r] for r in plan.out_order}
        out.append(feats)
    return out


class LevelTablesMixin:
    """Lazily built device tables of the generic levels (one per level, cached)."""

    def level_tables(self, lvl: int):
        from .. import _native as N
        from ..plan import build_local_tables
        cache = self.__dict__.setdefault("_level_tables", {})
        if lvl not in cache:
            cache[lvl] = N.DeviceTables(build_local_tables(self.plans[lvl], self.cg_dict), self.device)
        return cache[lvl]


def as_gvec(feats, order):
    return GVec({k: feats[k] for k in order})
