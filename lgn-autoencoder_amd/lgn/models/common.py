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


class FlatParameter(nn.Parameter):
    """The flat parameter block of a network.  A *subclass* of nn.Parameter on purpose: torch.optim's default
    implementation choice (``_default_to_fused_or_foreach``) takes the multi-tensor ("foreach") kernels only for exact
    ``Tensor`` / ``Parameter`` types.  Those kernels give every 64 k-element chunk of a tensor to ONE workgroup, so an update
    of one 34 k-element block would run on a single CU (measured: 18-26 us per op, 8 ops per Adam step); the single-tensor
    path the optimiser falls back to for other types uses ordinary element-wise kernels over the whole chip (~3 us per op)."""


class CGModule(nn.Module):
    """Device / dtype / cg_dict plumbing of the reference's CGModule (lgn/cg_lib/cg_module.py:7-210), plus the
    MI355X-first parameter storage of the two networks:

    **one flat fp64 block per network.**  The sub-modules are constructed exactly like the reference's (same shapes,
    names, init rules and RNG order), then ``_flatten_parameters`` moves every tensor into ONE ``nn.Parameter``
    (``flat_params``) and leaves plain views of it in the sub-modules.  Consequences:
      * ``parameters()`` is a single leaf: an optimiser step is one element-wise pass, ``l1_norm()`` two kernels,
        autograd accumulates one gradient, data parallelism all-reduces one buffer, and the native kernels address
        every weight as ``flat + offset`` (include/lgn_amd.h: parameter slots);
      * ``state_dict()`` / ``load_state_dict()`` keep the reference's keys and order (checkpoints are interchangeable,
        utils/train.py:114-129,376-385): they enumerate the named views, not the flat block;
      * per-name access: ``named_parameter_views()`` and ``named_grads()``.
    """

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

    # ---- flat parameter block -------------------------------------------------------------------------
    def _flatten_parameters(self):
        named = list(self.named_parameters())            # registration order == the reference's state_dict order
        self._p_names = [n for n, _ in named]
        self._p_shapes = [tuple(p.shape) for _, p in named]
        self._p_slots = []
        # STORAGE shape of every parameter: the reference's shape, except where a sub-module asks for a wider last axis
        # (``_kernel_pad = {attr: width}``, lgn/nn: RadPolyTrig with num_basis_fn < 10 -- the kernels read 20 bells per radial
        # network in place).  The padding columns are zeros and stay zeros: a bell with a = b = c = 0 and zero Linear weights feeds
        # nothing and receives an exactly zero gradient, which Adam and the L1 term leave at zero.  Everything the reference can see
        # -- names, shapes, state_dict entries, named_grads -- is the narrow view [..., :n] of the stored block.
        self._p_store = []
        for name, p in named:
            path, _, attr = name.rpartition(".")
            owner = self.get_submodule(path) if path else self
            self._p_slots.append((owner, attr))
            width = getattr(owner, "_kernel_pad", {}).get(attr)
            shape = tuple(p.shape)
            self._p_store.append(shape[:-1] + (width,) if width and shape and shape[-1] < width else shape)
        self._p_sizes = [int(torch.Size(s).numel()) for s in self._p_store]
        self._p_offsets = [0]
        for n in self._p_sizes:
            self._p_offsets.append(self._p_offsets[-1] + n)
        pieces = []
        for (_, p), store in zip(named, self._p_store):
            t = p.detach()
            if store != tuple(p.shape):
                t = torch.nn.functional.pad(t, (0, store[-1] - p.shape[-1]))
            pieces.append(t.reshape(-1))
        flat = torch.cat(pieces).contiguous()
        for owner, attr in self._p_slots:
            del owner._parameters[attr]
        self.flat_params = FlatParameter(flat)
        self._rebind_views()

    @staticmethod
    def _narrow(stored, shape):
        return stored if tuple(stored.shape) == tuple(shape) else stored[..., :shape[-1]]

    def _rebind_views(self):
        """(Re)create the plain views the sub-modules and the native calls read; called whenever ``flat_params``
        moved (``.to()``, a trainer re-homing it into a joint buffer, deepcopy)."""
        base = self.flat_params.detach()
        self._p_stores = [base[o:o + n].view(s) for o, n, s in zip(self._p_offsets, self._p_sizes, self._p_store)]
        self._p_views = [self._narrow(t, s) for t, s in zip(self._p_stores, self._p_shapes)]
        self._views_ptr = base.data_ptr()
        self._device = base.device
        self._bind(self._p_stores)
        self.__dict__.pop("_native_cache", None)

    def _bind(self, stores):
        """``owner.attr`` = the reference-shaped tensor (the narrow view of its stored block); where the storage is wider,
        ``owner.attr_store`` = the stored block itself (what the kernels read).  stores: plain views (_p_stores) or the
        autograd-tracked ones (_tracked_views)."""
        for (owner, attr), t, shape in zip(self._p_slots, stores, self._p_shapes):
            owner.__dict__[attr] = self._narrow(t, shape)
            if tuple(t.shape) != tuple(shape):
                owner.__dict__[attr + "_store"] = t

    def _check_views(self):
        if self._views_ptr != self.flat_params.data_ptr():
            self._rebind_views()

    def _tracked_views(self):
        """Views of ``flat_params`` that autograd follows (module/autograd path): ONE split node + metadata-only views,
        so the backward assembles the flat gradient with a single concatenation."""
        parts = torch.split_with_sizes(self.flat_params, self._p_sizes)
        return [p.view(s) for p, s in zip(parts, self._p_store)]

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if "flat_params" in self._parameters:
            self._rebind_views()
        return out

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            # derived, rebuilt on demand: device-side tables and descriptors hold ctypes pointers (not copyable), and the
            # parameter views must point into the COPY's flat block
            if k in ("_native_cache", "_level_tables", "_p_views", "_p_stores", "_views_ptr"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        if "flat_params" in new._parameters:
            new._rebind_views()
        return new

    def named_parameter_views(self):
        """(reference parameter name, view of the flat block) in state_dict order."""
        self._check_views()
        return list(zip(self._p_names, self._p_views))

    def named_grads(self):
        """(reference parameter name, view of ``flat_params.grad`` or None) in state_dict order."""
        g = self.flat_params.grad
        if g is None:
            return [(n, None) for n in self._p_names]
        return [(n, self._narrow(g[o:o + k].view(st), s))
                for n, o, k, st, s in zip(self._p_names, self._p_offsets, self._p_sizes, self._p_store, self._p_shapes)]

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        from collections import OrderedDict
        if args:      # the deprecated positional form of nn.Module.state_dict: (destination, prefix, keep_vars)
            destination = args[0] if len(args) > 0 else destination
            prefix = args[1] if len(args) > 1 else prefix
            keep_vars = args[2] if len(args) > 2 else keep_vars
        out = destination if destination is not None else OrderedDict()
        for name, view in self.named_parameter_views():
            out[prefix + name] = view if keep_vars else view.detach()
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Called when the network is a CHILD of the module being loaded (an autoencoder wrapper, ...): nn.Module walks the
        tree with this hook and never reaches the ``load_state_dict`` override below.  Consumes the reference-named entries
        under ``prefix`` (and removes them from ``state_dict``, which is this sub-tree's private copy, so that the parameter-less
        sub-modules below do not report them as unexpected)."""
        views = dict(self.named_parameter_views())
        mine = [k for k in list(state_dict.keys()) if k.startswith(prefix)]
        for k in mine:
            name = k[len(prefix):]
            v = views.get(name)
            if v is None:
                if strict:
                    unexpected_keys.append(k)
            elif tuple(state_dict[k].shape) != tuple(v.shape):
                error_msgs.append(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)} from checkpoint, "
                                  f"the shape in current model is {tuple(v.shape)}.")
            else:
                with torch.no_grad():
                    v.copy_(state_dict[k])
            del state_dict[k]
        seen = {k[len(prefix):] for k in mine}
        missing_keys.extend(prefix + n for n in views if n not in seen)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        from torch.nn.modules.module import _IncompatibleKeys
        views = dict(self.named_parameter_views())
        missing = [k for k in views if k not in state_dict]
        unexpected = [k for k in state_dict if k not in views]
        errors = []
        for k, v in views.items():
            if k in state_dict and tuple(state_dict[k].shape) != tuple(v.shape):
                errors.append(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)} from checkpoint, "
                              f"the shape in current model is {tuple(v.shape)}.")
        if strict and (missing or unexpected):
            errors.insert(0, f"Missing key(s) in state_dict: {missing}. Unexpected key(s) in state_dict: {unexpected}.")
        if errors:
            raise RuntimeError(f"Error(s) in loading state_dict for {self.__class__.__name__}:\n\t" + "\n\t".join(errors))
        with torch.no_grad():
            for k, v in views.items():
                if k in state_dict:
                    v.copy_(state_dict[k])
        return _IncompatibleKeys(missing, unexpected)

    def zero_grad(self, set_to_none: bool = True):
        """nn.Module.zero_grad walks the whole module tree (~100 sub-modules, 0.1 ms of host time per call); every parameter
        of the network lives in ``flat_params``, so this is all there is to clear."""
        p = self._parameters.get("flat_params")
        if p is None:
            return super().zero_grad(set_to_none=set_to_none)
        if p.grad is not None:
            if set_to_none:
                p.grad = None
            else:
                p.grad.detach_().zero_()

    def l1_norm(self) -> torch.Tensor:
        return ops.L1Fn.apply(self.flat_params)

    def l2_norm(self) -> torch.Tensor:
        return torch.pow(self.flat_params, 2).sum()

    def _require_gpu(self):
        if self.flat_params.device.type != "cuda":
            raise RuntimeError(
                "lgn (MI355X build): the LGN hot path runs only in the HIP kernels of liblgn_amd.so on a GPU device; "
                f"this module lives on '{self.flat_params.device}'. There is no CPU fallback.")


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
        radp = rad_funcs.rad_funcs[lvl].kernel_params()       # (whole groups of the kernels' 20 bells, zero padded)
        # (more than one group -- num_basis_fn > 10 -- runs the table-driven kernels at any maxdim: their moments are summed over the
        # groups, ops.GenericLevelFn; the closed-form kernels evaluate their one group inside the pair sweep.  The decoder's
        # edges carry the Linear bias alone -- no bells, no groups)
        if _is_fused_layout(plan, module.level_maxdim[lvl]) and (decoder or radp[0].shape[-1] == ops.BELLS):
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
            s = ops.CGMLPFn.apply(lgn_cg.mlp_levels[lvl].act_id, new[(0, 0)].squeeze(-1).contiguous(), *lgn_cg.mlp_levels[lvl].flat_params())
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
