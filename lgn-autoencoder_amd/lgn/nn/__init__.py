"""
Parameter containers with the reference's module tree and state_dict names
(lgn/nn/g_nn.py MixReps / CatMixReps, lgn/nn/position_levels.py RadPolyTrig / RadialFilters,
lgn/models/lgn_levels.py LGNNodeLevel / CGMLP, lgn/models/lgn_cg.py LGNCG).  They hold and initialise
parameters exactly like the reference (same shapes, names, init rules and RNG consumption order, so
that the same seed gives the same weights as the reference on CPU); the arithmetic is done by the
native kernels driven from lgn.models.
"""
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn

from ..g_lib import GTau
from ..plan import LevelPlan, param_key_order

Irrep = Tuple[int, int]


class MixReps(nn.Module):
    """Complex per-irrep mixing weights (2, C_out, C_in), registered as ``weights.(k, n)``.
    Init (lgn/nn/g_nn.py:59-93): randn * gain / max(2, C_out, C_in) / 10^k for irreps (k,k)."""

    def __init__(self, tau_in, tau_out, real=False, weight_init="randn", gain=1, device=None, dtype=torch.float64):
        super().__init__()
        tau_in = {k: v for k, v in GTau(tau_in).items() if v}
        if isinstance(tau_out, int):
            tau_out = {k: tau_out for k in tau_in}
        tau_out = dict(GTau(tau_out).items())
        if not set(tau_out) <= set(tau_in):
            raise AssertionError(f"Tau ({tau_out.keys()}) after mixing can't include more irreps than before ({tau_in.keys()})!")
        if weight_init not in ("randn", "rand"):
            raise NotImplementedError(f"weight_init can only be 'randn' or 'rand'; other choices are not implemented yet ({weight_init})!")
        self.tau_in, self.tau_out = GTau(tau_in), GTau(tau_out)
        self.real, self.weight_init = real, weight_init
        drawn = {}
        for key in tau_out:                                    # RNG consumption order = tau_out order (g_weight.py:95-109)
            shape = (2, tau_out[key], tau_in[key])
            w = (torch.randn if weight_init == "randn" else torch.rand)(shape, dtype=dtype)   # CPU RNG, then moved
            g = gain / max(shape) / (10 ** key[0] if key[0] == key[1] else 1)
            drawn[key] = (w * g).to(device)
        # registration order = sorted keys: ParameterDict.update() sorts a plain dict (g_weight.py:76-80)
        self.weights = nn.ParameterDict()
        for key in sorted(drawn):
            self.weights[str(key)] = nn.Parameter(drawn[key])
        self.out_order: List[Irrep] = param_key_order(sorted(drawn))

    def weight(self, key: Irrep) -> torch.Tensor:
        return self.weights[str(key)]

    @property
    def tau(self):
        return self.tau_out


class CatMixReps(nn.Module):
    """Holder reproducing ``cat_mix.mix_reps.weights.*`` (lgn/nn/g_nn.py:196-282)."""

    def __init__(self, tau_cat, tau_out, weight_init="randn", gain=1, device=None, dtype=torch.float64):
        super().__init__()
        self.mix_reps = MixReps(tau_cat, tau_out, weight_init=weight_init, gain=gain, device=device, dtype=dtype)
        self.taus_out = self.mix_reps.tau


class LGNNodeLevel(nn.Module):
    def __init__(self, plan: LevelPlan, level_gain, weight_init, device=None, dtype=torch.float64):
        super().__init__()
        self.plan = plan
        self.cat_mix = CatMixReps(plan.tau_cat, plan.tau_out, weight_init=weight_init, gain=level_gain,
                                  device=device, dtype=dtype)
        self.tau_out = self.cat_mix.taus_out


class CGMLP(nn.Module):
    """``linear.{i}`` = nn.Linear stack of lgn/models/lgn_levels.py:147-189 (default nn.Linear init)."""

    def __init__(self, num_channels, num_hidden=3, layer_width_mul=2, activation="sigmoid", device=None, dtype=torch.float64):
        super().__init__()
        from .. import _native
        self.activation = activation
        self.act_id = _native.activation_id(activation)      # get_activation_fn's names (lgn/nn/generic_levels.py:119-135)
        ns = 2 * num_channels
        width = layer_width_mul * ns
        if num_hidden > 0 and not ns <= width <= self.KERNEL_MAX_WIDTH:
            # (the reference takes any width, lgn/models/lgn_levels.py:124-189; the kernels hold a layer's weight image in LDS)
            raise NotImplementedError(f"the native CGMLP kernels take hidden widths 2C .. {self.KERNEL_MAX_WIDTH} (mlp_width x 2C); got "
                                      f"mlp_width={layer_width_mul} x {ns} scalars = {width}")
        self.num_scalars, self.width, self.num_hidden = ns, width, num_hidden
        self.linear = nn.ModuleList()
        self.linear.append(nn.Linear(ns, width))
        for _ in range(num_hidden - 1):
            self.linear.append(nn.Linear(width, width))
        self.linear.append(nn.Linear(width, ns) if num_hidden > 0 else nn.Linear(ns, ns))
        self.to(device=device, dtype=dtype)

    KERNEL_MAX_WIDTH = 96      # csrc/mlp.hip: mlp_dispatch -- 2C <= H <= 96 (mlp_mfma.hip up to 48, mlp_mfma_wide.hip beyond)

    def flat_params(self):
        out = []
        for lin in self.linear:
            out += [lin.weight, lin.bias]
        return out


class LGNCG(nn.Module):
    """``node_levels`` / ``mlp_levels`` module lists (lgn/models/lgn_cg.py:79-110): construction interleaves
    node level and MLP per layer (RNG order), registration groups them (state_dict order)."""

    def __init__(self, plans: Sequence[LevelPlan], level_gain, weight_init, mlp, mlp_depth, mlp_width, activation,
                 device=None, dtype=torch.float64):
        super().__init__()
        node_levels, mlp_levels = nn.ModuleList(), nn.ModuleList()
        for lvl, plan in enumerate(plans):
            node_levels.append(LGNNodeLevel(plan, level_gain[lvl], weight_init, device=device, dtype=dtype))
            if mlp:
                mlp_levels.append(CGMLP(plan.tau_out[(0, 0)], num_hidden=mlp_depth, layer_width_mul=mlp_width,
                                        activation=activation, device=device, dtype=dtype))
        self.node_levels = node_levels
        self.mlp = mlp
        if mlp:
            self.mlp_levels = mlp_levels
        self.tau_levels_node = [GTau(plans[0].tau_in)] + [GTau(p.tau_out) for p in plans]


class RadPolyTrig(nn.Module):
    """a, b, c (1,1,1,2*num_basis_fn) and ``linear.{l}`` of lgn/nn/position_levels.py:60-106."""

    def __init__(self, max_zf, num_basis_fn, num_channels, mix=True, input_basis="cartesian", device=None, dtype=torch.float64):
        super().__init__()
        if input_basis.lower() not in ("cartesian", "canonical"):
            raise ValueError("Input basis can only be 'cartesian' or 'canonical'!")
        if not (mix is True or mix == "cplx"):
            raise NotImplementedError("the native radial network implements mix='cplx' (the only mode the autoencoder uses)")
        if num_basis_fn < 1:
            raise ValueError(f"num_basis_fn must be at least 1; got {num_basis_fn}")
        self.max_zf, self.num_basis_fn, self.num_channels, self.input_basis = max_zf, num_basis_fn, num_channels, input_basis
        nb = 2 * num_basis_fn
        # drawn in the default dtype on the CPU, then cast/moved, like the reference (position_levels.py:67-73)
        self.a = nn.Parameter(torch.randn(1, 1, 1, nb).to(device=device, dtype=dtype))
        self.b = nn.Parameter(torch.randn(1, 1, 1, nb).to(device=device, dtype=dtype))
        self.c = nn.Parameter(torch.randn(1, 1, 1, nb).to(device=device, dtype=dtype))
        out = num_channels if input_basis == "canonical" else 2 * num_channels
        self.linear = nn.ModuleList([nn.Linear(nb, out).to(device=device, dtype=dtype) for _ in range(max_zf + 1)])
        self.radial_types = (num_channels,) * max_zf
        # The kernels read KERNEL_BELLS bells per radial network.  Fewer (num_basis_fn < 10): a network's flat parameter block STORES
        # these tensors KERNEL_BELLS wide, zero padded (lgn/models/common.py: CGModule._flatten_parameters), so that every native call --
        # whole step included -- reads them in place.  More (num_basis_fn > 10, round 6): stored as whole groups of KERNEL_BELLS; the
        # Linear layer is a sum over bells, so a level's moments are the sum of the moments of its groups (the bias in the first
        # one) -- lgn/ops.py: GenericLevelFn runs the moments kernels once per group, on the per-operator path.
        self.kernel_width = -(-nb // self.KERNEL_BELLS) * self.KERNEL_BELLS
        if nb != self.kernel_width:
            self._kernel_pad = {"a": self.kernel_width, "b": self.kernel_width, "c": self.kernel_width}
            for lin in self.linear:
                lin._kernel_pad = {"weight": self.kernel_width}

    KERNEL_BELLS = 20          # csrc/common.hpp: NB -- the width the level / moments kernels read (2 * num_basis_fn of the default 10)

    def flat_params(self):
        return [self.a, self.b, self.c, self.linear[0].weight, self.linear[0].bias, self.linear[1].weight, self.linear[1].bias]

    def kernel_params(self):
        """flat_params() in the width the kernels read: whole groups of 20 bells (``kernel_width``).  Other counts (num_basis_fn not a
        multiple of 10, lgn/nn/position_levels.py:44-64) are embedded by zero padding: a bell with a = b = c = 0 and zero Linear weights evaluates to 0 and feeds nothing.  Inside a network
        the padded blocks are the parameters' STORAGE (``a_store`` ...: views of the flat block, autograd-tracked on the per-operator
        path); a stand-alone module pads on the fly (a differentiable op: the padding's gradient slots are sliced away)."""
        if 2 * self.num_basis_fn == self.kernel_width:
            return self.flat_params()
        if "a_store" in self.__dict__:
            return [self.a_store, self.b_store, self.c_store, self.linear[0].weight_store, self.linear[0].bias,
                    self.linear[1].weight_store, self.linear[1].bias]
        pad = self.kernel_width - 2 * self.num_basis_fn
        a, b, c, w0, b0, w1, b1 = self.flat_params()
        P = torch.nn.functional.pad
        return [P(a, (0, pad)), P(b, (0, pad)), P(c, (0, pad)), P(w0, (0, pad)), b0, P(w1, (0, pad)), b1]


class RadialFilters(nn.Module):
    def __init__(self, max_zf, num_basis_fn, num_channels_out, num_levels, mix=True, input_basis="cartesian",
                 device=None, dtype=torch.float64):
        super().__init__()
        self.num_levels, self.max_zf = num_levels, max_zf
        self.rad_funcs = nn.ModuleList([
            RadPolyTrig(max_zf[l], num_basis_fn, num_channels_out[l], mix=mix, input_basis=input_basis, device=device, dtype=dtype)
            for l in range(num_levels)])
        self.tau = [{(l, l): rf.radial_types[l - 1] for l in range(0, mz + 1)} for rf, mz in zip(self.rad_funcs, max_zf)]
        self.num_rad_channels = self.tau[0][(1, 1)] if self.tau else 0


__all__ = ["MixReps", "CatMixReps", "LGNNodeLevel", "CGMLP", "LGNCG", "RadPolyTrig", "RadialFilters"]
