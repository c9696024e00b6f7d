"""LGNEncoder -- same constructor / forward / attribute surface as the reference's
lgn/models/lgn_encoder.py:20-416, with the message passing executed by liblgn_amd.so."""
import logging
from typing import Dict, List, Tuple, Union

import numpy as np
import torch

from .. import ops
from ..g_lib import GTau, GVec
from ..nn import LGNCG, MixReps, RadialFilters
from ..plan import build_level_plans
from .common import CGModule, LevelTablesMixin, adapt_var_list, as_gvec, run_levels


class LGNEncoder(CGModule, LevelTablesMixin):
    def __init__(self, num_input_particles: int, tau_input_scalars: int, tau_input_vectors: int,
                 tau_latent_scalars: int, tau_latent_vectors: int, maxdim, num_basis_fn: int, num_channels: List[int],
                 max_zf, weight_init, level_gain, activation: str = "leakyrelu", mlp: bool = True,
                 mlp_depth: int = None, mlp_width: int = None, scale: float = 1.0, jet_features: bool = False,
                 map_to_latent: str = "mean", device: torch.device = None, dtype: torch.dtype = None, cg_dict=None):
        num_cg_levels = len(num_channels) - 1
        level_gain = adapt_var_list(level_gain, num_cg_levels)
        maxdim = adapt_var_list(maxdim, num_cg_levels)
        max_zf = adapt_var_list(max_zf, num_cg_levels)
        super().__init__(maxdim=max(maxdim + max_zf), device=device, dtype=dtype, cg_dict=cg_dict)
        logging.info(f"Initializing encoder with device: {self.device} and dtype: {self.dtype}")
        if jet_features:
            raise NotImplementedError("jet_features=True is outside the accelerated path (SURVEY section 8)")
        if num_cg_levels < 1 or any(m not in (2, 3) for m in maxdim) or any(z != 1 for z in max_zf):
            raise NotImplementedError(
                f"this build implements maxdim 2 (fused kernels) and 3 (table-driven kernels) with max_zf=1; got maxdim={maxdim}, max_zf={max_zf}")
        if tau_input_scalars != 1 or tau_input_vectors != 1:
            raise NotImplementedError("the encoder input is one scalar (mass) and one vector (p4) per particle")
        misc = {"device": self.device, "dtype": self.dtype}

        self.num_input_particles = num_input_particles
        self.input_basis = "cartesian"
        self.num_cg_levels = num_cg_levels
        self.num_basis_fn = num_basis_fn
        self.max_zf = max_zf
        self.level_maxdim = maxdim
        self.num_channels = num_channels
        self.jet_features = jet_features
        self.map_to_latent = map_to_latent
        self.mlp, self.mlp_depth, self.mlp_width = mlp, mlp_depth, mlp_width
        self.activation = activation
        self.scale = scale

        # construction order == the reference's (RNG stream): radial filters, input mixing, CG levels, latent mixing
        self.rad_funcs = RadialFilters(max_zf=max_zf, num_basis_fn=num_basis_fn, num_channels_out=num_channels,
                                       num_levels=num_cg_levels, **misc)
        tau_in = GTau({(0, 0): tau_input_scalars, (1, 1): tau_input_vectors})
        self.tau_dict = {"input": tau_in}
        tau0 = {(l, l): num_channels[0] for l in range(max_zf[0] + 1)}
        self.input_func_node = MixReps(tau_in, tau0, **misc)
        self.plans = build_level_plans(num_channels, maxdim, max_zf, mlp, tau0, self.input_func_node.out_order)
        self.lgn_cg = LGNCG(self.plans, level_gain, weight_init, mlp, mlp_depth, mlp_width, activation, **misc)
        self.tau_cg_levels_node = self.lgn_cg.tau_levels_node
        self.tau_dict["cg_layers"] = self.tau_cg_levels_node.copy()

        tau_last = dict(self.plans[-1].tau_out)
        if map_to_latent.lower() == "mix":     # learned mixing over the particle axis as well (lgn_encoder.py:226-232)
            tau_last = {w: int(v * num_input_particles) for w, v in tau_last.items()}
        self.tau_output = {w: 1 for w in tau_last}
        self.tau_output[(0, 0)] = tau_latent_scalars
        self.tau_output[(1, 1)] = tau_latent_vectors
        self.tau_dict["latent"] = self.tau_output
        self.mix_reps = MixReps(tau_last, self.tau_output, **misc)
        self.tau_latent = self.tau_output
        self.__num_param = sum(p.nelement() for p in self.parameters() if p.requires_grad)
        self.use_fused = True      # False: force the per-operator module/autograd path (cross-checks, tests)
        self._flatten_parameters()

    @property
    def num_learnable_parameters(self) -> int:
        return self.__num_param

    def forward(self, data: Union[Dict[str, torch.Tensor], torch.Tensor, np.ndarray], covariance_test: bool = False
                ) -> Union[GVec, Tuple[GVec, List[GVec]]]:
        self._require_gpu()
        self._check_views()
        node_ps, node_mask = self._prepare_input(data)
        if not covariance_test and self.use_fused and self._fused_ok():
            # the whole encoder is one native call (and one more for its backward): csrc/step.hip lgn_encoder_fwd/bwd_f64
            lat_s, lat_v = ops.EncoderFn.apply(self, node_ps, node_mask, self.flat_params)
            return GVec({(0, 0): lat_s, (1, 1): lat_v})
        # module / autograd path: one native call per operator (all irreps, every map_to_latent, internal features)
        self._bind(self._tracked_views())
        try:
            return self._forward_modular(node_ps, node_mask, covariance_test)
        finally:
            self._bind(self._p_views)

    def _fused_ok(self) -> bool:
        """True when a whole-network native implementation covers this configuration (lgn/ops.py: native_kind)."""
        return ops.native_kind(self) is not None

    def _forward_modular(self, node_ps, node_mask, covariance_test):
        # input features: (0,0) = (sqrt|p^2|, 0), (1,1) = canonical(p)   (lgn_encoder.py:287-293,376)
        mass = ops.normsq4(node_ps).abs().sqrt()
        s0 = torch.stack([mass, torch.zeros_like(mass)], 0).unsqueeze(-1).unsqueeze(-1)      # (2,B,N,1,1)
        v0 = ops.cart_to_canonical_real(node_ps).unsqueeze(-2)                                 # (2,B,N,1,4)
        s = ops.MixFn.apply(self.input_func_node.weight((0, 0)), s0).squeeze(-1)
        v = ops.MixFn.apply(self.input_func_node.weight((1, 1)), v0)

        order0 = self.input_func_node.out_order
        f0 = {(0, 0): s.unsqueeze(-1), (1, 1): v}
        feats = run_levels(self, False, {r: f0[r] for r in order0}, node_ps, node_mask)

        # mix_reps acts on every irrep of the last level; only (0,0) and (1,1) are kept (lgn_encoder.py:322-325)
        last = feats[-1]
        if self.map_to_latent.lower() == "mix":    # (2,B,N,C,d) -> (2,B,1,N*C,d)  (lgn_encoder.py:313-319)
            last = {k: v.reshape(2, v.shape[1], 1, -1, v.shape[-1]) for k, v in last.items()}
        lat = {(0, 0): ops.MixFn.apply(self.mix_reps.weight((0, 0)), last[(0, 0)].contiguous()),
               (1, 1): ops.canonical_to_cart(ops.MixFn.apply(self.mix_reps.weight((1, 1)), last[(1, 1)].contiguous()))}
        latent = GVec(ops.aggregate_latent(self.map_to_latent, lat))
        if not covariance_test:
            return latent
        orders = [order0] + [p.out_order for p in self.plans]
        return latent, [as_gvec(f, o) for f, o in zip(feats, orders)]

    def _prepare_input(self, data):
        """lgn_encoder.py:338-412 (without the jet-feature node)."""
        if isinstance(data, torch.Tensor):
            data = {"p4": data}
        elif isinstance(data, np.ndarray):
            data = {"p4": torch.from_numpy(data)}
        node_ps = data["p4"].to(device=self.device, dtype=self.dtype)
        if self.scale != 1.0:
            node_ps = node_ps * self.scale
        for key in ("labels", "masks", "mask"):
            if key in data:
                node_mask = data[key].to(device=self.device, dtype=torch.uint8)
                break
        else:
            node_mask = (data["p4"][..., 0] != 0).to(device=self.device, dtype=torch.uint8)
        if "scalars" in data:
            raise NotImplementedError("extra input scalars are outside the accelerated path")
        return node_ps.contiguous(), node_mask.contiguous()
