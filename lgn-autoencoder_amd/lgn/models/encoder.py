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
        if jet_features:            # one more node (the jet's momentum) and one more input scalar (lgn_encoder.py:159-161)
            num_input_particles += 1
            tau_input_scalars += 1
        if num_cg_levels < 1 or any(m not in (2, 3) for m in maxdim) or any(z != 1 for z in max_zf):
            raise NotImplementedError(
                f"this build implements maxdim 2 (fused kernels) and 3 (table-driven kernels) with max_zf=1; got maxdim={maxdim}, max_zf={max_zf}")
        if tau_input_scalars < 1 or tau_input_vectors != 1:
            raise NotImplementedError("the encoder input is the mass (+ extra scalars) and ONE vector (p4) per particle")
        self.tau_input_scalars = tau_input_scalars
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
        node_ps, node_mask, scalars = self._prepare_input(data)
        if not covariance_test and self.use_fused and self._fused_ok():
            # the whole encoder is one native call (and one more for its backward): csrc/step.hip lgn_encoder_fwd/bwd_f64
            lat_s, lat_v = ops.EncoderFn.apply(self, node_ps, node_mask, self.flat_params, scalars)
            return GVec({(0, 0): lat_s, (1, 1): lat_v})
        # module / autograd path: one native call per operator (all irreps, every map_to_latent, internal features)
        self._bind(self._tracked_views())
        try:
            return self._forward_modular(node_ps, node_mask, covariance_test, scalars)
        finally:
            self._bind(self._p_stores)

    def _fused_ok(self) -> bool:
        """True when a whole-network native implementation covers this configuration (lgn/ops.py: native_kind)."""
        # (jet features / extra input scalars included: lgn_net_desc.n_in_scalars, round 4)
        # (and its per-jet end stages fit a CU's LDS: plan-time query, lgn/_native.py: end_stages_fit)
        if ops.native_kind(self) is None or self.tau_input_scalars > 8:
            return False
        fit = self.__dict__.get("_end_fit")
        if fit is None:
            fit = self.__dict__["_end_fit"] = ops.N.end_stages_fit(encoder=self)
        return fit

    def _forward_modular(self, node_ps, node_mask, covariance_test, scalars=None):
        # input features: (0,0) = (sqrt|p^2| [, jet mass, extra scalars], 0), (1,1) = canonical(p)   (lgn_encoder.py:287-293,376)
        mass = ops.normsq4(node_ps).abs().sqrt()
        if scalars is None:
            s0 = torch.stack([mass, torch.zeros_like(mass)], 0).unsqueeze(-1).unsqueeze(-1)  # (2,B,N,1,1)
        else:
            sc = torch.cat([mass.unsqueeze(-1), scalars], -1)                                  # (B,N,S)
            s0 = torch.stack([sc, torch.zeros_like(sc)], 0).unsqueeze(-1).contiguous()         # (2,B,N,S,1)
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
        """lgn_encoder.py:338-412.  Returns (momenta, node mask, extra scalars or None): with jet_features the jet's momentum is
        appended as one more (unmasked) node and every node gets the scalar normsq4(sum over ALL nodes) -- the reference sums after
        appending the jet node and takes no square root, i.e. 4 x the squared jet mass (lgn_encoder.py:377-390) -- as its second
        input scalar; data['scalars'] (B, N, k) are appended after it (lgn_encoder.py:403-408)."""
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
        scalars = None
        if self.jet_features:
            node_ps = torch.cat((node_ps, node_ps.sum(dim=-2, keepdim=True)), dim=-2)
            jet_mass = ops.normsq4(node_ps.sum(dim=-2))                                        # (B,)
            scalars = jet_mass.unsqueeze(-1).unsqueeze(-1).repeat(1, node_ps.shape[-2], 1)
            node_mask = torch.cat((node_mask, torch.ones_like(node_mask[..., 0:1])), dim=-1)
        if "scalars" in data:
            extra = data["scalars"].to(device=self.device, dtype=self.dtype)
            scalars = extra if scalars is None else torch.cat([scalars, extra], dim=-1)
        have = 1 + (0 if scalars is None else scalars.shape[-1])
        if have != self.tau_input_scalars:
            raise ValueError(f"the encoder was built for {self.tau_input_scalars} input scalars per particle, the batch gives {have}")
        return node_ps.contiguous(), node_mask.contiguous(), (None if scalars is None else scalars.contiguous())
