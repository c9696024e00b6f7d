"""LGNDecoder -- same constructor / forward / attribute surface as the reference's
lgn/models/lgn_decoder.py:16-349, with the message passing executed by liblgn_amd.so."""
import logging
from typing import List

import torch

from .. import ops
from ..g_lib import GTau, GVec
from ..nn import LGNCG, MixReps, RadialFilters
from ..plan import build_level_plans
from .common import CGModule, LevelTablesMixin, adapt_var_list, as_gvec, run_levels


class LGNDecoder(CGModule, LevelTablesMixin):
    def __init__(self, tau_latent_scalars: int, tau_latent_vectors: int, num_output_particles: int,
                 tau_output_scalars: int, tau_output_vectors: int, maxdim, num_basis_fn: int, num_channels: List[int],
                 max_zf, weight_init, level_gain, activation: str = "leakyrelu", mlp: bool = True,
                 mlp_depth: int = None, mlp_width: int = None, device: torch.device = None,
                 dtype: torch.dtype = None, cg_dict=None):
        num_cg_levels = len(num_channels) - 1
        level_gain = adapt_var_list(level_gain, num_cg_levels)
        maxdim = adapt_var_list(maxdim, num_cg_levels)
        max_zf = adapt_var_list(max_zf, num_cg_levels)
        super().__init__(maxdim=max(maxdim + max_zf), device=device, dtype=dtype, cg_dict=cg_dict)
        logging.info(f"Initializing decoder with device: {self.device} and dtype: {self.dtype}")
        if num_cg_levels < 1 or any(m not in (2, 3) for m in maxdim) or any(z != 1 for z in max_zf):
            raise NotImplementedError(
                f"this build implements maxdim 2 (fused kernels) and 3 (table-driven kernels) with max_zf=1; got maxdim={maxdim}, max_zf={max_zf}")
        misc = {"device": self.device, "dtype": self.dtype}

        self.input_basis = "canonical"
        self.tau_latent_scalars, self.tau_latent_vectors = tau_latent_scalars, tau_latent_vectors
        self.tau_dict = {"input": GTau({(0, 0): tau_latent_scalars, (1, 1): tau_latent_vectors})}
        self.num_output_particles = num_output_particles
        self.num_cg_levels = num_cg_levels
        self.num_basis_fn = num_basis_fn
        self.max_zf = max_zf
        self.level_maxdim = maxdim
        self.num_channels = num_channels
        self.mlp, self.mlp_depth, self.mlp_width = mlp, mlp_depth, mlp_width
        self.activation = activation

        # construction order == the reference's: latent_to_graph, input mixing, radial filters, CG levels, output
        tau_graph = GTau({(0, 0): num_output_particles, (1, 1): num_output_particles})
        self.latent_to_graph = MixReps(self.tau_dict["input"], tau_graph, **misc)
        tau0 = {(0, 0): num_channels[0], (1, 1): num_channels[0]}
        self.input_func_node = MixReps(GTau({(0, 0): 1, (1, 1): 1}), tau0, **misc)
        self.rad_funcs = RadialFilters(max_zf=max_zf, num_basis_fn=num_basis_fn, num_channels_out=num_channels,
                                       num_levels=num_cg_levels, input_basis=self.input_basis, **misc)
        self.plans = build_level_plans(num_channels, maxdim, max_zf, mlp, tau0, self.input_func_node.out_order)
        self.lgn_cg = LGNCG(self.plans, level_gain, weight_init, mlp, mlp_depth, mlp_width, activation, **misc)
        self.tau_cg_levels_node = self.lgn_cg.tau_levels_node
        self.tau_dict["cg_layers"] = self.tau_cg_levels_node.copy()

        tau_last = dict(self.plans[-1].tau_out)
        self.tau_output = {w: 1 for w in tau_last}
        self.tau_output[(0, 0)] = tau_output_scalars
        self.tau_output[(1, 1)] = tau_output_vectors
        self.tau_dict["output"] = self.tau_output
        self.mix_to_output = MixReps(tau_last, self.tau_output, **misc)
        self.__num_param = sum(p.nelement() for p in self.parameters() if p.requires_grad)
        self.use_fused = True      # False: force the per-operator module/autograd path (cross-checks, tests)
        self._flatten_parameters()

    @property
    def num_learnable_parameters(self) -> int:
        return self.__num_param

    def forward(self, latent_features, covariance_test: bool = False, nodes_all: List[GVec] = None):
        self._require_gpu()
        self._check_views()
        if covariance_test and nodes_all is None:
            raise ValueError("covariance_test is set to True, but the full node features from the encoder is not passed in!")
        lat_v = latent_features[(1, 1)].to(device=self.device, dtype=self.dtype)               # (2,B,1,T,4)
        if not covariance_test and self.use_fused and self._fused_ok():
            # the whole decoder is one native call (and one more for its backward): csrc/step.hip lgn_decoder_fwd/bwd_f64
            return ops.DecoderFn.apply(self, lat_v, self.flat_params)
        self._bind(self._tracked_views())
        try:
            return self._forward_modular(lat_v, covariance_test, nodes_all)
        finally:
            self._bind(self._p_stores)

    def _fused_ok(self) -> bool:
        """True when a whole-network native implementation covers this configuration (lgn/ops.py: native_kind)."""
        # (and its per-jet end stages fit a CU's LDS: plan-time query, lgn/_native.py: end_stages_fit)
        if ops.native_kind(self) is None:
            return False
        fit = self.__dict__.get("_end_fit")
        if fit is None:
            fit = self.__dict__["_end_fit"] = ops.N.end_stages_fit(decoder=self)
        return fit

    def _forward_modular(self, lat_v, covariance_test, nodes_all):
        B = lat_v.shape[1]
        N = self.num_output_particles

        # latent_to_graph: the mixed *channel* axis becomes the *particle* axis (lgn_decoder.py:327-344).
        # The latent scalars are mixed too but never used when there are CG levels.
        g_v = ops.MixFn.apply(self.latent_to_graph.weight((1, 1)), lat_v).squeeze(-3)          # (2,B,N,4) Cartesian
        node_ps = ops.cart_to_canonical_cplx(g_v).contiguous()                                 # (2,B,N,4) canonical

        # input features: zonal (0,0) = 1+1i, (1,1) = the canonical momenta  (zonal_functions.py:169-198)
        s0 = torch.ones(2, B, N, 1, 1, device=self.device, dtype=self.dtype)
        v0 = node_ps.unsqueeze(-2)
        s = ops.MixFn.apply(self.input_func_node.weight((0, 0)), s0).squeeze(-1)
        v = ops.MixFn.apply(self.input_func_node.weight((1, 1)), v0)

        order0 = self.input_func_node.out_order
        f0 = {(0, 0): s.unsqueeze(-1), (1, 1): v}
        feats = run_levels(self, True, {r: f0[r] for r in order0}, node_ps, None)

        last = feats[-1]
        gen_v = ops.MixFn.apply(self.mix_to_output.weight((1, 1)), last[(1, 1)].contiguous())      # (2,B,N,1,4)
        if not covariance_test:
            return ops.canonical_to_cart(gen_v).squeeze(-2)
        gen = GVec({(0, 0): ops.MixFn.apply(self.mix_to_output.weight((0, 0)), last[(0, 0)].contiguous()), (1, 1): gen_v})
        orders = [order0] + [p.out_order for p in self.plans]
        for f, o in zip(feats, orders):
            nodes_all.append(as_gvec(f, o))
        nodes_all.append(gen)
        return gen, nodes_all
