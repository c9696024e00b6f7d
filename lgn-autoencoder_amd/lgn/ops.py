"""
Autograd bindings of the native kernels + the small PyTorch glue around them.

Each ``torch.autograd.Function`` forwards to one C-ABI entry point of liblgn_amd.so
(include/lgn_amd.h) and its hand-written backward; nothing here computes the hot path in PyTorch.
The glue that stays PyTorch is what SURVEY 8 a-13 lists as IO/pooling glue: input preparation,
the 4x4 basis changes and the min/max latent pooling (argmin + gather).
"""
from math import sqrt

import torch

from . import _native as N


# ---------------------------------------------------------------------------------------------
# native ops
# ---------------------------------------------------------------------------------------------

class LevelFn(torch.autograd.Function):
    """Fused LGNNodeLevel + edge network (csrc/level_fwd.hip, csrc/level_bwd.hip)."""

    @staticmethod
    def forward(ctx, decoder, s_in, v_in, p, mask, ra, rb, rc, w0, b0, w1, b1, wm0, wm1):
        s_in, v_in, p = N.f64(s_in), N.f64(v_in), N.f64(p)
        rad = tuple(N.f64(t.detach()) for t in (ra, rb, rc, w0, b0, w1, b1))
        if decoder:
            rad = (None, None, None, None, rad[4], None, rad[6])
        wm0c, wm1c = N.f64(wm0.detach()), N.f64(wm1.detach())
        ag0, ag1, s_out, v_out = N.level_fwd(decoder, s_in, v_in, p, mask, rad, wm0c, wm1c)
        ctx.decoder = decoder
        ctx.mask = mask
        ctx.rad_full = (ra, rb, rc, w0, b0, w1, b1)
        ctx.save_for_backward(s_in, v_in, p, wm0c, wm1c, ag0, ag1, *[t for t in rad if t is not None])
        return s_out, v_out

    @staticmethod
    def backward(ctx, g_s, g_v):
        s_in, v_in, p, wm0, wm1, ag0, ag1, *radl = ctx.saved_tensors
        decoder = ctx.decoder
        if decoder:
            rad = (None, None, None, None, radl[0], None, radl[1])
            g_p = torch.zeros_like(p)
        else:
            rad = tuple(radl)
            g_p = None
        g_s_in, g_v_in, g_wm0, g_wm1, rg = N.level_bwd(decoder, s_in, v_in, p, ctx.mask, rad, wm0, wm1, ag0, ag1,
                                                       N.f64(g_s), N.f64(g_v), g_p)
        ra, rb, rc, w0, b0, w1, b1 = ctx.rad_full
        if decoder:
            # the decoder's mask is identically zero: basis and Linear weights get exactly zero gradient
            # (lgn/models/lgn_decoder.py:335-340, lgn/nn/position_levels.py:144-149)
            g_rad = (torch.zeros_like(ra), torch.zeros_like(rb), torch.zeros_like(rc), torch.zeros_like(w0),
                     rg[0].view_as(b0), torch.zeros_like(w1), rg[1].view_as(b1))
        else:
            g_rad = (rg[0].view_as(ra), rg[1].view_as(rb), rg[2].view_as(rc), rg[3], rg[4], rg[5], rg[6])
        return (None, g_s_in, g_v_in, g_p, None) + g_rad + (g_wm0, g_wm1)


class CGMLPFn(torch.autograd.Function):
    """CGMLP on the scalar irrep (csrc/mlp.hip).  args: activation id (N.ACTIVATIONS), s_in, w_0, b_0, ..., w_L, b_L."""

    @staticmethod
    def forward(ctx, act, s_in, *wb):
        s_in = N.f64(s_in)
        ws = [N.f64(t.detach()) for t in wb[0::2]]
        bs = [N.f64(t.detach()) for t in wb[1::2]]
        ctx.nl, ctx.act = len(ws), int(act)
        ctx.save_for_backward(s_in, *ws, *bs)
        return N.cgmlp_fwd(s_in, ws, bs, ctx.act)

    @staticmethod
    def backward(ctx, g_out):
        s_in, *rest = ctx.saved_tensors
        ws, bs = rest[:ctx.nl], rest[ctx.nl:]
        g_in, gws, gbs = N.cgmlp_bwd(s_in, ws, bs, N.f64(g_out), ctx.act)
        out = [None, g_in]
        for gw, gb in zip(gws, gbs):
            out += [gw, gb]
        return tuple(out)


class MixFn(torch.autograd.Function):
    """MixReps: y = W x per irrep (csrc/mixreps.hip)."""

    @staticmethod
    def forward(ctx, w, x):
        w, x = N.f64(w.detach()), N.f64(x)
        ctx.save_for_backward(w, x)
        ctx.need_gx = x.requires_grad
        return N.mixreps_fwd(w, x)

    @staticmethod
    def backward(ctx, g_y):
        w, x = ctx.saved_tensors
        g_x, g_w = N.mixreps_bwd(w, x, N.f64(g_y), need_gx=ctx.need_gx)
        return g_w, g_x


# ---------------------------------------------------------------------------------------------
# PyTorch glue (tiny, O(N) per jet): basis changes and pooling
# ---------------------------------------------------------------------------------------------
_H = 1.0 / sqrt(2.0)


def normsq4(p):
    """Minkowski square as the reference forms it: 2 E^2 - sum p^2 (zonal_functions.py:201-218)."""
    sq = p * p
    # explicit left-to-right sum: the order of the reference's CPU reduction (GPU reductions may differ,
    # and for near-massless particles the cancellation amplifies a 1-ulp difference to ~1e-9)
    return 2 * sq[..., 0] - (((sq[..., 0] + sq[..., 1]) + sq[..., 2]) + sq[..., 3])


def cart_to_canonical_real(p):
    """real (...,4) -> planar complex canonical (2,...,4)   (p_to_rep, zonal_functions.py:251-289)."""
    e, px, py, pz = p.unbind(-1)
    zero = torch.zeros_like(e)
    re = torch.stack([e, px * _H, pz, -px * _H], -1)
    im = torch.stack([zero, -py * _H, zero, -py * _H], -1)
    return torch.stack([re, im], 0)


def cart_to_canonical_cplx(p):
    """complex Cartesian (2,...,4) -> complex canonical (2,...,4)  (p_cplx_to_rep, zonal_functions.py:292-341)."""
    (er, xr, yr, zr), (ei, xi, yi, zi) = p[0].unbind(-1), p[1].unbind(-1)
    # c1 = (px - i py)/rt2, c3 = (-px - i py)/rt2
    re = torch.stack([er, (xr + yi) * _H, zr, (-xr + yi) * _H], -1)
    im = torch.stack([ei, (xi - yr) * _H, zi, (-xi - yr) * _H], -1)
    return torch.stack([re, im], 0)


def canonical_to_cart(c):
    """complex canonical (2,...,4) -> complex Cartesian (2,...,4)  (rep_to_p, zonal_functions.py:344-381):
    E = c0, px = (c1 - c3)/rt2, py = i (c1 + c3)/rt2, pz = c2."""
    (c0r, c1r, c2r, c3r), (c0i, c1i, c2i, c3i) = c[0].unbind(-1), c[1].unbind(-1)
    re = torch.stack([c0r, (c1r - c3r) * _H, -(c1i + c3i) * _H, c2r], -1)
    im = torch.stack([c0i, (c1i - c3i) * _H, (c1r + c3r) * _H, c2i], -1)
    return torch.stack([re, im], 0)


def _msq(f):
    """get_msq (lgn_encoder.py:499-505)."""
    return f[..., 0] ** 2 - torch.norm(f[..., 1:], dim=-1) ** 2


def _take_particle(feature, idx):
    """feature (2,B,N,T,d), idx (2,B,T) -> (2,B,1,T,d): per (plane, jet, channel) pick one particle."""
    fp = feature.permute(0, 1, 3, 2, 4)
    ix = idx.unsqueeze(-1).unsqueeze(-1).expand(fp.shape[:3] + (1, fp.shape[-1]))
    return torch.gather(fp, 3, ix).permute(0, 1, 3, 2, 4)


def pool_min(feature):
    """get_min_features (lgn_encoder.py:540-558): arg-min of the value (d=1) or of E^2-|p|^2 (d=4), taken
    independently on the re and im planes; padded particles are not excluded."""
    score = feature[..., 0] if feature.shape[-1] == 1 else _msq(feature)
    return _take_particle(feature, torch.min(score.detach(), dim=-2).indices)


def pool_max(feature):
    """get_max_features (lgn_encoder.py:561-583): the index is always taken from E^2-|p|^2, which for d=1
    is the *square* of the value."""
    return _take_particle(feature, torch.max(_msq(feature).detach(), dim=-2).indices)


def aggregate_latent(method, lat):
    """aggregate() of lgn/models/lgn_encoder.py:419-496 on a dict {(k,n): (2,B,N,T,d)}."""
    m = method.lower()
    if m == "sum":
        return {k: torch.sum(v, dim=-3, keepdim=True).unsqueeze(dim=-3) for k, v in lat.items()}
    if m in ("mean", "average"):
        return {k: torch.mean(v, dim=-3, keepdim=True) for k, v in lat.items()}
    if m == "max":
        return {k: pool_max(v) for k, v in lat.items()}
    if m == "min":
        return {k: pool_min(v) for k, v in lat.items()}
    if m == "mix":
        return lat
    if "+" in m:
        if "mix" in m:
            raise NotImplementedError("Adding with mix aggregation not implemented yet.")
        parts = [aggregate_latent(x, lat) for x in method.split("+")]
        return {k: sum(p[k] for p in parts) / len(parts) for k in lat}
    if "&" in method:
        if "mix" in m:
            raise NotImplementedError("Concatenating with mix aggregation not implemented yet.")
        parts = [aggregate_latent(x, lat) for x in method.split("&")]
        return {k: torch.cat([p[k] for p in parts], dim=3) for k in lat}
    raise NotImplementedError(f"{method} is not implemented.")


BELLS = 20      # bells per radial network the kernels read (csrc/common.hpp: NB; lgn/nn: RadPolyTrig.KERNEL_BELLS)


def _bell_groups(rad):
    """The radial parameters (a, b, c, w0, b0, w1, b1), stored in whole groups of BELLS bells (RadPolyTrig.kernel_params), as one
    tuple per group.  The Linear layer over the bells (lgn/nn/position_levels.py:144-170) is a sum, so radial functions, edges and
    neighbour moments of the level are the sums over the groups; the Linear bias rides with the first group, the others get zeros
    (a masked pair then contributes the bias once, as in the reference)."""
    a, b, c, w0, b0, w1, b1 = rad
    n = a.shape[-1] // BELLS
    if n == 1:
        return [rad]
    z0, z1 = torch.zeros_like(b0), torch.zeros_like(b1)
    out = []
    for k in range(n):
        cut = lambda t: t[..., k * BELLS:(k + 1) * BELLS].contiguous()      # noqa: E731
        out.append((cut(a), cut(b), cut(c), cut(w0), b0 if k == 0 else z0, cut(w1), b1 if k == 0 else z1))
    return out


class GenericLevelFn(torch.autograd.Function):
    """LGNNodeLevel + edge network for arbitrary irreps (csrc/generic_moments.hip, csrc/generic_local.hip).
    args: decoder, tables (N.DeviceTables), CO, X packed (2,B,N,C,Q), p, mask, 7 radial params, then the CatMix
    weights of the output irreps in ``tables.meta['out_irreps']`` order.  Returns the packed output (2,B,N,CO,Qout).
    More than 20 bells (num_basis_fn > 10): the moments kernels run once per group of 20 (see _bell_groups), the groups' moments
    are summed by the native row reduction, and the backward hands the ONE moment gradient to every group."""

    @staticmethod
    def forward(ctx, decoder, tables, CO, X, p, mask, ra, rb, rc, w0, b0, w1, b1, *wmix):
        X, p = N.f64(X), N.f64(p)
        rad = tuple(N.f64(t.detach()) for t in (ra, rb, rc, w0, b0, w1, b1))
        if decoder:       # (the decoder's edge mask is identically zero: only the Linear biases reach the output, whatever the bell count)
            groups = [(None, None, None, None, rad[4], None, rad[6])]
        else:
            groups = _bell_groups(rad)
        wcat = torch.cat([N.f64(w.detach()).reshape(-1) for w in wmix])
        if len(groups) == 1:
            U = N.moments_fwd(decoder, X, p, mask, groups[0])
        else:
            _, B, Nn, Cc, Q = X.shape
            Ug = torch.empty(len(groups), B * Nn * Cc * Q * 10, device=X.device, dtype=X.dtype)
            for k, g in enumerate(groups):
                N.moments_fwd(decoder, X, p, mask, g, out=Ug[k])
            U = torch.empty(B, Nn, Cc, Q, 5, 2, device=X.device, dtype=X.dtype)
            N.reduce_partials(Ug, U.view(-1))
        out = N.local_fwd(tables, CO, X, U, wcat)
        ctx.decoder, ctx.tables, ctx.CO, ctx.mask = decoder, tables, CO, mask
        ctx.rad_full = (ra, rb, rc, w0, b0, w1, b1)
        ctx.wshapes = [w.shape for w in wmix]
        ctx.n_groups = len(groups)
        ctx.save_for_backward(X, p, U, wcat, *([groups[0][4], groups[0][6]] if decoder else rad))
        return out

    @staticmethod
    def backward(ctx, g_out):
        X, p, U, wcat, *radl = ctx.saved_tensors
        decoder = ctx.decoder
        gU, gX, g_w = N.local_bwd(ctx.tables, ctx.CO, X, U, wcat, N.f64(g_out))
        ra, rb, rc, w0, b0, w1, b1 = ctx.rad_full
        if decoder:
            g_p = torch.zeros_like(p)
            rg = N.moments_bwd(decoder, X, p, ctx.mask, (None, None, None, None, radl[0], None, radl[1]), gU, gX, g_p)
            g_rad = (torch.zeros_like(ra), torch.zeros_like(rb), torch.zeros_like(rc), torch.zeros_like(w0),
                     rg[0].view_as(b0), torch.zeros_like(w1), rg[1].view_as(b1))
        else:
            g_p = None
            # every group's call adds its share of d X (d X = sum over pairs of dU conj(edge), linear in the radial functions) and
            # returns the gradients of its own bells; the bias gradients do not depend on the bells: the first group's are kept
            per = [N.moments_bwd(decoder, X, p, ctx.mask, g, gU, gX, g_p) for g in _bell_groups(tuple(radl))]
            if len(per) == 1:
                rg = per[0]
            else:
                cat = lambda i: torch.cat([r[i] for r in per], dim=-1)      # noqa: E731
                rg = (cat(0), cat(1), cat(2), cat(3), per[0][4], cat(5), per[0][6])
            g_rad = (rg[0].view_as(ra), rg[1].view_as(rb), rg[2].view_as(rc), rg[3], rg[4], rg[5], rg[6])
        g_ws, off = [], 0
        for shp in ctx.wshapes:
            n = shp.numel()
            g_ws.append(g_w[off:off + n].view(shp))
            off += n
        return (None, None, None, gX, g_p, None) + g_rad + tuple(g_ws)


# ---------------------------------------------------------------------------------------------
# whole networks: one native call per direction (csrc/step.hip: lgn_encoder_* / lgn_decoder_*)
# ---------------------------------------------------------------------------------------------

def native_kind(net):
    """Which whole-network native implementation (csrc/step.hip) covers this network: 'fused' (every level is the
    maxdim = 2 closed form), 'generic' (table-driven levels, e.g. maxdim = 3) or None (per-operator autograd path only)."""
    kind = net.__dict__.get("_native_kind", 0)
    if kind == 0:
        from .plan import check_maxdim2_layout
        ok = (bool(net.mlp) and 3 <= net.mlp_depth <= 6 and 1 <= net.num_basis_fn <= 10 and 1 <= net.num_cg_levels <= 4
              and all(1 <= c <= 8 for c in net.num_channels) and net.mlp_width * 2 * max(net.num_channels[1:]) <= 96
              and (not hasattr(net, "map_to_latent") or N.pool_code(net.map_to_latent) is not None))
        kind = None
        if ok:
            fused = all(m == 2 for m in net.level_maxdim)
            if fused:
                try:
                    for plan in net.plans:
                        check_maxdim2_layout(plan)
                except RuntimeError:
                    fused = False
            if fused:
                kind = "fused"
            else:
                p0 = net.plans[0]
                if sorted(p0.node_order) == [(0, 0), (1, 1)] and all((0, 0) in p.out_order and (1, 1) in p.out_order for p in net.plans):
                    kind = "generic"
        net.__dict__["_native_kind"] = kind
    return kind


def slot_tensors(net, decoder: bool):
    """Parameter views of a network in the slot order of include/lgn_amd.h (lgn_step_fwd_bwd_f64).  Table-driven levels
    have one CatMix base per level (the lowest-addressed weight of the level, twice: the second slot is ignored)."""
    generic = native_kind(net) == "generic"
    out = []
    if decoder:
        out += [net.latent_to_graph.weight((0, 0)), net.latent_to_graph.weight((1, 1))]
    out += [net.input_func_node.weight((0, 0)), net.input_func_node.weight((1, 1))]
    for rf in net.rad_funcs.rad_funcs:
        out += rf.flat_params()
    for lvl in net.lgn_cg.node_levels:
        mix = lvl.cat_mix.mix_reps
        if generic:
            first = min((mix.weight(r) for r in lvl.plan.out_order), key=lambda t: t.data_ptr())
            out += [first, first]
        else:
            out += [mix.weight((0, 0)), mix.weight((1, 1))]
    for mlp in net.lgn_cg.mlp_levels:
        out += mlp.flat_params()
    last = net.mix_to_output if decoder else net.mix_reps
    out += [last.weight((0, 0)), last.weight((1, 1))]
    return out


def flat_level_tables(net, lvl: int):
    """Device tables of a table-driven level whose CatMix weight offsets point into the network's flat parameter block
    (relative to the level's lowest-addressed CatMix weight); cached until the block moves."""
    from .plan import build_local_tables
    net._check_views()
    cache = net.__dict__.setdefault("_native_cache", {})
    key = ("tables", lvl)
    if key not in cache:
        mix = net.lgn_cg.node_levels[lvl].cat_mix.mix_reps
        plan = net.plans[lvl]
        base = min(mix.weight(r).data_ptr() for r in plan.out_order)
        offs = {r: (mix.weight(r).data_ptr() - base) // 8 for r in plan.out_order}
        cache[key] = N.DeviceTables(build_local_tables(plan, net.cg_dict, weight_offsets=offs), net.flat_params.device)
    return cache[key]


def describe_network(d, net, decoder: bool):
    """Fill the per-network part of an lgn_net_desc (channels, latent sizes, and for table-driven networks the level tables
    and packed-component layout).  Returns objects that must stay alive as long as the descriptor is used."""
    from .plan import packed_offsets
    keep = []
    ch = d.dec_channels if decoder else d.enc_channels
    for i, c in enumerate(net.num_channels):
        ch[i] = c
    if decoder:
        d.tau_v_in = net.tau_latent_vectors
    else:
        d.tau_s, d.tau_v = net.tau_latent[(0, 0)], net.tau_latent[(1, 1)]
        d.n_in_scalars = net.tau_input_scalars      # > 1: jet_features / data['scalars'] (per-network calls only)
        d.latent_pool = N.pool_code(net.map_to_latent) or 0
    if native_kind(net) == "generic":
        tabs = d.dec_tables if decoder else d.enc_tables
        Q, qs, qv = (d.dec_Q, d.dec_qs, d.dec_qv) if decoder else (d.enc_Q, d.enc_qs, d.enc_qv)
        orders = [net.plans[0].node_order] + [p.out_order for p in net.plans]
        for l, order in enumerate(orders):
            off, q = packed_offsets(order)
            Q[l], qs[l], qv[l] = q, off[(0, 0)], off[(1, 1)]
        import ctypes as C
        for l in range(net.num_cg_levels):
            t = flat_level_tables(net, l)
            keep.append(t)
            tabs[l] = C.pointer(t.struct)
    return keep


class NetHandle:
    """Descriptor + slot offsets + workspace sizes of one network for one batch size (cached on the module)."""

    def __init__(self, net, decoder: bool, B: int):
        import ctypes as C
        d = N.NetDesc()
        d.B, d.n_levels = B, net.num_cg_levels
        other = d.enc_channels if decoder else d.dec_channels
        for i in range(len(net.num_channels)):
            other[i] = 1
        d.N = net.num_output_particles if decoder else net.num_input_particles
        d.tau_s, d.tau_v, d.tau_v_in = 1, 1, 0
        self.keep = describe_network(d, net, decoder)
        d.mlp_hidden_mul, d.mlp_nlin = net.mlp_width, net.mlp_depth + 1
        d.activation = N.activation_id(net.activation)
        lib = N.lib()
        slots = slot_tensors(net, decoder)
        want = lib.lgn_step_param_slots(C.byref(d), int(decoder))
        if want < 0:
            raise RuntimeError(N.last_error())
        assert len(slots) == want, (len(slots), want)
        base = net.flat_params.data_ptr()
        offs = [(t.data_ptr() - base) // 8 for t in slots]
        assert all(0 <= o < net.flat_params.numel() for o in offs)
        self.desc, self.ref = d, C.byref(d)
        self.off = (C.c_int64 * len(offs))(*offs)
        self.n_act = lib.lgn_net_workspace_doubles(self.ref, int(decoder), 0)
        self.n_scratch = lib.lgn_net_workspace_doubles(self.ref, int(decoder), 1)
        if self.n_act < 0 or self.n_scratch < 0:
            raise RuntimeError(N.last_error())
        self.n_params = net.flat_params.numel()


def net_handle(net, decoder: bool, B: int) -> NetHandle:
    net._check_views()
    cache = net.__dict__.setdefault("_native_cache", {})
    key = (B, N.net_flags())           # a handle freezes the layout switches of its descriptor
    h = cache.get(key)
    if h is None:
        h = cache[key] = NetHandle(net, decoder, B)
    return h


def _r16(n: int) -> int:
    return (n + 15) & ~15


def _alloc(n: int, like: torch.Tensor) -> torch.Tensor:
    """Uninitialised buffer of n scalars for a whole-network call.  LGN_AMD_POISON=1 (tests) fills it with NaN: every scalar
    the kernels read must have been written by them first -- padding lanes of partly filled 64-node tiles included."""
    import os
    buf = torch.empty(n, device=like.device, dtype=like.dtype)
    if os.environ.get("LGN_AMD_POISON") == "1":
        buf.fill_(float("nan"))
    return buf


class EncoderFn(torch.autograd.Function):
    """LGNEncoder.forward (lgn/models/lgn_encoder.py:255-336; 'min&max' pooling) as ONE native call, and the backward
    autograd would run through it as one more.  args: net, p4 (B,N,4) already scaled, mask (B,N) uint8, flat_params, and the
    extra input scalars (B,N,K-1) or None (jet_features / data['scalars']: data, no gradient).
    One allocation per direction: forward = [latent scalars | latent vectors | activations kept for the backward],
    backward = [parameter gradients | scratch] (the native call zero-fills gradients + its zero block with one memset)."""

    @staticmethod
    def forward(ctx, net, p4, mask, flat, scalars=None):
        B = p4.shape[0]
        h = net_handle(net, False, B)
        K = h.desc.n_in_scalars
        if (K > 1) != (scalars is not None) or (scalars is not None and tuple(scalars.shape) != (B, p4.shape[1], K - 1)):
            raise ValueError(f"the encoder takes {K} input scalars per particle (the mass + {max(K, 1) - 1} given ones); got "
                             f"{None if scalars is None else tuple(scalars.shape)}")
        scalars = None if scalars is None else N.f64(scalars)
        Ts, Tv, P = h.desc.tau_s, h.desc.tau_v, N.pool_blocks(h.desc.latent_pool)
        ns, nv = _r16(2 * P * B * Ts), _r16(8 * P * B * Tv)
        buf = _alloc(ns + nv + h.n_act, flat)
        lat_s = buf[:2 * P * B * Ts].view(2, B, 1, P * Ts, 1)
        lat_v = buf[ns:ns + 8 * P * B * Tv].view(2, B, 1, P * Tv, 4)
        base = buf.data_ptr()
        rc = N.lib().lgn_encoder_fwd_f64(h.ref, flat.data_ptr(), h.off, N.ptr(p4), N.ptr(mask), N.ptr(scalars), base + 8 * (ns + nv),
                                         h.n_act, base, base + 8 * ns, N.stream_ptr())
        N._check(rc, "lgn_encoder_fwd_f64")
        ctx.h, ctx.act_off, ctx.scalars = h, ns + nv, scalars
        ctx.save_for_backward(p4, mask, flat, buf)
        ctx.set_materialize_grads(False)
        return lat_s, lat_v

    @staticmethod
    def backward(ctx, g_s, g_v):
        p4, mask, flat, buf = ctx.saved_tensors
        h = ctx.h
        npar = _r16(h.n_params)
        out = _alloc(npar + h.n_scratch, flat)
        grads = out[:h.n_params]
        if g_v is None:
            if g_s is None:
                return None, None, None, grads.zero_(), None
            g_v = torch.zeros(2, h.desc.B, 1, N.pool_blocks(h.desc.latent_pool) * h.desc.tau_v, 4, device=flat.device, dtype=flat.dtype)
        base = out.data_ptr()
        rc = N.lib().lgn_encoder_bwd_f64(h.ref, flat.data_ptr(), base, h.n_params, h.off, N.ptr(p4), N.ptr(mask), N.ptr(ctx.scalars),
                                         buf.data_ptr() + 8 * ctx.act_off, h.n_act, N.ptr(None if g_s is None else N.f64(g_s)),
                                         N.ptr(N.f64(g_v)), base + 8 * npar, h.n_scratch, N.stream_ptr())
        N._check(rc, "lgn_encoder_bwd_f64")
        return None, None, None, grads, None


class DecoderFn(torch.autograd.Function):
    """LGNDecoder.forward (lgn/models/lgn_decoder.py:218-303) as one native call per direction.
    args: net, latent vectors (2,B,1,T,4), flat_params -> reconstruction (2,B,N,4)."""

    @staticmethod
    def forward(ctx, net, lat_v, flat):
        lat_v = N.f64(lat_v)
        B = lat_v.shape[1]
        h = net_handle(net, True, B)
        nr = _r16(8 * B * h.desc.N)
        buf = _alloc(nr + h.n_act, flat)
        recon = buf[:8 * B * h.desc.N].view(2, B, h.desc.N, 4)
        base = buf.data_ptr()
        rc = N.lib().lgn_decoder_fwd_f64(h.ref, flat.data_ptr(), h.off, N.ptr(lat_v), base + 8 * nr, h.n_act, base, N.stream_ptr())
        N._check(rc, "lgn_decoder_fwd_f64")
        ctx.h, ctx.act_off = h, nr
        ctx.save_for_backward(lat_v, flat, buf)
        return recon

    @staticmethod
    def backward(ctx, g_recon):
        lat_v, flat, buf = ctx.saved_tensors
        h = ctx.h
        npar, nl = _r16(h.n_params), _r16(lat_v.numel())
        out = _alloc(nl + npar + h.n_scratch, flat)
        g_lat = out[:lat_v.numel()].view(lat_v.shape)
        grads = out[nl:nl + h.n_params]
        base = out.data_ptr()
        rc = N.lib().lgn_decoder_bwd_f64(h.ref, flat.data_ptr(), base + 8 * nl, h.n_params, h.off, N.ptr(lat_v),
                                         buf.data_ptr() + 8 * ctx.act_off, h.n_act, N.ptr(N.f64(g_recon)), base,
                                         base + 8 * (nl + npar), h.n_scratch, N.stream_ptr())
        N._check(rc, "lgn_decoder_bwd_f64")
        return None, g_lat, grads


class L1Fn(torch.autograd.Function):
    """sum |w| over a flat parameter block (CGModule.l1_norm, lgn_encoder.py:249-250): one reduction kernel forward,
    g * sign(w) backward (torch's abs backward: sign(0) = 0)."""

    @staticmethod
    def forward(ctx, flat):
        ctx.save_for_backward(flat)
        return torch.linalg.vector_norm(flat.detach(), 1)

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        return torch.sign(flat.detach()).mul_(g)
