"""
ctypes binding of liblgn_amd.so (C ABI: include/lgn_amd.h).

This is the only place the product touches the native library.  There is NO fallback: if the
library is missing, was built for another ABI version, or a call fails, a RuntimeError is raised.
PyTorch is used for device memory and streams only (raw ``data_ptr()`` + the current HIP stream).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch  # imported before the .so so that the process-wide libamdhip64 is torch's

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LGN_AMD_LIB") or os.path.join(_HERE, "_lib", "liblgn_amd.so")   # LGN_AMD_LIB: debug builds (tools/)
ABI_VERSION = 17
FINALIZE_SCRATCH = 2048      # include/lgn_amd.h: LGN_FINALIZE_SCRATCH

_lib: Optional[C.CDLL] = None

_vp, _i, _ip = C.c_void_p, C.c_int, C.POINTER(C.c_int)

# name -> argtypes (all return int); mirrors include/lgn_amd.h one to one
_SIGNATURES = {
    "lgn_level_fwd_f64": [_i] * 5 + [_vp] * 18,
    "lgn_level_bwd_partial_rows": [_i, _i, _i, _ip, _ip],
    "lgn_level_rad_partial_len": [_i, _i],
    "lgn_level_jet_split": [_i, _i],
    "lgn_cg_product_fwd_f64": [_i] * 8 + [_vp] * 7,
    "lgn_cg_product_bwd_f64": [_i] * 8 + [_vp] * 9,
    "lgn_level_bwd_f64": [_i] * 5 + [_vp] * 24,
    "lgn_reduce_partials_f64": [_vp, _i, _i, _vp, _i, _vp],
    "lgn_radial_finalize_f64": [_vp, _i] + [_vp] * 13,
    "lgn_cgmlp_fwd_f64": [_i] * 5 + [_vp] * 5,
    "lgn_cgmlp_partial_rows": [_i, _i],
    "lgn_cgmlp_bwd_f64": [_i] * 5 + [_vp] * 6 + [_i, _vp],
    "lgn_mixreps_fwd_f64": [_i] * 4 + [_vp] * 4,
    "lgn_chamfer_f64": [_i] * 3 + [_vp] * 2 + [_i] + [_vp] * 4,
    "lgn_mixreps_partial_rows": [_i],
    "lgn_mixreps_bwd_f64": [_i] * 4 + [_vp] * 6,
}


class LocalTables(C.Structure):
    """lgn_local_tables of include/lgn_amd.h (device pointers)."""
    _fields_ = [("n_rows", C.c_int), ("n_out", C.c_int), ("n_w", C.c_int), ("n_terms", C.c_int), ("n_u", C.c_int),
                ("n_x", C.c_int), ("n_units", C.c_int), ("static_kind", C.c_int)] + [
        (name, C.c_void_p) for name in ("row_ptr", "t_type", "t_a", "t_b", "t_coef", "out_dim", "out_nblk", "out_row0", "out_q0",
                                        "out_w0", "u_ptr", "u_row", "u_coef", "x_ptr", "x_row", "x_other", "x_coef")] + [("h_out_w0", C.c_int * 8)]


_tp = C.POINTER(LocalTables)


_POOL_OPS = {"min": 0, "max": 1, "mean": 2, "average": 2}


def pool_code(map_to_latent: str):
    """LGN_POOL(...) code of include/lgn_amd.h for a --map-to-latent string (aggregate(), lgn/models/lgn_encoder.py:419-496):
    up to four of min / max / mean joined by '&' (concatenated) or '+' (averaged), or 'mix' (no pooling: the latent MixReps takes
    all N C (particle, channel) pairs).  None: not a map the whole-network kernels implement ('sum' -- which returns an extra axis
    in the reference -- or mixed separators)."""
    m = map_to_latent.lower()
    if m == "mix":
        return 1 | (3 << 4)
    if "&" in m and "+" in m:
        return None
    avg = "+" in m
    ops = m.split("+") if avg else m.split("&")
    if not 1 <= len(ops) <= 4 or any(o not in _POOL_OPS for o in ops):
        return None
    code = len(ops) | (int(avg) << 3)
    for i, o in enumerate(ops):
        code |= _POOL_OPS[o] << (4 + 2 * i)
    return code


def end_stages_fit(encoder=None, decoder=None, junction: bool = False) -> bool:
    """Plan-time fit query (lgn_*_lds_bytes): do the per-jet end stages of the whole-network calls -- input / latent stage of the
    encoder, input / output stage of the decoder, with `junction` also the fused encoder-latent + decoder-input kernels of the
    whole-step call -- fit the 160 KiB of LDS of a CU?  A jet is one workgroup there; e.g. 'mean&min&max' at N = 150 does not fit
    and takes the per-operator path."""
    L = lib()
    need = 0
    if encoder is not None:
        code = pool_code(encoder.map_to_latent)
        if code is None:
            return False
        n = encoder.num_input_particles           # (counts the jet node of jet_features: lgn/models/encoder.py)
        ch = encoder.num_channels
        ts, tv = encoder.tau_latent[(0, 0)], encoder.tau_latent[(1, 1)]
        need = max(need, L.lgn_encoder_end_lds_bytes(n, ch[0], max(1, encoder.tau_input_scalars), ch[-1], ts, tv, code))
        if junction and decoder is not None:
            need = max(need, L.lgn_junction_lds_bytes(n, ch[-1], ts, tv, code, decoder.num_channels[0]))
    if decoder is not None:
        ch = decoder.num_channels
        need = max(need, L.lgn_decoder_end_lds_bytes(decoder.num_output_particles, ch[0], decoder.tau_latent_vectors, ch[-1]))
    return 0 <= need <= LDS_LIMIT


def pool_blocks(code: int) -> int:
    """Output blocks per latent channel: one per pooling under '&', one under '+' (0 = min&max)."""
    if code == 0:
        return 2
    return 1 if (code >> 3) & 1 else code & 7


class NetDesc(C.Structure):
    """lgn_net_desc of include/lgn_amd.h."""
    _fields_ = [("B", C.c_int), ("N", C.c_int), ("n_levels", C.c_int), ("enc_channels", C.c_int * 5),
                ("dec_channels", C.c_int * 5), ("tau_s", C.c_int), ("tau_v", C.c_int), ("mlp_hidden_mul", C.c_int),
                ("mlp_nlin", C.c_int), ("tau_v_in", C.c_int),
                ("enc_tables", _tp * 4), ("dec_tables", _tp * 4),
                ("enc_Q", C.c_int * 5), ("enc_qs", C.c_int * 5), ("enc_qv", C.c_int * 5),
                ("dec_Q", C.c_int * 5), ("dec_qs", C.c_int * 5), ("dec_qv", C.c_int * 5), ("flags", C.c_int),
                ("activation", C.c_int), ("n_in_scalars", C.c_int), ("latent_pool", C.c_int), ("dec_N", C.c_int)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.flags = net_flags()


NET_NO_STATIC = 1
# kernel-selecting cross-check switches -> LGN_NET_* bits of include/lgn_amd.h
_NET_FLAG_ENV = {"LGN_AMD_NO_STATIC": 1, "LGN_AMD_DEC_PAIRWISE": 2, "LGN_AMD_LEVEL_V2": 4, "LGN_AMD_MLP_V1": 8,
                 "LGN_AMD_MOMENTS_V1": 16, "LGN_AMD_BWD_ORDERED": 64, "LGN_AMD_SPLIT_TAIL": 128, "LGN_AMD_DEC_UNFUSED": 256, "LGN_AMD_MOMENTS_SPLIT": 512,
                 "LGN_AMD_MLP_BWD1": 1024}
# LGN_ACT_* of include/lgn_amd.h: the names get_activation_fn accepts (lgn/nn/generic_levels.py:119-135)
ACTIVATIONS = {"leakyrelu": 0, "relu": 1, "elu": 2, "sigmoid": 3, "logsigmoid": 4, "atan": 5}


def activation_id(name: str) -> int:
    try:
        return ACTIVATIONS[name.lower()]
    except KeyError:
        raise ValueError(f"Activation function {name} not implemented!") from None


def net_flags() -> int:
    """LGN_NET_* bits of a descriptor created now (include/lgn_amd.h): the debug switches are read HERE, once, and frozen into
    the descriptor -- the native whole-network calls never read the environment."""
    return sum(bit for name, bit in _NET_FLAG_ENV.items() if os.environ.get(name, "") == "1")


_dp = C.POINTER(NetDesc)
_ll = C.c_longlong
_d = C.c_double
_SIGNATURES.update({
    "lgn_moments_fwd_f64": [_i] * 5 + [_vp] * 12,
    "lgn_moments_bwd_f64": [_i] * 5 + [_vp] * 16,
    "lgn_local_fwd_f64": [_i] * 5 + [_tp] + [_vp] * 5,
    "lgn_local_partial_rows": [_i],
    "lgn_local_fwd_static_f64": [_i] * 4 + [_vp] * 3 + [_ip] + [_vp] * 3 + [_i, _vp],
    "lgn_local_bwd_static_f64": [_i] * 4 + [_vp] * 3 + [_ip] + [_vp] * 8,
    "lgn_local_bwd_f64": [_i] * 5 + [_tp] + [_vp] * 8,
    "lgn_step_param_slots": [_dp, _i],
    "lgn_step_fwd_bwd_f64": [_dp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp],
    "lgn_encoder_fwd_f64": [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp],
    "lgn_encoder_bwd_f64": [_dp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _ll, _vp],
    "lgn_decoder_fwd_f64": [_dp, _vp, _vp, _vp, _vp, _ll, _vp, _vp],
    "lgn_decoder_bwd_f64": [_dp, _vp, _vp, _ll, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _ll, _vp],
    "lgn_step_finalize_f64": [_vp, _vp, _ll, _vp, _i, _d, _vp, _vp, _vp, _d, _d, _d, _d, _i, _vp, _vp],
    "lgn_step_train_f64": [_dp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _i, _d, _vp, _vp, _vp, _d, _d, _d, _d, _i,
                           _vp, _vp],
})
_LL_SIGNATURES = {          # entry points that return a long long
    "lgn_step_workspace_doubles": [_dp],
    "lgn_net_workspace_doubles": [_dp, _i, _i],
    "lgn_moments_scratch_doubles": [_i, _i, _i, _i],
    "lgn_local_static_packed_doubles": [_i, _i, _i],
    "lgn_encoder_end_lds_bytes": [_i] * 7,
    "lgn_decoder_end_lds_bytes": [_i] * 4,
    "lgn_junction_lds_bytes": [_i] * 6,
}
LDS_LIMIT = 160 * 1024      # LGN_LDS_LIMIT of include/lgn_amd.h
EXPORTED_SYMBOLS = ["lgn_abi_version", "lgn_last_error"] + list(_LL_SIGNATURES) + list(_SIGNATURES)


def lib() -> C.CDLL:
    """Load (once) and return the native library; raises if it is absent or mismatched."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"liblgn_amd.so not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C lgn-autoencoder_amd/csrc`. There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        l.lgn_abi_version.restype = C.c_int
        l.lgn_last_error.restype = C.c_char_p
        got = l.lgn_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"liblgn_amd.so ABI version {got} != expected {ABI_VERSION}; rebuild the library")
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name, argtypes in _LL_SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = C.c_longlong
        _lib = l
    return _lib


def last_error() -> str:
    return lib().lgn_last_error().decode()


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("liblgn_amd.so operates on GPU tensors only (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("non-contiguous tensor passed to the native library")
    return t.data_ptr()


def f64(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float64:
        raise RuntimeError(f"the native path is fp64 (reference precision); got {t.dtype}")
    return t.contiguous()


# ---------------------------------------------------------------------------------------------
# thin wrappers (shape bookkeeping only)
# ---------------------------------------------------------------------------------------------

def level_fwd(decoder, s_in, v_in, p, mask, rad, wm0, wm1):
    """rad = (a, b, c, w0, b0, w1, b1); returns (ag0, ag1, s_out, v_out)."""
    _, B, N, Cc = s_in.shape
    CO = wm0.shape[1]
    dev, dt = s_in.device, s_in.dtype
    ag0 = torch.empty(2, B, N, 2 * Cc, device=dev, dtype=dt)
    ag1 = torch.empty(2, B, N, 2 * Cc, 4, device=dev, dtype=dt)
    s_out = torch.empty(2, B, N, CO, device=dev, dtype=dt)
    v_out = torch.empty(2, B, N, CO, 4, device=dev, dtype=dt)
    a, b, c, w0, b0, w1, b1 = rad
    rc = lib().lgn_level_fwd_f64(B, N, Cc, CO, int(decoder), ptr(s_in), ptr(v_in), ptr(p), ptr(mask),
                                 ptr(a), ptr(b), ptr(c), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(wm0), ptr(wm1),
                                 ptr(ag0), ptr(ag1), ptr(s_out), ptr(v_out), stream_ptr())
    _check(rc, "lgn_level_fwd_f64")
    return ag0, ag1, s_out, v_out


def reduce_partials(part: torch.Tensor, out: torch.Tensor, accumulate: bool = False):
    rows, n = part.shape
    _check(lib().lgn_reduce_partials_f64(ptr(part), rows, n, ptr(out), int(accumulate), stream_ptr()),
           "lgn_reduce_partials_f64")
    return out


def level_bwd(decoder, s_in, v_in, p, mask, rad, wm0, wm1, ag0, ag1, g_s_out, g_v_out, g_p=None):
    """Returns (g_s_in, g_v_in, g_wm0, g_wm1, rad_grads) where rad_grads is
    encoder: (g_a, g_b, g_c, g_w0, g_b0, g_w1, g_b1); decoder: (g_b0, g_b1).  g_p is accumulated in place."""
    _, B, N, Cc = s_in.shape
    CO = wm0.shape[1]
    dev, dt = s_in.device, s_in.dtype
    L = lib()
    rm, rr = C.c_int(), C.c_int()
    _check(L.lgn_level_bwd_partial_rows(B, N, int(decoder), C.byref(rm), C.byref(rr)), "lgn_level_bwd_partial_rows")
    nmix = 4 * CO * 5 * Cc
    nrad = L.lgn_level_rad_partial_len(Cc, int(decoder))
    part_mix = torch.empty(rm.value, nmix, device=dev, dtype=dt)
    part_rad = torch.empty(rr.value, nrad, device=dev, dtype=dt)
    g_ag = torch.empty(B, N, 20 * Cc, device=dev, dtype=dt)
    g_s_in = torch.empty_like(s_in)
    g_v_in = torch.empty_like(v_in)
    a, b, c, w0, b0, w1, b1 = rad
    rc = L.lgn_level_bwd_f64(B, N, Cc, CO, int(decoder), ptr(s_in), ptr(v_in), ptr(p), ptr(mask),
                             ptr(a), ptr(b), ptr(c), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(wm0), ptr(wm1),
                             ptr(ag0), ptr(ag1), ptr(g_s_out), ptr(g_v_out), ptr(g_ag), ptr(g_s_in), ptr(g_v_in),
                             ptr(g_p), ptr(part_mix), ptr(part_rad), stream_ptr())
    _check(rc, "lgn_level_bwd_f64")
    g_mix = torch.empty(nmix, device=dev, dtype=dt)
    reduce_partials(part_mix, g_mix)
    g_wm0 = g_mix[: nmix // 2].view(2, CO, 5 * Cc)
    g_wm1 = g_mix[nmix // 2:].view(2, CO, 5 * Cc)
    tot = torch.empty(nrad, device=dev, dtype=dt)
    reduce_partials(part_rad, tot)
    if decoder:
        rad_grads = (tot[:Cc], tot[Cc:])
    else:
        g_a, g_b, g_c = (torch.empty_like(x) for x in (a, b, c))
        g_w0, g_b0, g_w1, g_b1 = (torch.empty_like(x) for x in (w0, b0, w1, b1))
        _check(L.lgn_radial_finalize_f64(ptr(tot), Cc, ptr(a), ptr(b), ptr(c), ptr(w0), ptr(w1), ptr(g_a), ptr(g_b),
                                         ptr(g_c), ptr(g_w0), ptr(g_b0), ptr(g_w1), ptr(g_b1), stream_ptr()),
               "lgn_radial_finalize_f64")
        rad_grads = (g_a, g_b, g_c, g_w0, g_b0, g_w1, g_b1)
    return g_s_in, g_v_in, g_wm0, g_wm1, rad_grads


def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = ptr(t)
    return arr


def cgmlp_fwd(s_in, ws, bs, act: int = 0):
    _, B, N, Cc = s_in.shape
    H = ws[0].shape[0]
    s_out = torch.empty_like(s_in)
    rc = lib().lgn_cgmlp_fwd_f64(B * N, Cc, H, len(ws), act, _ptr_array(ws), _ptr_array(bs), ptr(s_in), ptr(s_out), stream_ptr())
    _check(rc, "lgn_cgmlp_fwd_f64")
    return s_out


def cgmlp_bwd(s_in, ws, bs, g_out, act: int = 0):
    """Returns (g_in, [g_w...], [g_b...])."""
    _, B, N, Cc = s_in.shape
    H = ws[0].shape[0]
    L = lib()
    rows = L.lgn_cgmlp_partial_rows(B * N, H)
    psize = sum(w.numel() + b.numel() for w, b in zip(ws, bs))
    part = torch.empty(rows, psize, device=s_in.device, dtype=s_in.dtype)
    g_in = torch.empty_like(s_in)
    rc = L.lgn_cgmlp_bwd_f64(B * N, Cc, H, len(ws), act, _ptr_array(ws), _ptr_array(bs), ptr(s_in), ptr(g_out), ptr(g_in),
                             ptr(part), psize, stream_ptr())
    _check(rc, "lgn_cgmlp_bwd_f64")
    flat = torch.empty(psize, device=s_in.device, dtype=s_in.dtype)
    reduce_partials(part, flat)
    gws, gbs, off = [], [], 0
    for w, b in zip(ws, bs):
        gws.append(flat[off: off + w.numel()].view_as(w)); off += w.numel()
        gbs.append(flat[off: off + b.numel()].view_as(b)); off += b.numel()
    return g_in, gws, gbs


def chamfer(x, y, jet_features=False):
    """x (B,N,4), y (B,M,4) real 4-vectors -> per-jet loss terms (B,), d loss / d x, d loss / d y (lgn_chamfer_f64)."""
    x, y = f64(x), f64(y)
    B, Np, M = x.shape[0], x.shape[1], y.shape[1]
    out = torch.empty(B + 4 * B * (Np + M), device=x.device, dtype=x.dtype)       # one allocation: [loss_part | gx | gy]
    part, gx, gy = out[:B], out[B:B + 4 * B * Np].view(B, Np, 4), out[B + 4 * B * Np:].view(B, M, 4)
    _check(lib().lgn_chamfer_f64(B, Np, M, ptr(x), ptr(y), int(bool(jet_features)), ptr(part), ptr(gx), ptr(gy), stream_ptr()),
           "lgn_chamfer_f64")
    return part, gx, gy


def mixreps_fwd(w, x):
    """w (2,Co,Ci); x (2,*batch,Ci,d) -> (2,*batch,Co,d)."""
    Co, Ci = w.shape[1], w.shape[2]
    d = x.shape[-1]
    rows = x[0].numel() // (Ci * d)
    y = torch.empty(x.shape[:-2] + (Co, d), device=x.device, dtype=x.dtype)
    _check(lib().lgn_mixreps_fwd_f64(rows, Ci, Co, d, ptr(w), ptr(x), ptr(y), stream_ptr()), "lgn_mixreps_fwd_f64")
    return y


def mixreps_bwd(w, x, g_y, need_gx=True):
    Co, Ci = w.shape[1], w.shape[2]
    d = x.shape[-1]
    rows = x[0].numel() // (Ci * d)
    L = lib()
    part = torch.empty(L.lgn_mixreps_partial_rows(rows), 2 * Co * Ci, device=x.device, dtype=x.dtype)
    g_x = torch.empty_like(x) if need_gx else None
    _check(L.lgn_mixreps_bwd_f64(rows, Ci, Co, d, ptr(w), ptr(x), ptr(g_y), ptr(g_x), ptr(part), stream_ptr()),
           "lgn_mixreps_bwd_f64")
    g_w = torch.empty_like(w)
    reduce_partials(part, g_w.view(-1))
    return g_x, g_w


# ---------------------------------------------------------------------------------------------
# generic (any maxdim) level
# ---------------------------------------------------------------------------------------------

class DeviceTables:
    """Device copy of the CSR tables built by lgn.plan.build_local_tables."""

    def __init__(self, tab: dict, device):
        self.meta = tab
        self.tensors = {}
        st = LocalTables()
        st.n_rows, st.n_out, st.n_w = tab["n_rows"], tab["n_out"], tab["n_w"]
        st.n_terms, st.n_u, st.n_x = len(tab["ints"]["t_type"]), len(tab["ints"]["u_row"]), len(tab["ints"]["x_row"])
        st.n_units = 0
        from .plan import static_kind
        st.static_kind = static_kind(tab)
        for i, w0 in enumerate(tab["ints"]["out_w0"][:8]):
            st.h_out_w0[i] = w0
        for name, vals in tab["ints"].items():
            t = torch.tensor(vals if len(vals) else [0], dtype=torch.int32, device=device)
            self.tensors[name] = t
            setattr(st, name, t.data_ptr())
        for name, vals in tab["dbls"].items():
            t = torch.tensor(vals if len(vals) else [0.0], dtype=torch.float64, device=device)
            self.tensors[name] = t
            setattr(st, name, t.data_ptr())
        self.struct = st


def moments_fwd(decoder, X, p, mask, rad, out=None):
    _, B, N, Cc, Q = X.shape
    U = torch.empty(B, N, Cc, Q, 5, 2, device=X.device, dtype=X.dtype) if out is None else out.view(B, N, Cc, Q, 5, 2)
    a, b, c, w0, b0, w1, b1 = rad
    _check(lib().lgn_moments_fwd_f64(B, N, Cc, Q, int(decoder), ptr(X), ptr(p), ptr(mask), ptr(a), ptr(b), ptr(c), ptr(w0),
                                     ptr(b0), ptr(w1), ptr(b1), ptr(U), stream_ptr()), "lgn_moments_fwd_f64")
    return U


def moments_bwd(decoder, X, p, mask, rad, gU, gX, g_p):
    """gX (and g_p for the decoder) are accumulated in place; returns the radial gradients like level_bwd."""
    _, B, N, Cc, Q = X.shape
    L = lib()
    nrad = L.lgn_level_rad_partial_len(Cc, int(decoder))
    part = torch.empty(B, nrad, device=X.device, dtype=X.dtype)
    nscr = L.lgn_moments_scratch_doubles(B, N, Cc, int(decoder))
    scratch = torch.empty(nscr, device=X.device, dtype=X.dtype) if nscr > 0 else None
    a, b, c, w0, b0, w1, b1 = rad
    _check(L.lgn_moments_bwd_f64(B, N, Cc, Q, int(decoder), ptr(X), ptr(p), ptr(mask), ptr(a), ptr(b), ptr(c), ptr(w0), ptr(b0),
                                 ptr(w1), ptr(b1), ptr(gU), ptr(gX), ptr(g_p), ptr(part), ptr(scratch), stream_ptr()),
           "lgn_moments_bwd_f64")
    tot = torch.empty(nrad, device=X.device, dtype=X.dtype)
    reduce_partials(part, tot)
    if decoder:
        return (tot[:Cc], tot[Cc:])
    g_a, g_b, g_c = (torch.empty_like(x) for x in (a, b, c))
    g_w0, g_b0, g_w1, g_b1 = (torch.empty_like(x) for x in (w0, b0, w1, b1))
    _check(L.lgn_radial_finalize_f64(ptr(tot), Cc, ptr(a), ptr(b), ptr(c), ptr(w0), ptr(w1), ptr(g_a), ptr(g_b), ptr(g_c),
                                     ptr(g_w0), ptr(g_b0), ptr(g_w1), ptr(g_b1), stream_ptr()), "lgn_radial_finalize_f64")
    return (g_a, g_b, g_c, g_w0, g_b0, g_w1, g_b1)


def local_fwd(tables: DeviceTables, CO, X, U, wcat):
    _, B, N, Cc, Q = X.shape
    Qo = tables.meta["Qout"]
    out = torch.empty(2, B, N, CO, Qo, device=X.device, dtype=X.dtype)
    _check(lib().lgn_local_fwd_f64(B * N, Cc, CO, Q, Qo, C.byref(tables.struct), ptr(X), ptr(U), ptr(wcat), ptr(out), stream_ptr()),
           "lgn_local_fwd_f64")
    return out


def local_bwd(tables: DeviceTables, CO, X, U, wcat, g_out):
    _, B, N, Cc, Q = X.shape
    Qo = tables.meta["Qout"]
    L = lib()
    rows = L.lgn_local_partial_rows(B * N)
    part = torch.empty(rows, wcat.numel(), device=X.device, dtype=X.dtype)
    gU = torch.empty(B, N, Cc, Q, 5, 2, device=X.device, dtype=X.dtype)
    gX = torch.empty_like(X)
    _check(L.lgn_local_bwd_f64(B * N, Cc, CO, Q, Qo, C.byref(tables.struct), ptr(X), ptr(U), ptr(wcat), ptr(g_out), ptr(gU), ptr(gX),
                               ptr(part), stream_ptr()), "lgn_local_bwd_f64")
    g_w = torch.empty_like(wcat)
    reduce_partials(part, g_w)
    return gU, gX, g_w
