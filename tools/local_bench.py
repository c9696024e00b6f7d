#!/usr/bin/env python3
"""Time (and, with the stamps build, phase-stamp) the table-driven per-node kernels of one heavy cfg5 level
(C=6 -> CO=6, Q=20): lgn_local_fwd_f64 / lgn_local_bwd_f64, and the moments kernels.
    python tools/local_bench.py [stamps]"""
import ctypes
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch
from lgn import _native as Nn
STAMPS = len(sys.argv) > 1 and sys.argv[1] == "stamps"
if STAMPS:
    Nn.LIB_PATH = Nn.LIB_PATH.replace("liblgn_amd.so", "liblgn_amd_stamps.so")
import __graft_entry__ as G

dev = torch.device("cuda:0")
enc, dec = G._models(30, (4, 4, 6, 6), (6, 6, 4, 4), dev, seed=0, maxdim=3)
lvl = 2
tables = enc.level_tables(lvl)
plan = enc.plans[lvl]
B, N, C, CO, Q = 512, 30, plan.channels_in, plan.channels_out, tables.meta["Q"]
g = torch.Generator().manual_seed(0)
X = torch.randn(2, B, N, C, Q, dtype=torch.float64, generator=g).to(dev)
U = torch.randn(B, N, C, Q, 5, 2, dtype=torch.float64, generator=g).to(dev)
mix = enc.lgn_cg.node_levels[lvl].cat_mix.mix_reps
wcat = torch.cat([mix.weight(r).detach().reshape(-1) for r in tables.meta["out_irreps"]])
gout = torch.randn(2, B, N, CO, tables.meta["Qout"], dtype=torch.float64, generator=g).to(dev)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def stamps(name):
    if not STAMPS or not hasattr(Nn.lib(), name):
        return
    buf = (ctypes.c_longlong * 64)()
    getattr(Nn.lib(), name)(buf)
    st = list(buf)
    print("  stamps", name, [(i, st[i] - st[0]) for i in range(1, 64) if st[i] > st[0]])


print("local_fwd us", timed(lambda: Nn.local_fwd(tables, CO, X, U, wcat)))
# compile-time-table kernel on node-innermost layouts
from lgn.plan import static_kind
kind = static_kind(tables.meta)
if kind:
    M = B * N
    tiles = (M + 63) // 64
    Mp = tiles * 64

    def to_tb(t2):          # [2][M][C][K] -> [tile][C][K][2][64]
        z = torch.zeros(2, Mp, t2.shape[2], t2.shape[3], device=dev, dtype=torch.float64)
        z[:, :M] = t2
        return z.reshape(2, tiles, 64, t2.shape[2], t2.shape[3]).permute(1, 3, 4, 0, 2).contiguous()

    def from_tb(tb, K):     # [tile][C][K][2][64] -> [2][M][C][K]
        return tb.permute(3, 0, 4, 1, 2).reshape(2, Mp, tb.shape[1], K)[:, :M]

    XT = to_tb(X.reshape(2, M, C, Q))
    UT = to_tb(U.reshape(M, C, Q * 5, 2).permute(3, 0, 1, 2))
    Qo = tables.meta["Qout"]
    outT = torch.empty(tiles, CO, Qo, 2, 64, device=dev, dtype=torch.float64)
    w0 = (ctypes.c_int * 5)(*tables.meta["ints"]["out_w0"])
    wp = torch.empty(Nn.lib().lgn_local_static_packed_doubles(kind, C, CO), device=dev, dtype=torch.float64)
    call = lambda: Nn._check(Nn.lib().lgn_local_fwd_static_f64(kind, M, C, CO, Nn.ptr(XT), Nn.ptr(UT), Nn.ptr(wcat), w0, Nn.ptr(wp),
                                                               Nn.ptr(outT), None, -1, Nn.stream_ptr()), "static")
    print("local_fwd_static us", timed(call))
    ref = Nn.local_fwd(tables, CO, X, U, wcat)                                   # [2][B][N][CO][Qo]
    got = from_tb(outT, Qo).reshape(2, B, N, CO, Qo)
    print("  static vs v1 max rel err", ((got - ref).abs().max() / ref.abs().max()).item())
print("local_bwd us", timed(lambda: Nn.local_bwd(tables, CO, X, U, wcat, gout)))
if kind:
    goT = to_tb(gout.reshape(2, M, CO, Qo))
    gUT = torch.empty_like(UT)
    gXT = torch.empty_like(XT)
    npk = wp.numel()
    part = torch.empty(tiles, npk, device=dev, dtype=torch.float64)
    gpk = torch.empty(npk, device=dev, dtype=torch.float64)
    gw = torch.zeros_like(wcat)

    def bcall():
        gw.zero_()
        Nn._check(Nn.lib().lgn_local_bwd_static_f64(kind, M, C, CO, Nn.ptr(XT), Nn.ptr(UT), Nn.ptr(wcat), w0, Nn.ptr(wp), Nn.ptr(goT),
                                                    Nn.ptr(gUT), Nn.ptr(gXT), Nn.ptr(part), Nn.ptr(gpk), Nn.ptr(gw), Nn.stream_ptr()), "static bwd")
    print("local_bwd_static us", timed(bcall))
    stamps("lgn_debug_stamps_local_static")
    rgU, rgX, rgw = Nn.local_bwd(tables, CO, X, U, wcat, gout)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    print("  static bwd vs v1: gU", rel(from_tb(gUT, Q * 5).reshape(2, M, C, Q, 5).permute(1, 2, 3, 4, 0).reshape(B, N, C, Q, 5, 2), rgU),
          "gX", rel(from_tb(gXT, Q).reshape(2, B, N, C, Q), rgX), "gW", rel(gw, rgw))
stamps("lgn_debug_stamps_local")
p4 = torch.randn(B, N, 4, dtype=torch.float64, generator=g).to(dev)
mask = torch.ones(B, N, dtype=torch.uint8, device=dev)
rad = tuple(t.detach().contiguous() for t in enc.rad_funcs.rad_funcs[lvl].flat_params())
print("moments_fwd us", timed(lambda: Nn.moments_fwd(False, X, p4, mask, rad)))
gU = torch.randn_like(U)
gX = torch.zeros_like(X)
print("moments_bwd (nodes + G + reduce) us", timed(lambda: Nn.moments_bwd(False, X, p4, mask, rad, gU, gX, None)))
