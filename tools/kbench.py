#!/usr/bin/env python3
"""Kernel micro-benchmark: run one native entry point in isolation at the BASELINE cfg2 shape
(for rocprofv3 --pmc / --kernel-trace runs).   python tools/kbench.py level_fwd_enc [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from lgn import _native as Nn  # noqa: E402


def main():
    stamps = os.environ.get("KB_STAMPS")          # name of a stamp reader of the debug build, e.g. lgn_debug_stamps_bwd3
    if stamps:
        Nn.LIB_PATH = Nn.LIB_PATH.replace("liblgn_amd.so", os.environ.get("KB_STAMPS_LIB", "liblgn_amd_stamps.so"))
    what = sys.argv[1] if len(sys.argv) > 1 else "level_fwd_enc"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    B = int(os.environ.get("KB_BATCH", "512"))
    N = int(os.environ.get("KB_N", "30"))
    dev = torch.device("cuda:0")
    enc, dec = G._models(N, bench.CH_ENC, bench.CH_DEC, dev, seed=0)
    p4, labels = bench.synthetic_jets(B, N, seed=0)
    if os.environ.get("KB_SAMEJET"):      # every workgroup gets the same jet: position effects without content effects
        k = int(os.environ["KB_SAMEJET"])
        p4, labels = p4[k:k + 1].expand(B, -1, -1).contiguous(), labels[k:k + 1].expand(B, -1).contiguous()
    p4, labels = p4.to(dev), labels.to(dev)
    g = torch.Generator().manual_seed(1)
    decoder = what.endswith("_dec")
    net = dec if decoder else enc
    lvl = int(os.environ.get("KB_LEVEL", 2 if not decoder else 0))            # default: the C=4 -> 4 level
    C, CO = net.num_channels[lvl], net.num_channels[lvl + 1]
    s = torch.randn(2, B, N, C, dtype=torch.float64, generator=g).to(dev)
    v = torch.randn(2, B, N, C, 4, dtype=torch.float64, generator=g).to(dev)
    rad = tuple(t.detach().contiguous() for t in net.rad_funcs.rad_funcs[lvl].flat_params())
    if decoder:
        rad = (None, None, None, None, rad[4], None, rad[6])
        p = torch.randn(2, B, N, 4, dtype=torch.float64, generator=g).to(dev)
        mask = None
    else:
        p, mask = p4, labels
    mix = net.lgn_cg.node_levels[lvl].cat_mix.mix_reps
    wm0, wm1 = mix.weight((0, 0)).detach().contiguous(), mix.weight((1, 1)).detach().contiguous()
    mlp = net.lgn_cg.mlp_levels[lvl]
    ws = [l.weight.detach().contiguous() for l in mlp.linear]
    bs = [l.bias.detach().contiguous() for l in mlp.linear]
    s_mlp = torch.randn(2, B, N, CO, dtype=torch.float64, generator=g).to(dev)

    if what.startswith("level_fwd"):
        fn = lambda: Nn.level_fwd(decoder, s, v, p, mask, rad, wm0, wm1)                      # noqa: E731
    elif what.startswith("level_bwd"):
        ag0, ag1, so, vo = Nn.level_fwd(decoder, s, v, p, mask, rad, wm0, wm1)
        gs, gv = torch.randn_like(so), torch.randn_like(vo)
        gp = torch.zeros_like(p) if decoder else None
        fn = lambda: Nn.level_bwd(decoder, s, v, p, mask, rad, wm0, wm1, ag0, ag1, gs, gv, gp)  # noqa: E731
    elif what == "mlp_fwd":
        fn = lambda: Nn.cgmlp_fwd(s_mlp, ws, bs)                                                 # noqa: E731
    elif what == "mlp_bwd":
        gy = torch.randn_like(s_mlp)
        fn = lambda: Nn.cgmlp_bwd(s_mlp, ws, bs, gy)                                             # noqa: E731
    else:
        raise SystemExit(f"unknown kernel {what}")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if stamps:      # (reading resets the longest-lifetime slot: the warm-up launches do not count)
        import ctypes
        getattr(Nn.lib(), stamps)((ctypes.c_longlong * 256)())
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{what}: B={B} N={N} C={C}->{CO}  {e0.elapsed_time(e1) * 1e3 / reps:.1f} us per call")
    if stamps:
        import ctypes
        if True:
            n = 256
            buf = (ctypes.c_longlong * n)()
            rc = getattr(Nn.lib(), stamps)(buf)
            st = list(buf)
            w = [x for x in st[64:128] + st[192:256] if x > 0]
            if w and any(x > 0 for x in st[192:256]):
                w0 = min(w)
                for nm, off in (("first", 64), ("last", 192)):
                    ww = [x for x in st[off:off + 64] if x > 0]
                    print(f"{nm} workgroup of the grid: alive from {(min(ww) - w0) / 100:.2f} to {(max(ww) - w0) / 100:.2f} us (100 MHz counter)")
            if 0 < st[63] < 10 ** 9:       # STAMP_LIFE_END: longest lifetime of ANY workgroup over all launches so far
                print(f"longest workgroup lifetime: {st[63] / 100:.2f} us")
                st[63] = 0
            first = min(i for i in range(64) if st[i] > 0)
            print(f"stamps ({stamps}, rc={rc}), shader-clock cycles relative to stamp {first}: first workgroup | last workgroup of the grid")
            for i in range(64):
                if st[i] > st[first]:
                    print(f"  {i:3d}  t={st[i] - st[first]:8d}  | {st[128 + i] - st[128 + first]:8d}")
            last = max(range(64), key=lambda i: st[i])
            cyc, wall = st[last] - st[first], st[64 + last] - st[64 + first]
            if wall > 0:      # wall_clock64 ticks at 100 MHz on gfx9 (csrc/probes/clock_probe.hip calibrates it against the host clock)
                print(f"effective shader clock of workgroup 0 between stamps {first} and {last}: {cyc} cycles / {wall} ticks of the 100 MHz "
                      f"counter = {cyc / wall * 0.1:.3f} GHz")


if __name__ == "__main__":
    main()
