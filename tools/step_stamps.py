#!/usr/bin/env python3
"""Phase stamps of the per-jet end kernels (junction_fwd / junction_bwd / dec_output_loss) inside one native step, debug build:
    make -C lgn-autoencoder_amd/csrc stamps && python tools/step_stamps.py [reader] [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from lgn import _native as Nn  # noqa: E402

Nn.LIB_PATH = Nn.LIB_PATH.replace("liblgn_amd.so", "liblgn_amd_stamps.so")
from lgn.step import NativeTrainStep  # noqa: E402

reader = sys.argv[1] if len(sys.argv) > 1 else "lgn_debug_stamps_net"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
enc, dec = G._models(30, bench.CH_ENC, bench.CH_DEC, dev, seed=0)
p4, labels = bench.synthetic_jets(B, 30, seed=0)
tr = NativeTrainStep(enc, dec, batch_size=B, lr=5e-4, l1_lambda=1e-8, use_graph=False)
tr.load_batch({"p4": p4, "labels": labels})
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()          # (the readers copy all 256 slots)
rc = getattr(Nn.lib(), reader)(buf)
st = list(buf)[:63]
print(f"stamps ({reader}, rc={rc}); ticks = s_memtime (100 MHz constant clock on gfx950: 10 ns each)")
groups = {}
for i, t in enumerate(st):
    if t:
        groups.setdefault(i // 10, []).append((i, t))
for gk, items in sorted(groups.items()):
    t0 = items[0][1]
    print("  group", gk, " ".join(f"{i}:{t - t0}" for i, t in items))
