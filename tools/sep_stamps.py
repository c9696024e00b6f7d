#!/usr/bin/env python3
"""In-kernel phase stamps of local_bwd_sep_kernel (stamps build: make -C lgn-autoencoder_amd/csrc stamps): one cfg5 step, then the
clock64 stamps of the first workgroup of the LAST Q = 20 decoder level backward (wave 0: 0..8, wave 1: 16..21)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
from lgn import _native as Nn  # noqa: E402
Nn.LIB_PATH = Nn.LIB_PATH.replace("liblgn_amd.so", "liblgn_amd_stamps.so")
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from lgn.step import NativeTrainStep  # noqa: E402

cfg = bench.CONFIGS["cfg5"]
dev = torch.device("cuda:0")
enc, dec = G._models(cfg["N"], cfg["ch_enc"], cfg["ch_dec"], dev, seed=0, maxdim=3)
st = NativeTrainStep(enc, dec, batch_size=cfg["B"], use_graph=False)
p4, labels = bench.synthetic_jets(cfg["B"], cfg["N"], seed=0)
st.load_batch({"p4": p4.to(dev), "labels": labels.to(dev)})
for _ in range(3):
    st.step()
torch.cuda.synchronize()
out = (C.c_longlong * 256)()
if len(sys.argv) > 1 and sys.argv[1] == "static":      # the LAST local_bwd_static launch of the step: the encoder's first level (Kind1)
    Nn.lib().lgn_debug_stamps_local_static(out)
    s = list(out)
    print("local_bwd_static, encoder level 0: wave 0 stamps 0..6:", [s[i] - s[0] for i in range(0, 7)], " wave 1 stamps 10..13:",
          [s[i] - s[0] for i in range(10, 14)])
    sys.exit(0)
Nn.lib().lgn_debug_stamps_local_sep(out)
s = list(out)
w0 = [s[i] - s[0] for i in range(0, 10)]
w1 = [s[i] - s[0] for i in range(16, 22)]
print("wave 0 (start, irrep 0, 1, 2u, 3u, 4u | barrier | tail | end | sums compact):", w0)
print("wave 1 (start, irrep 2p, 3p, 4p, 1 last | tail):", w1)
