#!/usr/bin/env python3
"""In-kernel stamps of the LAST launch of a stamped kernel family inside a whole native step (debug build: make stamps):
    python tools/step_stamps.py lgn_debug_stamps_mlp_chain [batch]"""
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

reader, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
enc, dec = G._models(bench.N_PART, bench.CH_ENC, bench.CH_DEC, dev, seed=0)
p4, labels = bench.synthetic_jets(B, bench.N_PART, seed=0)
step = NativeTrainStep(enc, dec, batch_size=B, use_graph=False)
batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
for _ in range(5):
    step.step(batch)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
getattr(Nn.lib(), reader)(buf)
st = list(buf)
first = min((i for i in range(63) if st[i] > 0), key=lambda i: st[i])
print(f"stamps ({reader}), cycles relative to stamp {first}: first workgroup | last workgroup")
for i in sorted((i for i in range(63) if st[i] > 0), key=lambda i: st[i]):
    print(f"  {i:3d}  t={st[i] - st[first]:8d}  | {st[128 + i] - st[128 + first]:8d}")
