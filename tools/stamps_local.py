"""Phase stamps of the last local_bwd launch of one cfg5 training step (debug build: make -C lgn-autoencoder_amd/csrc stamps)."""
import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch
from lgn import _native as Nn
Nn.LIB_PATH = Nn.LIB_PATH.replace("liblgn_amd.so", "liblgn_amd_stamps.so")
import bench, __graft_entry__ as G
from lgn.step import TrainStep
dev = torch.device("cuda:0")
enc, dec = G._models(30, (4, 4, 6, 6), (6, 6, 4, 4), dev, seed=0, maxdim=3)
tr = TrainStep(enc, dec)
p4, labels = bench.synthetic_jets(512, 30, seed=0)
tr.step({"p4": p4.to(dev), "labels": labels.to(dev)})
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
Nn.lib().lgn_debug_stamps_local(buf)
st = list(buf)
for i in range(1, 64):
    if st[i] > st[0]:
        print(i, st[i] - st[0])
