#!/usr/bin/env python3
"""cfg2's shape (512 jets x 30 particles, maxdim 2) with 5 ... 8 channels on every level: the level kernels whose C >= 5
instantiations spill registers (tools/kregs.py) -- supported, parity-tested, here TIMED.  Step time per channel count; run under
tools/kprof.sh for the per-kernel averages (profiles/r06_c5to8_levels.txt):
    bash tools/kprof.sh c5to8 40 -- python3 tools/wide_levels.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from lgn.step import NativeTrainStep  # noqa: E402

dev = torch.device("cuda:0")
B, N = 512, 30
p4, labels = bench.synthetic_jets(B, N, seed=0)
batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
for C in (4, 5, 6, 7, 8):
    enc, dec = G._models(N, (C,) * 4, (C,) * 4, dev, seed=0)
    st = NativeTrainStep(enc, dec, batch_size=B, use_graph=True)
    st.load_batch(batch)
    for _ in range(30):
        st.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"C = {C} on every level: {ms:.4f} ms per step, {B / ms * 1e3:.0f} jets/s", flush=True)
    del st, enc, dec
