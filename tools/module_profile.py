#!/usr/bin/env python3
"""Host-side profile of the module-API training loop (ReferenceLoopStep) at cfg2: where the Python / launch time of a step goes.
    python tools/module_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from lgn.step import ReferenceLoopStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    B = int(os.environ.get("KB_BATCH", "512"))
    dev = torch.device("cuda:0")
    enc, dec = G._models(30, bench.CH_ENC, bench.CH_DEC, dev, seed=0)
    p4, labels = bench.synthetic_jets(B, 30, seed=0)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    tr = ReferenceLoopStep(enc, dec, lr=5e-4, l1_lambda=1e-8)
    for _ in range(20):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(batch)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{steps} steps: host enqueue {1e3 * t_host / steps:.3f} ms/step, with final sync {1e3 * t_all / steps:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        tr.step(batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
