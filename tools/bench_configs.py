#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity-test cases, not the headline bench line):
cfg1 bs=32, cfg2 bs=512, cfg4 N=150 bs=256 (native fused step) and cfg5 maxdim=3 bs=512 (module/autograd path on the
table-driven kernels).  Prints one JSON line per configuration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from lgn.step import NativeTrainStep, TrainStep  # noqa: E402


def run(name, B, N, ch_enc, ch_dec, maxdim, native, steps, warmup=3):
    dev = torch.device("cuda:0")
    enc, dec = G._models(N, ch_enc, ch_dec, dev, seed=0, maxdim=maxdim)
    tr = NativeTrainStep(enc, dec, batch_size=B) if native else TrainStep(enc, dec)
    p4, labels = bench.synthetic_jets(B, N, seed=0)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    for _ in range(warmup):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = tr.step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "jets_per_s": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "B": B, "N": N,
                      "maxdim": maxdim, "harness": "native+graph" if native else "modules+autograd",
                      "loss": float(loss)}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg2", "cfg4", "cfg5"]
    if "cfg1" in which:
        run("cfg1 bs=32 N=30 maxdim=2", 32, 30, (3, 3, 4, 4), (4, 4, 3, 3), 2, True, 50)
    if "cfg2" in which:
        run("cfg2 bs=512 N=30 maxdim=2", 512, 30, (3, 3, 4, 4), (4, 4, 3, 3), 2, True, 50)
    if "cfg4" in which:
        run("cfg4 bs=256 N=150 maxdim=2", 256, 150, (3, 3, 4, 4), (4, 4, 3, 3), 2, True, 10)
    if "cfg5" in which:
        run("cfg5 bs=512 N=30 maxdim=3 ch 4466/6644", 512, 30, (4, 4, 6, 6), (6, 6, 4, 4), 3, False, 10)
