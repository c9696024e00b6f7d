#!/usr/bin/env python3
"""Where does a module-API step (ReferenceLoopStep) spend its time?  Prints host enqueue time per step (loop without
device sync), wall time per step, and -- run under `rocprofv3 --kernel-trace --stats` -- the kernel list gives the GPU side.
    python tools/module_api_probe.py [cfg] [steps]"""
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
from lgn.step import ReferenceLoopStep  # noqa: E402


def main():
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    enc, dec = G._models(cfg["N"], cfg["ch_enc"], cfg["ch_dec"], dev, seed=0, maxdim=cfg["maxdim"])
    tr = ReferenceLoopStep(enc, dec)
    p4, labels = bench.synthetic_jets(cfg["B"], cfg["N"], seed=0)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    for _ in range(5):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # host-only cost of the pieces (each followed by a sync so the GPU never back-pressures the host)
    parts = {}

    def host(name, fn, reps=20):
        torch.cuda.synchronize()
        acc = 0.0
        for _ in range(reps):
            a = time.perf_counter()
            r = fn()
            acc += time.perf_counter() - a
            torch.cuda.synchronize()
        parts[name] = 1e6 * acc / reps
        return r

    from lgn.step import chamfer_loss, get_real
    lat = host("encoder_fwd", lambda: enc(batch))
    rec = host("decoder_fwd", lambda: dec(lat))
    loss = host("loss_fwd", lambda: chamfer_loss(get_real(rec, "sum"), batch["p4"]) + 1e-8 * (enc.l1_norm() + dec.l1_norm()))

    def bwd():
        l = chamfer_loss(get_real(dec(enc(batch)), "sum"), batch["p4"]) + 1e-8 * (enc.l1_norm() + dec.l1_norm())
        enc.zero_grad(); dec.zero_grad()
        torch.cuda.synchronize()
        a = time.perf_counter()
        l.backward()
        return time.perf_counter() - a
    parts["backward"] = 1e6 * sum(bwd() for _ in range(20)) / 20
    host("adam_x2", lambda: (tr.opt_enc.step(), tr.opt_dec.step()))
    print(json.dumps({"host_enqueue_us_per_step": 1e6 * (t1 - t0) / steps, "wall_us_per_step": 1e6 * (t2 - t0) / steps,
                      "host_us": parts}))


if __name__ == "__main__":
    main()
