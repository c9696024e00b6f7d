#!/usr/bin/env python3
"""The DATA-PARALLEL branch of the native step on one GPU: an `nccl` group of world size 1 with force_collective=True runs
[fwd + bwd | all-reduce | L1 + Adam] exactly as a rank of an N-GPU job does (the all-reduce of one rank is a copy).
    python tools/dp_bench.py [jets] [steps]        LGN_AMD_SPLIT_TAIL=1: reductions and radial finalisation as separate launches"""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from lgn.step import NativeTrainStep  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for collective in (True, False):
            enc, dec = G._models(30, bench.CH_ENC, bench.CH_DEC, dev, seed=0)
            p4, labels = bench.synthetic_jets(B, 30, seed=0)
            st = NativeTrainStep(enc, dec, batch_size=B, lr=5e-4, l1_lambda=1e-8, use_graph=True, force_collective=collective)
            st.load_batch({"p4": p4.to(dev), "labels": labels.to(dev)})
            for _ in range(5):
                st.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            print(f"{B} jets, {'data-parallel branch (1 rank, all-reduce in the graph: %s)' % st._in_graph if collective else 'single-process step'}: "
                  f"{ms:.4f} ms per step, {st.launches_per_step} graph launch(es)")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
