"""Worker of tests/test_gpu_step.py::test_native_step_two_ranks_match_single_process: one rank of a 2-rank
data-parallel NativeTrainStep.  Both ranks share the one GPU of the test box and talk over gloo (RCCL refuses two ranks
on one device); the code path is the one bench.py runs under torch.distributed.run with RCCL.
    python _dist_native_worker.py RANK WORLD PORT OUTDIR JETS_PER_RANK STEPS [native|captured]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lgn-autoencoder_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, outdir, per_rank, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import __graft_entry__ as G
    from lgn.step import CapturedModuleStep, NativeTrainStep
    cls = CapturedModuleStep if (len(sys.argv) > 7 and sys.argv[7] == "captured") else NativeTrainStep
    dev = torch.device("cuda:0")
    enc, dec = G._models(bench.N_PART, bench.CH_ENC, bench.CH_DEC, dev, seed=0)
    step = cls(enc, dec, batch_size=per_rank, lr=5e-4, l1_lambda=1e-8, use_graph=True)
    p4, labels = bench.synthetic_jets(per_rank * world, bench.N_PART, seed=5)
    sl = slice(rank * per_rank, (rank + 1) * per_rank)
    batch = {"p4": p4[sl].to(dev), "labels": labels[sl].to(dev)}
    losses = []
    for _ in range(steps):
        loss, _ = step.step(batch)
        losses.append(float(loss))
    torch.cuda.synchronize()
    torch.save({"params": step.flat.flat.detach().cpu(), "grad": step.flat.grad.detach().cpu(), "losses": losses},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
