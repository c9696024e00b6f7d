"""World-size-2 gloo tests (CPU) of the data-parallel step logic in lgn/step.py: flat parameter / gradient
buffers, one all-reduce(SUM) of the flat gradient, the L1 sub-gradient added once after the all-reduce.

The HIP kernels cannot run here (no GPU), so the encoder/decoder are replaced by tiny torch stand-ins with
the same call shape; what is under test is the harness' collective + bookkeeping, which is device agnostic.
The full native path under the same harness is covered on the GPU by tests/test_gpu_step.py."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Enc(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(4, 4, dtype=torch.float64) * 0.3)
        self.dead = torch.nn.Parameter(torch.randn(3, dtype=torch.float64))       # never used: L1 gradient only

    def forward(self, batch):
        return batch["p4"] @ self.w


class _Dec(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(4, 4, dtype=torch.float64) * 0.3)

    def forward(self, lat):
        y = lat @ self.w
        return torch.stack([y * 0.75, y * 0.25], 0)          # "complex" output; get_real('sum') adds the planes


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
    from lgn.step import TrainStep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                      # identical replicas
    enc, dec = _Enc(), _Dec()
    step = TrainStep(enc, dec, lr=1e-2, l1_lambda=1e-3)
    g = torch.Generator().manual_seed(123)
    p4 = torch.randn(8, 5, 4, dtype=torch.float64, generator=g)
    shard = p4[rank * 4:(rank + 1) * 4]
    total, _ = step.step({"p4": shard})
    out[rank] = (step.flat.grad.clone(), step.flat.flat.detach().clone(), float(total))
    dist.barrier()
    dist.destroy_process_group()


def _single():
    sys.path.insert(0, os.path.join(ROOT, "lgn-autoencoder_amd"))
    from lgn.step import TrainStep, chamfer_loss, get_real
    torch.manual_seed(0)
    enc, dec = _Enc(), _Dec()
    ref_params = [p.detach().clone() for p in list(enc.parameters()) + list(dec.parameters())]
    step = TrainStep(enc, dec, lr=1e-2, l1_lambda=1e-3)
    g = torch.Generator().manual_seed(123)
    p4 = torch.randn(8, 5, 4, dtype=torch.float64, generator=g)
    total, _ = step.step({"p4": p4})
    # independent autograd reference of the same loss, including the L1 term through autograd
    torch.manual_seed(0)
    e2, d2 = _Enc(), _Dec()
    loss = chamfer_loss(get_real(d2(e2({"p4": p4})), "sum"), p4)
    loss = loss + 1e-3 * sum(p.abs().sum() for p in list(e2.parameters()) + list(d2.parameters()))
    loss.backward()
    gref = torch.cat([p.grad.flatten() for p in list(e2.parameters()) + list(d2.parameters())])
    return step, total, gref, float(loss), ref_params


def test_single_process_step_matches_autograd_and_adam():
    step, total, gref, loss_ref, p0 = _single()
    assert torch.allclose(step.flat.grad, gref, rtol=1e-12, atol=1e-14)
    assert float(total) == pytest.approx(loss_ref, rel=1e-13)
    # first Adam step moves every parameter by lr * sign(grad) (bias-corrected m/sqrt(v) = sign)
    p0 = torch.cat([p.flatten() for p in p0])
    delta = step.flat.flat.detach() - p0
    assert torch.allclose(delta, -1e-2 * torch.sign(gref), atol=1e-7)
    # parameters are views of the flat buffer
    assert step.encoder.w.data_ptr() == step.flat.flat.data_ptr()


def test_two_rank_data_parallel_equals_single_process():
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = 29500 + (os.getpid() % 2000)
        procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
            assert p.exitcode == 0, "distributed worker failed"
        res = dict(out)
    step, total, gref, loss_ref, _ = _single()
    for r in range(2):
        grad, flat, _ = res[r]
        # summed shard gradients + one L1 term == gradient of the full batch on one process
        assert torch.allclose(grad, step.flat.grad, rtol=1e-12, atol=1e-14), f"rank {r} gradient"
        assert torch.allclose(flat, step.flat.flat.detach(), rtol=1e-12, atol=1e-14), f"rank {r} parameters after Adam"
    assert torch.equal(res[0][1], res[1][1]), "replicas diverged"
