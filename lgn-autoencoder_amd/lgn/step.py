"""
Training-step harness for the accelerated path: encoder -> decoder -> get_real('sum') -> Chamfer
+ l1_lambda * L1 -> backward -> (gradient all-reduce) -> Adam, i.e. the inner loop of the reference's
utils/train.py:280-343 with utils/initialize.py:153-173's optimisers, restated for one process per GPU.

MI355X-first choices (none of them exist in the reference, which is single-device):
  * every parameter of encoder + decoder is a view into ONE flat fp64 buffer, and every ``.grad`` a
    view into one flat gradient buffer: the L1 term, ``zero_grad`` and the optimiser touch two tensors
    instead of ~170, and data parallelism is a single RCCL all-reduce(SUM) of 63.5 k scalars per step
    (latency-bound on xGMI, so one bucket, no overlap machinery).
  * the loss is a SUM over jets (utils/losses/chamfer_loss/chamfer_loss.py:23), so summing the ranks'
    gradients reproduces the single-GPU step on the concatenated batch exactly; the L1 sub-gradient
    l1_lambda * sign(w) is batch independent and is added once, after the all-reduce.
"""
from typing import Dict, Optional

import torch
import torch.distributed as dist


def get_real(x: torch.Tensor, method: str = "sum", eps: float = 1e-16) -> torch.Tensor:
    """utils/utils.py:194-207."""
    m = method.lower()
    if m == "real":
        return x[0]
    if m == "imag":
        return x[1]
    if m == "norm":
        return torch.sqrt(x[0] ** 2 + x[1] ** 2 + eps)
    if m == "sum":
        return x[0] + x[1]
    if m == "mean":
        return (x[0] + x[1]) / 2
    return x[0]


def chamfer_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """ChamferLoss.forward (utils/losses/chamfer_loss/chamfer_loss.py:17-23) with cdist p=2
    (distance_sq.py:263-304): sum over jets of 1/2 (sum_i min_j d_ij + sum_j min_i d_ij)."""
    diffs = -(x.unsqueeze(-2) - y.unsqueeze(-3))
    dist_sq = torch.sum(diffs ** 2, dim=-1)
    return torch.sum((dist_sq.min(dim=-1).values + dist_sq.min(dim=-2).values) / 2)


def normalize_p4(p4: torch.Tensor) -> torch.Tensor:
    """'overall_max' normalisation (utils/normalize_p4.py:39-52)."""
    return p4 / (torch.abs(p4).amax(dim=-1, keepdim=True).amax(dim=-2, keepdim=True) + 1e-16)


class FlatParams:
    """Re-homes the parameters (and gradients) of several modules into two flat buffers."""

    def __init__(self, *modules):
        params = [p for m in modules for p in m.parameters()]
        assert params, "no parameters"
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.flat = torch.empty(total, device=dev, dtype=dt)
        self.grad = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                off += n
        self.params = params

    def zero_grad(self):
        self.grad.zero_()


class TrainStep:
    """One data-parallel training step of the autoencoder (see module docstring)."""

    def __init__(self, encoder, decoder, lr: float = 5e-4, l1_lambda: float = 1e-8, get_real_method: str = "sum",
                 process_group: Optional["dist.ProcessGroup"] = None, optimizer: bool = True):
        self.encoder, self.decoder = encoder, decoder
        self.l1_lambda, self.get_real_method = l1_lambda, get_real_method
        self.flat = FlatParams(encoder, decoder)
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.group = process_group
        # two Adam optimisers with identical hyper-parameters act on disjoint parameters (initialize.py:156-158);
        # one Adam over the flat buffer performs the same element-wise update.
        self.flat_param = torch.nn.Parameter(self.flat.flat)
        self.flat_param.grad = self.flat.grad
        self.opt = torch.optim.Adam([self.flat_param], lr=lr) if optimizer else None

    def forward_backward(self, batch: Dict[str, torch.Tensor]):
        """Returns (total loss as the reference logs it, reconstruction)."""
        self.flat.zero_grad()
        latent = self.encoder(batch)
        recon = self.decoder(latent)
        real = get_real(recon, self.get_real_method)
        target = batch["p4"].to(device=real.device, dtype=real.dtype)
        loss = chamfer_loss(real, target)
        loss.backward()
        if self.world > 1:
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
        total = loss.detach()
        if self.l1_lambda:
            # d/dw lambda*|w| = lambda*sign(w)  (utils/train.py:484-487; torch's abs backward uses sign, sign(0)=0)
            self.flat.grad.add_(torch.sign(self.flat.flat), alpha=self.l1_lambda)
            total = total + self.l1_lambda * self.flat.flat.abs().sum()
        return total, recon

    def step(self, batch):
        total, recon = self.forward_backward(batch)
        if self.opt is not None:
            self.opt.step()
        return total, recon
