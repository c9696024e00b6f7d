"""Losses of the training step as native operators.

``ChamferLoss`` is the drop-in of the reference's ``utils.losses.ChamferLoss`` (utils/losses/chamfer_loss/chamfer_loss.py:7-31,
cdist of distance_sq.py:263-304 with its default even p: the plain sum of squared component differences): same constructor, same
``forward(x, y, jet_features=False)``, one HIP kernel (csrc/net_kernels.hip: chamfer_kernel) for the loss and both gradients
instead of ~25 elementwise / reduction launches.  ``lgn.step.NativeTrainStep`` does not use it -- there the loss is the tail of the
decoder's last kernel."""
from typing import Optional

import torch
from torch import nn

from . import _native as N


class ChamferFn(torch.autograd.Function):
    """(x (B,N,4), y (B,M,4), jet_features) -> scalar loss.  The kernel returns the gradients for an upstream gradient of one;
    backward scales them."""

    @staticmethod
    def forward(ctx, x, y, jet_features):
        part, gx, gy = N.chamfer(x.detach(), y.detach(), jet_features)
        ctx.save_for_backward(gx, gy)
        return part.sum()

    @staticmethod
    def backward(ctx, g):
        gx, gy = ctx.saved_tensors
        return (gx * g if ctx.needs_input_grad[0] else None), (gy * g if ctx.needs_input_grad[1] else None), None


class ChamferLoss(nn.Module):
    """utils/losses/chamfer_loss/chamfer_loss.py:7-31.  x, y: real 4-vectors (..., N, 4) / (..., M, 4) with the same leading
    batch shape; returns sum over the batch of (sum_i min_j d_ij + sum_j min_i d_ij) / 2, plus nn.MSELoss() of the summed
    momenta when ``jet_features``."""

    def __init__(self, device: Optional[torch.device] = None):
        super().__init__()
        self.device = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def forward(self, x: torch.Tensor, y: torch.Tensor, jet_features: bool = False):
        x, y = x.to(self.device), y.to(self.device)
        if x.shape[-1] != 4 or y.shape[-1] != 4:
            raise ValueError(f"x and y must be 4-vectors. Found: {x.shape[-1]=} and {y.shape[-1]=}.")     # (3-vectors: reference only)
        if x.device.type != "cuda":
            raise RuntimeError("lgn (MI355X build): ChamferLoss runs only in the HIP kernels of liblgn_amd.so on a GPU device; "
                               f"got tensors on '{x.device}'. There is no CPU fallback.")
        if x.dim() == 2 and y.dim() == 2:                       # one unbatched jet (N, 4): the reference's formula takes it as it is
            x, y = x.unsqueeze(0), y.unsqueeze(0)
        if x.shape[:-2] != y.shape[:-2] or x.dim() < 3:
            raise ValueError(f"x {tuple(x.shape)} and y {tuple(y.shape)} must share their batch shape")
        if x.shape[-2] != y.shape[-2]:
            # the reference adds the (..., N) row minima to the (..., M) column minima elementwise (chamfer_loss.py:20-23): N != M
            # raises there (or silently broadcasts when one of them is 1).  The kernel itself handles N != M (lgn_chamfer_f64).
            raise RuntimeError(f"The size of tensor a ({x.shape[-2]}) must match the size of tensor b ({y.shape[-2]}) at non-singleton "
                               "dimension 1 (the reference's ChamferLoss takes sets of equal size)")
        return ChamferFn.apply(x.reshape(-1, x.shape[-2], 4), y.reshape(-1, y.shape[-2], 4), jet_features)
