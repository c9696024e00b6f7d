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

    def __init__(self, *modules, grad_tail: int = 0):
        """grad_tail: extra scalars allocated right behind the gradients (``self.grad_buf`` = gradients | tail) so that
        a caller can all-reduce gradients and per-jet loss terms with ONE collective."""
        params = [p for m in modules for p in m.parameters()]
        assert params, "no parameters"
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.flat = torch.empty(total, device=dev, dtype=dt)
        self.grad_buf = torch.zeros(total + grad_tail, device=dev, dtype=dt)
        self.grad = self.grad_buf[:total]
        self.tail = self.grad_buf[total:]
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                off += n
        self.params = params
        for m in modules:                       # the networks keep plain views of their flat block: refresh them
            if hasattr(m, "_rebind_views"):
                m._rebind_views()

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


class ReferenceLoopStep:
    """The reference's inner loop, line for line in shape (utils/train.py:283-343, utils/initialize.py:153-158), on
    the module API: ``latent = encoder(batch)``, ``recon = decoder(latent)``, ``ChamferLoss(get_real(recon), p4) +
    l1_lambda * (encoder.l1_norm() + decoder.l1_norm())``, ``zero_grad`` x 2, ``loss.backward()``, two ``torch.optim.Adam``.
    Nothing is re-homed or captured: this is what a user who only swaps the ``lgn`` package gets.  Under data
    parallelism the two flat gradients are all-reduced (SUM) and the L1 term is weighted 1/world per rank.
    ``native_loss``: ``lgn.losses.ChamferLoss`` (one kernel, the drop-in of utils.losses.ChamferLoss) instead of the torch
    restatement of the reference's loss (~25 launches)."""

    def __init__(self, encoder, decoder, lr: float = 5e-4, l1_lambda: float = 1e-8, get_real_method: str = "sum",
                 process_group=None, optimizer: bool = True, native_loss: bool = True):
        self.encoder, self.decoder = encoder, decoder
        if native_loss:
            from .losses import ChamferLoss
            self.loss_fn = ChamferLoss(device=encoder.device)
        else:
            self.loss_fn = chamfer_loss
        self.l1_lambda, self.get_real_method = l1_lambda, get_real_method
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.group = process_group
        self.opt_enc = torch.optim.Adam(encoder.parameters(), lr) if optimizer else None
        self.opt_dec = torch.optim.Adam(decoder.parameters(), lr) if optimizer else None

    def step(self, batch):
        latent = self.encoder(batch)
        recon = self.decoder(latent)
        real = get_real(recon, self.get_real_method)
        target = batch["p4"].to(device=real.device, dtype=real.dtype)
        chamfer = self.loss_fn(real, target)
        l1 = self.encoder.l1_norm() + self.decoder.l1_norm()
        loss = chamfer + (self.l1_lambda / self.world) * l1
        if self.opt_enc is not None:         # utils/train.py:324-325
            self.opt_enc.zero_grad()
            self.opt_dec.zero_grad()
        else:
            self.encoder.zero_grad()
            self.decoder.zero_grad()
        loss.backward()
        if self.world > 1:
            for m in (self.encoder, self.decoder):
                for p in m.parameters():
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
        if self.opt_enc is not None:
            self.opt_enc.step()
            self.opt_dec.step()
        return (chamfer + self.l1_lambda * l1).detach(), recon


# ---------------------------------------------------------------------------------------------------
# fully native step: one C call for encoder -> decoder -> loss -> backward, captured in a HIP graph
# ---------------------------------------------------------------------------------------------------

from .ops import slot_tensors as _slot_tensors  # noqa: E402  (parameter slots of include/lgn_amd.h)


def any_rank(flag: bool, group, device) -> bool:
    """True on EVERY rank iff `flag` is true on ANY rank: one eager all-reduce(MAX) of a one-element tensor."""
    t = torch.tensor([1.0 if flag else 0.0], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.item() != 0.0


def agree_in_graph(try_capture, replay_matches, reset, group, device, strict: bool = False) -> bool:
    """Shall the gradient all-reduce live INSIDE the step's graph?  The same answer on every rank of `group`.

    try_capture() captures [fwd+bwd | all-reduce | L1 + Adam] into the step graph (may raise: the backend refuses to be captured);
    replay_matches() replays that graph once and says whether it reproduced the eager step; reset() discards the graph.  Both
    outcomes are LOCAL -- a capture can fail on one rank only (allocator state, a watchdog) -- and both are exchanged before anybody
    acts on them: a rank that fell back by itself would pair its next gradient-sized eager all-reduce with the other ranks' replayed
    in-graph all-reduce and their one-element flag all-reduce (collectives mismatched in size and order: a hang or silently wrong
    gradients).  The sequence of collectives below is therefore identical on every rank whatever happens locally:
        any_rank(capture failed)  ->  [all ranks captured: replay (one in-graph all-reduce each) -> any_rank(replay wrong)]
    strict: raise instead of falling back (NativeTrainStep(graph_collective=True))."""
    import warnings
    err = None
    try:
        try_capture()
    except Exception as exc:      # noqa: BLE001
        err = exc
    if any_rank(err is not None, group, device):
        if strict:
            raise RuntimeError("the all-reduce could not be captured in the step graph on every rank") from err
        warnings.warn("all-reduce could not be captured in the step graph" +
                      (f" ({type(err).__name__}: {err})" if err is not None else " on another rank") +
                      "; every rank falls back to graph | all-reduce | graph")
        reset()
        return False
    if any_rank(not replay_matches(), group, device):
        if strict:
            raise RuntimeError("the all-reduce captured in the step graph does not reproduce the eager step")
        warnings.warn("the all-reduce captured in the step graph does not reproduce the eager step; "
                      "every rank falls back to graph | all-reduce | graph")
        reset()
        return False
    return True


class NativeTrainStep:
    """Same step as TrainStep, executed by lgn_step_fwd_bwd_f64 / lgn_step_finalize_f64 (csrc/step.hip):
    no autograd graph, no PyTorch kernels, every buffer static.  With ``use_graph=True`` the two native calls
    are captured once into HIP graphs (torch.cuda.CUDAGraph around the ctypes calls -- the kernels are
    enqueued on the capturing stream) and replayed; the gradient all-reduce sits between the two graphs."""

    def __init__(self, encoder, decoder, batch_size: int, lr: float = 5e-4, l1_lambda: float = 1e-8,
                 betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, optimizer: bool = True, use_graph: bool = True,
                 force_collective: bool = False, graph_collective: Optional[bool] = None):
        import ctypes as C
        from . import _native as N
        self.N = N
        from .ops import native_kind as _kind
        if _kind(encoder) is None or _kind(encoder) != _kind(decoder):
            # lgn_step_fwd_bwd_f64 covers networks whose levels are all the fused maxdim=2 closed form or all table driven
            # (maxdim=3), with 20 radial basis functions and 7-layer CGMLPs.  Anything else would be read with the wrong
            # layout -> refuse instead of computing garbage.
            raise NotImplementedError(
                "the native step implements maxdim=2 / maxdim=3 networks (the same kind for encoder and decoder) with "
                "map_to_latent = min / max / mean joined by '&' or '+', CGMLP levels (mlp_depth 3 .. 6), num_basis_fn <= 10 and <= 8 channels; got encoder "
                f"maxdim={encoder.level_maxdim} map_to_latent={encoder.map_to_latent!r} mlp={encoder.mlp} mlp_depth="
                f"{encoder.mlp_depth}, decoder maxdim={decoder.level_maxdim} mlp={decoder.mlp}")
        # jet_features (the encoder works on one node more than the decoder reconstructs) and data['scalars']: the whole-step call
        # takes them for maxdim = 2 networks (lgn_net_desc.dec_N / n_in_scalars, ABI 17), its four end stages then run as launches
        # of their own; table-driven networks keep the module route
        self.split = getattr(encoder, "tau_input_scalars", 1) != 1 or bool(getattr(encoder, "jet_features", False))
        if self.split and (_kind(encoder) != "fused" or encoder.tau_input_scalars > 8):
            raise NotImplementedError("the native step of table-driven (maxdim 3) networks takes the particle masses as the only input "
                                      "scalars: jet_features / extra input scalars run through the module API there (CapturedModuleStep "
                                      "/ native_train_step capture that step into one graph)")
        encoder._require_gpu()
        if not N.end_stages_fit(encoder, decoder, junction=not self.split):
            # a jet's latent / junction stage is ONE workgroup: refused here, at plan time, so that native_train_step (and any caller
            # catching NotImplementedError) takes the module route instead of failing at the first launch
            raise NotImplementedError(
                f"the per-jet latent stage of map_to_latent={encoder.map_to_latent!r} at {encoder.num_input_particles} particles needs "
                "more than the 160 KiB of LDS of a CU; this configuration runs through the module API (per-operator path)")
        self.encoder, self.decoder = encoder, decoder
        self.l1_lambda, self.lr, self.betas, self.eps = l1_lambda, lr, betas, eps
        self.flat = FlatParams(encoder, decoder, grad_tail=batch_size)   # gradients | per-jet Chamfer terms
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.group = process_group
        # force_collective: take the two-graph + all-reduce branch even with one rank (exercises the capture boundaries and
        # the RCCL call on the flat buffer on a single GPU; a 1-rank SUM leaves the buffer unchanged)
        self.collective = self.world > 1 or force_collective
        # graph_collective: capture the all-reduce INSIDE the step's graph (one launch per step under data parallelism).
        # None = try, and fall back to [graph | all-reduce | graph] if the backend refuses the capture; LGN_AMD_GRAPH_COLLECTIVE=0
        # forces the three-launch form.  `self.launches_per_step` says which one is in use after the first step.
        import os as _os
        if graph_collective is None and _os.environ.get("LGN_AMD_GRAPH_COLLECTIVE") == "0":
            graph_collective = False
        if graph_collective is None and self.collective and dist.get_backend(process_group) != "nccl":
            graph_collective = False        # gloo stages through the host: nothing a stream capture could record
        self.graph_collective = graph_collective
        self.launches_per_step = None
        self.optimizer = optimizer
        dev, dt = self.flat.flat.device, self.flat.flat.dtype
        L = encoder.num_cg_levels
        d = N.NetDesc()
        from .ops import describe_network, native_kind
        d.B, d.N, d.n_levels = batch_size, encoder.num_input_particles, L       # (num_input_particles counts the jet node of jet_features)
        self._keep = describe_network(d, encoder, False) + describe_network(d, decoder, True)    # (after FlatParams re-homed the blocks)
        d.mlp_hidden_mul, d.mlp_nlin = encoder.mlp_width, encoder.mlp_depth + 1
        if N.activation_id(encoder.activation) != N.activation_id(decoder.activation):
            raise NotImplementedError("the native step takes ONE activation for the CGMLPs of both networks (as --activation gives them); "
                                      f"got {encoder.activation} / {decoder.activation}")
        d.activation = N.activation_id(encoder.activation)
        fused = native_kind(encoder) == "fused"
        d.dec_N = decoder.num_output_particles if self.split else 0
        if decoder.tau_latent_vectors != N.pool_blocks(d.latent_pool) * d.tau_v or \
                decoder.num_output_particles != encoder.num_input_particles - int(bool(getattr(encoder, "jet_features", False))):
            raise ValueError(f"decoder latent size / particle count does not match the encoder (map_to_latent={encoder.map_to_latent!r} "
                             f"gives {N.pool_blocks(d.latent_pool)} x {d.tau_v} latent vectors, the decoder takes {decoder.tau_latent_vectors})")
        self.desc = d
        lib = N.lib()
        base = self.flat.flat.data_ptr()

        def offsets(net, dec):
            ts = _slot_tensors(net, dec)
            want = lib.lgn_step_param_slots(C.byref(d), int(dec))
            if want < 0:
                raise RuntimeError(N.last_error())
            assert len(ts) == want, (len(ts), want)
            ch = net.num_channels
            for l in range(L):                   # sizes the kernels assume for the per-level slots
                mix0 = ts[(2 if dec else 0) + 2 + 7 * L + 2 * l]
                assert not fused or mix0.numel() == 2 * ch[l + 1] * 5 * ch[l], "CatMix weight is not [2][CO][5C]: not a maxdim=2 level"
                rf = net.rad_funcs.rad_funcs[l]      # (num_basis_fn < 10: stored 20 wide, zero padded -- lgn/nn: RadPolyTrig._kernel_pad)
                assert rf.kernel_params()[0].numel() == 20 and rf.kernel_params()[0].data_ptr() == ts[(2 if dec else 0) + 2 + 7 * l].data_ptr(), \
                    "the radial parameters must be stored 20 bells wide"
            offs = [(t.data_ptr() - base) // 8 for t in ts]
            assert all(0 <= o < self.flat.flat.numel() for o in offs)
            return (C.c_int64 * len(offs))(*offs)

        self.enc_off, self.dec_off = offsets(encoder, False), offsets(decoder, True)
        nws = lib.lgn_step_workspace_doubles(C.byref(d))
        if nws < 0:
            raise RuntimeError(N.last_error())
        self.workspace = torch.empty(nws, device=dev, dtype=dt)
        Nd = decoder.num_output_particles
        self.recon = torch.empty(2, d.B, Nd, 4, device=dev, dtype=dt)
        self.loss_part = self.flat.tail
        self._loss_buf = torch.zeros(3 + N.FINALIZE_SCRATCH, device=dev, dtype=dt)   # results | scratch
        self.loss_out = self._loss_buf[:3]
        self.adam_m = torch.zeros_like(self.flat.flat)
        self.adam_v = torch.zeros_like(self.flat.flat)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        self.p4 = torch.empty(d.B, d.N, 4, device=dev, dtype=dt)           # encoder input (p4 * scale; with jet_features + the jet node)
        # Chamfer target = the UNscaled batch (utils/train.py:285-292); aliases the input when scale == 1 and the nodes are the same
        self.target = self.p4 if encoder.scale == 1.0 and not self.split else torch.empty(d.B, Nd, 4, device=dev, dtype=dt)
        self.mask = torch.empty(d.B, d.N, device=dev, dtype=torch.uint8)
        K = max(1, encoder.tau_input_scalars)
        self.in_scalars = torch.empty(d.B, d.N, K - 1, device=dev, dtype=dt) if K > 1 else None     # jet mass term, data['scalars']
        self.use_graph = use_graph
        self._g1 = self._g2 = None

    # -- raw native calls on the current stream
    def _fwd_bwd(self):
        import ctypes as C
        N = self.N
        rc = N.lib().lgn_step_fwd_bwd_f64(C.byref(self.desc), N.ptr(self.flat.flat), N.ptr(self.flat.grad), self.flat.flat.numel(),
                                          self.enc_off, self.dec_off, N.ptr(self.p4), N.ptr(self.target), N.ptr(self.mask),
                                          N.ptr(self.in_scalars) if self.in_scalars is not None else None,
                                          N.ptr(self.workspace), self.workspace.numel(), N.ptr(self.recon), N.ptr(self.loss_part),
                                          N.stream_ptr())
        N._check(rc, "lgn_step_fwd_bwd_f64")

    def _finalize(self, do_adam: bool):
        N = self.N
        rc = N.lib().lgn_step_finalize_f64(N.ptr(self.flat.flat), N.ptr(self.flat.grad), self.flat.flat.numel(), N.ptr(self.loss_part),
                                           self.loss_part.numel(), float(self.l1_lambda), N.ptr(self.adam_m), N.ptr(self.adam_v),
                                           N.ptr(self.step_dev), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                           float(self.eps), int(do_adam), N.ptr(self._loss_buf), N.stream_ptr())
        N._check(rc, "lgn_step_finalize_f64")

    def _train(self, do_adam: bool):
        """Single process: the whole step in ONE native call (lgn_step_train_f64) -- with no all-reduce between the gradients and the
        optimiser, the reductions, the radial finalisation, L1 + Adam and the loss assembly are one launch (csrc/step_tail.hip)
        instead of three; same results bit for bit (LGN_AMD_SPLIT_TAIL=1 when the step is built: the separate launches)."""
        import ctypes as C
        N = self.N
        rc = N.lib().lgn_step_train_f64(C.byref(self.desc), N.ptr(self.flat.flat), N.ptr(self.flat.grad), self.flat.flat.numel(),
                                        self.enc_off, self.dec_off, N.ptr(self.p4), N.ptr(self.target), N.ptr(self.mask),
                                        N.ptr(self.in_scalars) if self.in_scalars is not None else None,
                                        N.ptr(self.workspace), self.workspace.numel(), N.ptr(self.recon), N.ptr(self.loss_part),
                                        self.loss_part.numel(), float(self.l1_lambda), N.ptr(self.adam_m), N.ptr(self.adam_v),
                                        N.ptr(self.step_dev), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                        float(self.eps), int(do_adam), N.ptr(self._loss_buf), N.stream_ptr())
        N._check(rc, "lgn_step_train_f64")

    def _capture(self):
        # warm up on a side stream (lazy module loads, hipFuncSetAttribute), then capture
        snap = (self.flat.flat.clone(), self.adam_m.clone(), self.adam_v.clone(), self.step_dev.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            if self.collective:       # communicator / algorithm set-up of this message size happens outside the capture
                self._fwd_bwd()
                dist.all_reduce(self.flat.grad_buf, op=dist.ReduceOp.SUM, group=self.group)
                self._finalize(False)
            else:
                self._train(False)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ref = (self.flat.grad_buf.clone(), self._loss_buf[:3].clone())     # the eager step's reduced gradients | loss terms, loss
        self._g1, self._g2, self._in_graph = torch.cuda.CUDAGraph(), None, False
        if self.collective and self.graph_collective is not False:
            # ONE graph: forward + backward | all-reduce(SUM) of gradients and loss terms | L1 + Adam.  RCCL enqueues its
            # kernel on the capturing stream like any other launch; if this backend refuses -- on ANY rank -- or the captured
            # collective does not reproduce the eager step -- on ANY rank -- every rank uses the three-launch form below
            # (agree_in_graph: the ranks exchange both outcomes, none decides from what it saw locally)
            def try_capture():
                with torch.cuda.graph(self._g1):
                    self._fwd_bwd()
                    dist.all_reduce(self.flat.grad_buf, op=dist.ReduceOp.SUM, group=self.group)
                    self._finalize(self.optimizer)

            def reset():
                torch.cuda.synchronize()
                self._g1 = torch.cuda.CUDAGraph()

            self._in_graph = agree_in_graph(try_capture, lambda: self._captured_collective_matches(snap, ref), reset, self.group,
                                            self.flat.flat.device, strict=self.graph_collective is True)
        if self.collective and not self._in_graph:      # the gradient all-reduce sits between two graphs
            self._g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g1):
                self._fwd_bwd()
            with torch.cuda.graph(self._g2, pool=self._g1.pool()):
                self._finalize(self.optimizer)
        elif not self.collective:     # single process: the whole step is ONE graph launch
            with torch.cuda.graph(self._g1):
                self._train(self.optimizer)
        self.launches_per_step = 3 if self._g2 is not None else 1
        with torch.no_grad():   # capture does not execute, but restore anyway in case a backend replays eagerly
            self.flat.flat.copy_(snap[0]); self.adam_m.copy_(snap[1]); self.adam_v.copy_(snap[2]); self.step_dev.copy_(snap[3])

    def _captured_collective_matches(self, snap, ref) -> bool:
        """One replay of the freshly captured [fwd+bwd | all-reduce | L1 + Adam] graph from the snapshotted state: the reduced
        gradient buffer (gradients + L1 sub-gradient | per-jet loss terms of ALL ranks) and the loss must be what the eager
        warm-up step produced from the same state -- a capture that silently dropped the collective would leave the local sums.
        LOCAL verdict; agree_in_graph makes it every rank's."""
        with torch.no_grad():
            self.flat.flat.copy_(snap[0]); self.adam_m.copy_(snap[1]); self.adam_v.copy_(snap[2]); self.step_dev.copy_(snap[3])
        self._g1.replay()
        torch.cuda.synchronize()
        tol = dict(rtol=1e-11, atol=1e-300)       # same kernels; only the reduction order inside RCCL may differ
        return bool(torch.allclose(self.flat.grad_buf, ref[0], **tol) and torch.allclose(self._loss_buf[:3], ref[1], **tol))

    def load_batch(self, batch: Dict[str, torch.Tensor]):
        """Stage a batch into the static input buffers (device-to-device copy; labels/masks as in
        LGNEncoder._prepare_input, lgn/models/lgn_encoder.py:386-398)."""
        p4 = batch["p4"]
        if tuple(p4.shape) != tuple(self.target.shape):
            raise ValueError(f"NativeTrainStep was built for batches of shape {tuple(self.target.shape)}, got {tuple(p4.shape)} "
                             "(static buffers / captured graph: pad or drop the last short batch)")
        if self.split:
            # jet node, jet-mass scalar, data['scalars']: the encoder's own input preparation (lgn_encoder.py:372-411), on the device
            ps, mask, scalars = self.encoder._prepare_input(batch)
            self.p4.copy_(ps)
            self.mask.copy_(mask)
            if self.in_scalars is not None:
                self.in_scalars.copy_(scalars)
            self.target.copy_(p4)
            return
        if self.target is not self.p4:
            self.target.copy_(p4)
            torch.mul(self.target, self.encoder.scale, out=self.p4)
        else:
            self.p4.copy_(p4)
        for key in ("labels", "masks", "mask"):
            if key in batch:
                if tuple(batch[key].shape) != tuple(self.mask.shape):
                    raise ValueError(f"mask shape {tuple(batch[key].shape)} != {tuple(self.mask.shape)}")
                self.mask.copy_(batch[key].to(torch.uint8))
                break
        else:
            self.mask.copy_((p4[..., 0] != 0).to(torch.uint8))

    def step(self, batch: Optional[Dict[str, torch.Tensor]] = None):
        """Runs one step on `batch` (or on the already staged static buffers when batch is None).
        Returns (total loss tensor (device scalar, as the reference logs it), reconstruction (2,B,N,4))."""
        if batch is not None:
            self.load_batch(batch)
        if self.use_graph and self._g1 is None:
            self._capture()
        if self.use_graph:
            self._g1.replay()
            if self._g2 is not None:    # ONE collective per step: gradients and the per-jet loss terms share a buffer
                dist.all_reduce(self.flat.grad_buf, op=dist.ReduceOp.SUM, group=self.group)
                self._g2.replay()
        elif self.collective:
            self._fwd_bwd()
            dist.all_reduce(self.flat.grad_buf, op=dist.ReduceOp.SUM, group=self.group)
            self._finalize(self.optimizer)
        else:
            self._train(self.optimizer)
        return self.loss_out[0], self.recon


class CapturedModuleStep:
    """The training step for every configuration the MODULES run but lgn_step_fwd_bwd_f64 refuses -- ``jet_features`` / extra
    input scalars (the encoder has one node more than the decoder), ``--chamfer-jet-features``, mixed maxdim-2 / maxdim-3 networks,
    ``map_to_latent='sum'``, levels without CGMLP: ``encoder(batch) -> decoder(latent) -> lgn.losses.ChamferLoss -> backward()``
    under autograd (one native call per network and direction where the configuration allows it, per operator otherwise), then
    lgn_step_finalize_f64 (L1 sub-gradient, loss assembly, Adam) -- all of it, input preparation included, captured ONCE into a
    HIP graph on static buffers and replayed (``use_graph``), so that no Python / autograd work is left in the step.  Same
    interface as NativeTrainStep (``load_batch``, ``step``, ``loss_out``, ``flat``); under data parallelism the flat gradient buffer
    (gradients | this rank's Chamfer term) is all-reduced between two graphs."""

    def __init__(self, encoder, decoder, batch_size: int, lr: float = 5e-4, l1_lambda: float = 1e-8, betas=(0.9, 0.999),
                 eps: float = 1e-8, process_group=None, optimizer: bool = True, use_graph: bool = True, get_real_method: str = "sum",
                 chamfer_jet_features: bool = False, extra_scalars: int = 0):
        from . import _native as N
        from .losses import ChamferLoss
        self.N = N
        encoder._require_gpu()
        self.encoder, self.decoder = encoder, decoder
        self.l1_lambda, self.lr, self.betas, self.eps = l1_lambda, lr, betas, eps
        self.get_real_method, self.chamfer_jet_features = get_real_method, chamfer_jet_features
        self.flat = FlatParams(encoder, decoder, grad_tail=1)             # gradients | this rank's Chamfer term
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.group, self.optimizer, self.use_graph = process_group, optimizer, use_graph
        dev, dt = self.flat.flat.device, self.flat.flat.dtype
        n_in = encoder.num_input_particles - (1 if getattr(encoder, "jet_features", False) else 0)    # particles per jet in the batch
        self.batch = {"p4": torch.zeros(batch_size, n_in, 4, device=dev, dtype=dt),
                      "labels": torch.zeros(batch_size, n_in, device=dev, dtype=torch.uint8)}
        if extra_scalars:
            self.batch["scalars"] = torch.zeros(batch_size, encoder.num_input_particles, extra_scalars, device=dev, dtype=dt)
        self.loss_fn = ChamferLoss(device=dev)
        self.loss_part = self.flat.tail
        self._loss_buf = torch.zeros(3 + N.FINALIZE_SCRATCH, device=dev, dtype=dt)
        self.loss_out = self._loss_buf[:3]
        self.adam_m, self.adam_v = torch.zeros_like(self.flat.flat), torch.zeros_like(self.flat.flat)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        self.recon = None
        self._g1 = self._g2 = None
        self.launches_per_step = None

    def _fwd_bwd(self):
        self.flat.grad_buf.zero_()
        recon = self.decoder(self.encoder(self.batch))
        loss = self.loss_fn(get_real(recon, self.get_real_method), self.batch["p4"], jet_features=self.chamfer_jet_features)
        loss.backward()                                   # accumulates into the views of flat.grad the parameters hold
        self.loss_part.copy_(loss.detach().reshape(1))
        self.recon = recon.detach()

    def _finalize(self, do_adam: bool):
        N = self.N
        rc = N.lib().lgn_step_finalize_f64(N.ptr(self.flat.flat), N.ptr(self.flat.grad), self.flat.flat.numel(), N.ptr(self.loss_part), 1,
                                           float(self.l1_lambda), N.ptr(self.adam_m), N.ptr(self.adam_v), N.ptr(self.step_dev),
                                           float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), int(do_adam),
                                           N.ptr(self._loss_buf), N.stream_ptr())
        N._check(rc, "lgn_step_finalize_f64")

    def _capture(self):
        snap = (self.flat.flat.clone(), self.adam_m.clone(), self.adam_v.clone(), self.step_dev.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                        # warm-up: lazy loads, autograd's first-use set-up, allocator state
            for _ in range(2):
                self._fwd_bwd()
                self._finalize(False)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._g1 = torch.cuda.CUDAGraph()
        if self.world > 1:
            self._g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g1):
                self._fwd_bwd()
            with torch.cuda.graph(self._g2, pool=self._g1.pool()):
                self._finalize(self.optimizer)
        else:
            with torch.cuda.graph(self._g1):
                self._fwd_bwd()
                self._finalize(self.optimizer)
        self.launches_per_step = 3 if self._g2 is not None else 1
        with torch.no_grad():
            self.flat.flat.copy_(snap[0]); self.adam_m.copy_(snap[1]); self.adam_v.copy_(snap[2]); self.step_dev.copy_(snap[3])

    def load_batch(self, batch: Dict[str, torch.Tensor]):
        """Stage a batch into the static input tensors the captured graph reads (the encoder's own input preparation -- scale,
        jet node, masks: lgn/models/lgn_encoder.py:338-412 -- is part of the graph)."""
        p4 = batch["p4"]
        if tuple(p4.shape) != tuple(self.batch["p4"].shape):
            raise ValueError(f"CapturedModuleStep was built for batches of shape {tuple(self.batch['p4'].shape)}, got {tuple(p4.shape)}")
        self.batch["p4"].copy_(p4)
        for key in ("labels", "masks", "mask"):
            if key in batch:
                if tuple(batch[key].shape) != tuple(self.batch["labels"].shape):
                    raise ValueError(f"mask shape {tuple(batch[key].shape)} != {tuple(self.batch['labels'].shape)}")
                self.batch["labels"].copy_(batch[key].to(torch.uint8))
                break
        else:
            self.batch["labels"].copy_((p4[..., 0] != 0).to(torch.uint8))
        if ("scalars" in batch) != ("scalars" in self.batch):
            raise ValueError("CapturedModuleStep: data['scalars'] must be present exactly when the step was built with extra_scalars")
        if "scalars" in batch:
            self.batch["scalars"].copy_(batch["scalars"])

    def step(self, batch: Optional[Dict[str, torch.Tensor]] = None):
        """One step on `batch` (or on the staged static buffers).  Returns (total loss (device scalar), reconstruction)."""
        if batch is not None:
            self.load_batch(batch)
        if self.use_graph and self._g1 is None:
            self._capture()
        if self.use_graph:
            self._g1.replay()
            if self._g2 is not None:
                dist.all_reduce(self.flat.grad_buf, op=dist.ReduceOp.SUM, group=self.group)
                self._g2.replay()
        else:
            self._fwd_bwd()
            if self.world > 1:
                dist.all_reduce(self.flat.grad_buf, op=dist.ReduceOp.SUM, group=self.group)
            self._finalize(self.optimizer)
        return self.loss_out[0], self.recon


def native_train_step(encoder, decoder, batch_size: int, **kw):
    """NativeTrainStep where lgn_step_fwd_bwd_f64 covers the configuration (one native call per step), else CapturedModuleStep
    (the module-API step captured into one graph).  Keyword arguments the two do not share go to the one that takes them."""
    import inspect
    import warnings
    takes = {cls: set(inspect.signature(cls.__init__).parameters) - {"self"} for cls in (NativeTrainStep, CapturedModuleStep)}
    unknown = set(kw) - takes[NativeTrainStep] - takes[CapturedModuleStep]
    if unknown:
        raise TypeError(f"native_train_step: unknown keyword argument(s) {sorted(unknown)}")

    def only(cls):
        return {k: v for k, v in kw.items() if k in takes[cls]}

    # (extra_scalars: sizes CapturedModuleStep's buffers; NativeTrainStep reads the count off the encoder -- it does not force the module route)
    needs_modules = bool(kw.get("chamfer_jet_features") or kw.get("get_real_method", "sum") != "sum")
    if not needs_modules:
        try:
            return NativeTrainStep(encoder, decoder, batch_size, **only(NativeTrainStep))
        except NotImplementedError:
            pass
    dropped = sorted(k for k in kw if k not in takes[CapturedModuleStep])
    if dropped:
        warnings.warn(f"native_train_step: this configuration runs as CapturedModuleStep, which does not take {dropped}; ignored")
    return CapturedModuleStep(encoder, decoder, batch_size, **only(CapturedModuleStep))
