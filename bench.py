#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the LGN message-passing hot path on MI355X.

Metric (BASELINE.json): jets/sec, forward + backward, 30-particle jets, maxdim=2, bs=512 per GPU, at
1/2/4/8 MI355X.  One "step" = one pass of the hot path over one batch of synthetic jets already
resident in HBM: encoder -> decoder -> get_real('sum') -> Chamfer + 1e-8 L1 -> backward ->
(gradient all-reduce over RCCL when N > 1) -> Adam, i.e. the inner loop of the reference's
utils/train.py:280-343.  fp64 throughout (the reference's precision).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1|cfg2|cfg4|cfg5] [--batch B | --global-batch G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launch: under ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) each process is one
rank; a plain ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE starts the N ranks ITSELF as fresh child processes
(the parent never touches a GPU) and exits with their worst exit code.

Scaling modes: default = WEAK -- jets are independent graphs, the batch is the unit a GPU works on: every rank runs the config's
batch (bs=512 per GPU at cfg2, the batch a data-parallel training run gives each GPU) and one RCCL all-reduce of the flat gradient
per step; `value` = N x batch x steps / time, `"scaling": "weak"`.  With N > 1 the line also carries ``strong_scaling`` -- BASELINE's
cfg3 read as ONE 512-jet batch split over the N ranks (64 jets per GPU at N = 8: the step is then a chain of 25 launches at their
one-workgroup latency, see ``predicted_strong_scaling`` in the N = 1 line) -- timed in the same run.  ``--strong`` /
``--global-batch G`` make the strong figure the primary one, ``--batch B`` picks another per-GPU batch.  The mode is named in
``scaling`` and in ``config.workload``.  N = 1: both modes are the same run.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline     -- dominant kernel (the fused level BACKWARD of the widest encoder level; its forward twin is reported
                  next to it), timed live with events on the launch stream, priced with SURVEY 8(d)'s algorithmic flops;
                  ``decoder_pairwise`` = the same step with the decoder levels as O(N^2) pair sweeps (SURVEY a-14)
  module_api   -- the same step through the reference's own loop shape on the nn.Module API (enc(batch), dec(latent),
                  torch Chamfer + l1_norm(), loss.backward(), two torch Adams): what a drop-in user gets (N=1 only)
  cpu_baseline -- the oracle (CPU restatement of the reference, materialised like the reference) timed on
                  this box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "lgn-autoencoder_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# (multi-process GPU work on this pool needs dmabuf IPC: the image exports this already; kept here for launchers that scrub the environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# BASELINE.json configs (cfg3 = cfg2 sharded: --global-batch 512 under torch.distributed.run)
CONFIGS = {
    "cfg1": dict(B=32, N=30, ch_enc=(3, 3, 4, 4), ch_dec=(4, 4, 3, 3), maxdim=2,
                 text="cfg1: synthetic 30-particle jets, bs=32, maxdim=2, enc 3-3-4-4 / dec 4-4-3-3"),
    "cfg2": dict(B=512, N=30, ch_enc=(3, 3, 4, 4), ch_dec=(4, 4, 3, 3), maxdim=2,
                 text="cfg2: synthetic 30-particle jets, bs=512, maxdim=2, enc 3-3-4-4 / dec 4-4-3-3"),
    "cfg4": dict(B=256, N=150, ch_enc=(3, 3, 4, 4), ch_dec=(4, 4, 3, 3), maxdim=2,
                 text="cfg4: synthetic 150-particle jets, bs=256, maxdim=2, enc 3-3-4-4 / dec 4-4-3-3"),
    "cfg5": dict(B=512, N=30, ch_enc=(4, 4, 6, 6), ch_dec=(6, 6, 4, 4), maxdim=3,
                 text="cfg5: synthetic 30-particle jets, bs=512, maxdim=3, enc 4-4-6-6 / dec 6-6-4-4"),
}
N_PART, BATCH, CH_ENC, CH_DEC = 30, 512, (3, 3, 4, 4), (4, 4, 3, 3)      # cfg2 (used by the tests' workers)
FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X fp64 vector == fp64 matrix peak (MI355X_MICROARCH.md: FP32 157.3 / 2)
# algorithmic flops per jet of one training step (forward + backward = 3 x forward), decoder levels as pair sweeps
WHOLE_STEP_FLOPS_PER_JET = {"cfg1": 31.2e6, "cfg2": 31.2e6, "cfg4": 533.7e6, "cfg5": 109.9e6}
SETTLE_STEPS = 40                  # untimed steps before the caller's warm-up: the GPU's clocks settle (see _time_steps)
PROFILE_ROUND = "r06"              # profiles/<round>_pmc_<cfg>.json: the PMC passes the `traffic` figures come from


def synthetic_jets(B, N, seed):
    """p3 ~ N(0,1), E = sqrt(|p3|^2 + 1e-6), 'overall_max' normalisation; Nobj ~ U{10..N} zero padded."""
    g = torch.Generator().manual_seed(seed)
    p3 = torch.randn(B, N, 3, dtype=torch.float64, generator=g)
    e = torch.sqrt((p3 * p3).sum(-1, keepdim=True) + 1e-6)
    p4 = torch.cat([e, p3], -1)
    p4 = p4 / (p4.abs().amax(dim=-1, keepdim=True).amax(dim=-2, keepdim=True) + 1e-16)
    nobj = torch.randint(min(10, N), N + 1, (B,), generator=g)
    order = os.environ.get("LGN_BENCH_JET_ORDER")       # experiment: which jets share a CU
    if order and B % 2 == 0:
        idx = torch.argsort(nobj, descending=True, stable=True)
        if order == "pair":       # workgroup i: i-th heaviest jet, workgroup B/2 + i: i-th lightest
            idx = torch.cat([idx[: B // 2], idx[B // 2:].flip(0)])
        elif order == "interleave":   # heaviest, lightest, 2nd heaviest, 2nd lightest, ...
            idx = torch.stack([idx[: B // 2], idx[B // 2:].flip(0)], 1).reshape(-1)
        p4, nobj = p4[idx], nobj[idx]
    labels = (torch.arange(N).unsqueeze(0) < nobj.unsqueeze(1)).to(torch.uint8)
    return p4 * labels.unsqueeze(-1).to(p4.dtype), labels


def level_fwd_flops(N, C, CO, decoder):
    """Algorithmic flops of one fused maxdim=2 level forward per jet, SURVEY 8(d) counting rules."""
    edge = N * N * ((40 + 30 * C) if decoder else (25 + 120 + 160 * C + 30 * C))
    pairs_d1d2 = 4 * 1 + 4 * 4 + 1 * 1 + 1 * 4          # (11,00) (11,11) (00,00) (00,11)
    nnz = 4 + 4 + 1 + 4
    aggregate = N * N * C * 8 * pairs_d1d2 + N * C * 4 * nnz
    power = N * C * (6 * pairs_d1d2 + 4 * nnz)
    catmix = N * 8 * (1 + 4) * CO * 5 * C
    return edge + aggregate + power + catmix


def _events_us(fn, reps, mode_out=None):
    """Average duration of fn's launches: `reps` of them captured into ONE HIP graph (back to back on the device, as in the
    step's graph: eager launches of a 10-50 us kernel add 3-5 us of dispatch gap each and make the host the pacer at 64 jets),
    the replay bracketed by events on the stream it runs on.  If the CAPTURE is refused (a RuntimeError out of the capture; a failing
    launch raises out of fn's own checks first, in the warm-up below, and is not caught) the launches are timed eagerly on a fresh
    stream; mode_out (a dict) receives which of the two happened."""
    for _ in range(3):
        fn()                                 # (launch errors surface here, uncaught)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph, stream=side):       # (its __exit__ ends the capture also when fn raises)
            for _ in range(reps):
                fn()
        run, per, mode = graph.replay, reps, "graph of %d launches" % reps
    except RuntimeError as exc:
        torch.cuda.synchronize()
        side = torch.cuda.Stream()                       # the stream of the aborted capture is not reused
        run, per, mode = fn, 1, "eager launches (capture refused: %s)" % str(exc).splitlines()[0][:120]
    if mode_out is not None:
        mode_out["timing"] = mode
    total, rounds = 0.0, 0
    with torch.cuda.stream(side):
        run()
        for _ in range(3 if per > 1 else 1):
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record(side)
            for _ in range(1 if per > 1 else reps):
                run()
            stop.record(side)
            side.synchronize()
            total, rounds = total + start.elapsed_time(stop) * 1e3 / reps, rounds + 1
    torch.cuda.synchronize()
    return total / rounds


def _time_level(net, decoder, lvl, batch, reps=20):
    """Average duration of the fused forward and backward kernels of maxdim = 2 level `lvl` of a network, through the C ABI with
    preallocated buffers and pre-marshalled arguments (the loop's host time per call stays below the kernel's duration), measured
    with events on the stream they are launched on.  For N <= 40 the backward is ONE kernel (level_bwd3_kernel); its partial-row
    reductions are separate launches and not timed here.  Returns (us_fwd, us_bwd, timing mode)."""
    import ctypes as C
    from lgn import _native as Nn
    Cc, CO = net.num_channels[lvl], net.num_channels[lvl + 1]
    dev = net.device
    B, N = batch["p4"].shape[:2]
    g = torch.Generator(device="cpu").manual_seed(1 + lvl + 10 * int(decoder))
    s = torch.randn(2, B, N, Cc, dtype=torch.float64, generator=g).to(dev)
    v = torch.randn(2, B, N, Cc, 4, dtype=torch.float64, generator=g).to(dev)
    rad = tuple(t.detach().contiguous() for t in net.rad_funcs.rad_funcs[lvl].flat_params())
    mix = net.lgn_cg.node_levels[lvl].cat_mix.mix_reps
    wm0, wm1 = mix.weight((0, 0)).detach().contiguous(), mix.weight((1, 1)).detach().contiguous()
    if decoder:        # complex canonical momenta, every edge masked: only the two Linear biases of the radial network are read
        rad = (None, None, None, None, rad[4], None, rad[6])
        p, mask = torch.randn(2, B, N, 4, dtype=torch.float64, generator=g).to(dev), None
    else:
        p, mask = batch["p4"].to(dev).contiguous(), batch["labels"].to(dev).contiguous()
    L = Nn.lib()
    a, b, c, w0, b0, w1, b1 = rad
    P = Nn.ptr
    dec = int(decoder)
    ag0, ag1, so, vo = Nn.level_fwd(decoder, s, v, p, mask, rad, wm0, wm1)
    fargs = (B, N, Cc, CO, dec, P(s), P(v), P(p), P(mask), P(a), P(b), P(c), P(w0), P(b0), P(w1), P(b1), P(wm0), P(wm1),
             P(ag0), P(ag1), P(so), P(vo))

    def fwd():
        Nn._check(L.lgn_level_fwd_f64(*fargs, Nn.stream_ptr()), "lgn_level_fwd_f64")

    mode = {}
    us_fwd = _events_us(fwd, reps, mode)
    gs = torch.randn(so.shape, dtype=torch.float64, generator=g).to(dev)
    gv = torch.randn(vo.shape, dtype=torch.float64, generator=g).to(dev)
    rm, rr = C.c_int(), C.c_int()
    Nn._check(L.lgn_level_bwd_partial_rows(B, N, dec, C.byref(rm), C.byref(rr)), "lgn_level_bwd_partial_rows")
    part_mix = torch.empty(rm.value, 4 * CO * 5 * Cc, device=dev, dtype=torch.float64)
    part_rad = torch.empty(rr.value, L.lgn_level_rad_partial_len(Cc, dec), device=dev, dtype=torch.float64)
    g_ag = torch.empty(B, N, 20 * Cc, device=dev, dtype=torch.float64)
    g_s_in, g_v_in = torch.empty_like(s), torch.empty_like(v)
    g_p = torch.zeros_like(p) if decoder else None
    bargs = (B, N, Cc, CO, dec, P(s), P(v), P(p), P(mask), P(a), P(b), P(c), P(w0), P(b0), P(w1), P(b1), P(wm0), P(wm1), P(ag0), P(ag1),
             P(gs), P(gv), P(g_ag), P(g_s_in), P(g_v_in), P(g_p), P(part_mix), P(part_rad))

    def bwd():
        Nn._check(L.lgn_level_bwd_f64(*bargs, Nn.stream_ptr()), "lgn_level_bwd_f64")

    us_bwd = _events_us(bwd, reps, mode)
    return us_fwd, us_bwd, mode.get("timing")


def _level_kernel_names(B, N, Cc, decoder):
    """Names of the level kernels a launch of this shape runs (N <= 40), from the library's own launch geometry (lgn_level_jet_split)
    and the switches the per-operator entry points read -- not rebuilt by hand (round 4's label was wrong for C > 4)."""
    from lgn import _native as Nn
    sep = decoder and os.environ.get("LGN_AMD_DEC_PAIRWISE") != "1"
    split = 1 if sep else Nn.lib().lgn_level_jet_split(B, N)
    sym = (not decoder) and Cc <= 4 and split == 1 and os.environ.get("LGN_AMD_BWD_ORDERED") != "1"
    t = lambda x: "true" if x else "false"          # noqa: E731
    return (f"level_fwd2_kernel<{Cc}, {t(decoder)}, {t(sep)}>",
            f"level_bwd3_kernel<{Cc}, {t(decoder)}, {t(sep)}, 4, {t(sym)}>")


def time_dominant_kernel(enc, batch, reps=20):
    """The dominant kernel -- the fused BACKWARD of the widest encoder level, the largest single launch of the step (profiles/) --
    and the matching fused forward (maxdim = 2 levels only)."""
    lvl = max(range(enc.num_cg_levels), key=lambda l: enc.num_channels[l] * enc.num_channels[l + 1])
    Cc, CO = enc.num_channels[lvl], enc.num_channels[lvl + 1]
    B, N = batch["p4"].shape[:2]
    us_fwd, us_bwd, timing = _time_level(enc, False, lvl, batch, reps)
    fwd_flops = B * level_fwd_flops(N, Cc, CO, False)
    kf, kb = _level_kernel_names(B, N, Cc, False)
    single = N <= 40
    return {"kernel": kb if single else "level_bwd (level_bwd_mix + level_bwd_sweep_enc kernels)",
            "level": lvl, "us": us_bwd, "flops": 2 * fwd_flops,          # backward = 2 x forward
            "timing": timing, "forward": {"kernel": kf, "us": us_fwd, "flops": fwd_flops}}


def cgmlp_fwd_flops(M, C, H):
    """CGMLP forward, 2 (2C W + 5 W^2 + W 2C) flops per row."""
    return 2 * M * (2 * C * H + 5 * H * H + H * 2 * C)


def _time_cgmlp(net, lvl, M, reps=20):
    """The CGMLP of level `lvl` at M rows through lgn_cgmlp_fwd/bwd_f64 (kernel only: the partial rows are reduced elsewhere)."""
    from lgn import _native as Nn
    dev = net.device
    mlp = net.lgn_cg.mlp_levels[lvl]
    ws = [l.weight.detach().contiguous() for l in mlp.linear]
    bs = [l.bias.detach().contiguous() for l in mlp.linear]
    C, H = net.num_channels[lvl + 1], ws[0].shape[0]
    g = torch.Generator(device="cpu").manual_seed(3 + lvl)
    s_in = torch.randn(2, M, C, dtype=torch.float64, generator=g).to(dev)
    s_out, g_out, g_in = torch.empty_like(s_in), torch.randn(2, M, C, dtype=torch.float64, generator=g).to(dev), torch.empty_like(s_in)
    L = Nn.lib()
    wp, bp = Nn._ptr_array(ws), Nn._ptr_array(bs)
    psize = sum(w.numel() + b.numel() for w, b in zip(ws, bs))
    part = torch.empty(L.lgn_cgmlp_partial_rows(M, H), psize, device=dev, dtype=torch.float64)
    P = Nn.ptr

    def fwd():
        Nn._check(L.lgn_cgmlp_fwd_f64(M, C, H, len(ws), 0, wp, bp, P(s_in), P(s_out), Nn.stream_ptr()), "lgn_cgmlp_fwd_f64")

    def bwd():
        Nn._check(L.lgn_cgmlp_bwd_f64(M, C, H, len(ws), 0, wp, bp, P(s_in), P(g_out), P(g_in), P(part), psize, Nn.stream_ptr()), "lgn_cgmlp_bwd_f64")

    us_f, us_b = _events_us(fwd, reps), _events_us(bwd, reps)
    chain = H in (12, 24, 36, 48) and H == 12 * C and os.environ.get("LGN_AMD_MLP_V1") != "1"
    one_role = os.environ.get("LGN_AMD_MLP_BWD1") == "1"
    # (csrc/mlp_chain.hip: 64-row workgroups from 8 129 rows on -- eight waves in two roles unless LGN_AMD_MLP_BWD1 --, 16-row ones below)
    name = (("mlp_chain_{}_kernel<%d, %d, ...> (one role per wave)" if one_role else "mlp_chain_{}_kernel<%d, %d, ...> (two roles per workgroup: mlp_chain_bwd2_kernel / mlp_chain_fwd_kernel<..., TWO>)")
            if M >= 8129 else "mlp_chain_{}16_kernel<%d, %d, ...>") % (H, 2 * C) if chain else ("mlp_{}_mfma_kernel (H = %d)" % H)
    return C, H, us_f, us_b, name


def price_step_kernels(enc, dec, batch, ms_per_step):
    """roofline.kernels: every kernel family of the maxdim = 2 step that takes >= 3 % of it, timed live in isolation (graph of 20
    launches, events on the launch stream) and priced with SURVEY 8(d)'s algorithmic flops -- launches per step from the step's own
    structure: every level forward and its CGMLP once, every level backward once, the CGMLP backward of all but each network's
    last level (whose scalars never reach the loss).  Decoder levels run the separable O(N C) form but are priced with the pair
    sweep's flops SURVEY counts (`note`); their `frac` therefore overstates the executed work."""
    B, N = batch["p4"].shape[:2]
    M = B * N
    step_us = ms_per_step * 1e3
    rows = []

    def add(kernel, what, launches, us, flops, note=None):
        ach = flops / (us * 1e-6) / 1e12
        r = {"kernel": kernel, "what": what, "launches_per_step": launches, "us_per_launch": us, "share_of_step": launches * us / step_us,
             "algorithmic_flops_per_launch": flops, "achieved": ach, "frac": ach / FP64_VECTOR_PEAK_TFLOPS}
        if note:
            r["note"] = note
        rows.append(r)

    for net, decoder, tag in ((enc, False, "encoder"), (dec, True, "decoder")):
        L = net.num_cg_levels
        shapes = {}
        for lvl in range(L):
            shapes.setdefault((net.num_channels[lvl], net.num_channels[lvl + 1]), []).append(lvl)
        for (Cc, CO), lvls in shapes.items():
            us_f, us_b, _ = _time_level(net, decoder, lvls[0], batch)
            kf, kb = _level_kernel_names(B, N, Cc, decoder)
            fl = B * level_fwd_flops(N, Cc, CO, decoder)
            note = "separable O(N C) form executed; flops of the reference's pair-sweep formulation" if decoder else None
            add(kf, f"{tag} level forward, C {Cc} -> {CO}", len(lvls), us_f, fl, note)
            add(kb, f"{tag} level backward, C {Cc} -> {CO}", len(lvls), us_b, 2 * fl, note)
        mshapes = {}
        for lvl in range(L):
            mshapes.setdefault(net.num_channels[lvl + 1], []).append(lvl)
        for Cm, lvls in mshapes.items():
            C, H, us_f, us_b, name = _time_cgmlp(net, lvls[0], M)
            nb = sum(1 for l in lvls if l + 1 < L)
            add(name.format("fwd"), f"{tag} CGMLP forward, H = {H}", len(lvls), us_f, cgmlp_fwd_flops(M, C, H))
            if nb:
                add(name.format("bwd"), f"{tag} CGMLP backward, H = {H}", nb, us_b, 2 * cgmlp_fwd_flops(M, C, H),
                    "the hidden activations are recomputed: 3 x the forward's matrix work executed for 2 x counted")
    rows.sort(key=lambda r: -r["share_of_step"])
    return [r for r in rows if r["share_of_step"] >= 0.03], sum(r["share_of_step"] for r in rows)


def local_level_flops(net, lvl):
    """Algorithmic flops per node of the PER-NODE part of a table-driven level, forward (SURVEY 8(d) counting rules): the CG
    matrix of the aggregate 4 nnz per (pair -> irrep) block and channel, the power product 6 d1 d2 per pair + 4 nnz per block,
    CatMix 8 d_r C_out tau_cat(r).  (The N^2 part -- neighbour sums of node (x) edge -- is the moments kernels' work.)"""
    plan, cg = net.plans[lvl], net.cg_dict
    C, CO = plan.channels_in, plan.channels_out
    dim = lambda r: (r[0] + 1) * (r[1] + 1)          # noqa: E731
    per_channel, sq_pairs, catmix = 0, set(), 0
    for r, blocks in plan.cat_blocks.items():
        for src, r1, r2 in blocks:
            if src in ("ag", "sq"):
                per_channel += 4 * int((cg[(r1, r2)][r] != 0).sum().item())
            if src == "sq":
                sq_pairs.add((r1, r2))
        catmix += 8 * dim(r) * CO * C * len(blocks)
    per_channel += sum(6 * dim(r1) * dim(r2) for r1, r2 in sq_pairs)
    return C * per_channel + catmix


def time_dominant_kernel_generic(net, B, N, reps=10):
    """maxdim 3: the largest single launch of the step is the per-node backward of the widest table-driven level
    (local_bwd_static_kernel: d CatMix, d power, d aggregate-CG of 64 nodes x one channel per workgroup).  Timed through its
    C-ABI entry point on buffers in the kernel's tile-blocked layouts; the entry point also launches the weight packing, the
    reduction of the partial rows and their unpacking (~3 short kernels: the kernel alone is the rocprof row under profiles/)."""
    import ctypes as C
    from lgn import _native as Nn
    from lgn.plan import static_kind
    lvl = max(range(net.num_cg_levels), key=lambda l: net.num_channels[l] * net.num_channels[l + 1] * (l > 0))
    tables = net.level_tables(lvl)
    kind = static_kind(tables.meta)
    if not kind:
        return None
    dev = net.device
    Cc, CO, Q, Qo = net.num_channels[lvl], net.num_channels[lvl + 1], tables.meta["Q"], tables.meta["Qout"]
    M = B * N
    tiles = (M + 63) // 64
    g = torch.Generator(device="cpu").manual_seed(2)
    rnd = lambda *shape: torch.randn(*shape, dtype=torch.float64, generator=g).to(dev)      # noqa: E731
    XT, UT, goT = rnd(tiles, Cc, Q, 2, 64), rnd(tiles, Cc, 5 * Q, 2, 64), rnd(tiles, CO, Qo, 2, 64)
    mix = net.lgn_cg.node_levels[lvl].cat_mix.mix_reps
    wcat = torch.cat([mix.weight(r).detach().reshape(-1) for r in tables.meta["out_irreps"]]).contiguous()
    L = Nn.lib()
    w0 = (C.c_int * 5)(*tables.meta["ints"]["out_w0"])
    npk = L.lgn_local_static_packed_doubles(kind, Cc, CO)
    wp, gpk = torch.empty(npk, device=dev, dtype=torch.float64), torch.empty(npk, device=dev, dtype=torch.float64)
    gUT, gXT, part, gw = torch.empty_like(UT), torch.empty_like(XT), torch.empty(tiles, npk, device=dev, dtype=torch.float64), torch.zeros_like(wcat)
    P = Nn.ptr
    args = (kind, M, Cc, CO, P(XT), P(UT), P(wcat), w0, P(wp), P(goT), P(gUT), P(gXT), P(part), P(gpk), P(gw))

    def bwd():
        Nn._check(L.lgn_local_bwd_static_f64(*args, Nn.stream_ptr()), "lgn_local_bwd_static_f64")

    mode = {}
    us = _events_us(bwd, reps, mode)
    return {"kernel": f"local_bwd_static_kernel<Kind{kind}, {4 if CO <= 4 else 6 if CO <= 6 else 8}>", "level": lvl, "us": us, "timing": mode.get("timing"),
            "flops": 2 * M * local_level_flops(net, lvl)}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _lease_cpus(host_cpus):
    """Cores this process may actually use: the affinity mask AND the cgroup CPU quota (the GPU box shows all 256 hardware
    threads of the node in both os.cpu_count() and the affinity mask, while cpu.max grants 16 of them -- 256 OpenMP threads on a
    16-core quota do not finish a step in minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cpus
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, seconds_budget=28.0):
    """Oracle (port of the reference's CPU path, same op sequence and materialised temporaries) on the host cores, on a
    BOUNDED sample of the same workload: a few steps at bs=32 (anomaly detection off, and on as main.py:420 runs it) and,
    for 30-particle configs, one step at the full bs=512 (peak RSS ~7.4 GB).  `value` = the fastest of
    the anomaly-off figures (i.e. the most favourable one for the CPU)."""
    from oracle import lgn_oracle as O
    # the GPU box gives one GPU a share of 16 host cores of the node (cgroup quota); the small sample runs at min(16, lease)
    # threads and, when the lease is larger, at every core of it too (`by_threads`), the rest at whichever was faster
    host_cpus = os.cpu_count() or 1
    lease_cpus = _lease_cpus(host_cpus)
    ncores = min(lease_cpus, 16)
    torch.set_num_threads(ncores)
    N, maxdim = cfg["N"], cfg["maxdim"]
    ce = O.NetConfig(num_particles=N, num_channels=cfg["ch_enc"], maxdim=maxdim)
    cd = O.NetConfig(num_particles=N, num_channels=cfg["ch_dec"], maxdim=maxdim)
    torch.manual_seed(0)
    Pe = {k: v.requires_grad_(True) for k, v in O.init_encoder_params(ce).items()}
    Pd = {k: v.requires_grad_(True) for k, v in O.init_decoder_params(cd, (2, 16)).items()}

    def timed(bs, budget, max_steps, warm):
        p4, labels = synthetic_jets(bs, N, seed=0)

        def one():
            for P in (Pe, Pd):
                for v in P.values():
                    v.grad = None
            loss, _ = O.autoencoder_loss(Pe, Pd, ce, cd, p4, labels, l1_lambda=1e-8)
            loss.backward()
        if warm:
            one()                            # allocator, CG tables
        t0 = time.perf_counter(); n = 0
        while True:
            one(); n += 1
            if time.perf_counter() - t0 > budget or n >= max_steps:
                break
        dt = time.perf_counter() - t0
        return bs * n / dt, n, dt

    small = 32 if N <= 40 else 4            # N=150: bs=16 already needs 5.7 GB and ~13 s per step
    t_all = time.perf_counter()
    r_small, n_small, dt_small = timed(small, seconds_budget * 0.3, 16, warm=True)
    by_threads = {str(ncores): r_small}
    if lease_cpus > ncores:
        torch.set_num_threads(lease_cpus)
        r_all, n_all, dt_all = timed(small, seconds_budget * 0.15, 8, warm=True)
        by_threads[str(lease_cpus)] = r_all
        if r_all > r_small:
            r_small, n_small, dt_small = r_all, n_all, dt_all
        else:
            torch.set_num_threads(ncores)
    with torch.autograd.set_detect_anomaly(True):
        r_anom, n_anom, dt_anom = timed(small, seconds_budget * 0.15, 4, warm=False)
    out = {"unit": "jets/s", "cores": torch.get_num_threads(), "host_cpus": host_cpus, "lease_cpus": lease_cpus,
           "by_threads": by_threads, "cpu_model": _cpu_model(), "kind": "port",
           "anomaly_off": {"value": r_small, "batch": small, "steps": n_small, "seconds": dt_small},
           "anomaly_on": {"value": r_anom, "batch": small, "steps": n_anom, "seconds": dt_anom,
                          "note": "torch.autograd.set_detect_anomaly(True), as main.py:420 / test.py:374 run"}}
    best = r_small
    sample = f"{n_small} fwd+bwd steps of bs={small}"
    if N <= 40 and cfg["B"] >= 512 and time.perf_counter() - t_all < seconds_budget * 0.6:
        r_full, n_full, dt_full = timed(cfg["B"], 0.0, 1, warm=False)
        out["full_batch"] = {"value": r_full, "batch": cfg["B"], "steps": n_full, "seconds": dt_full}
        sample += f" and {n_full} step of bs={cfg['B']}"
        best = max(best, r_full)
    else:
        out["full_batch"] = None
        out["full_batch_note"] = ("not run: at N=150 the reference-style materialised temporaries need ~90 GB at bs=256"
                                  if N > 40 else "not run (time budget)")
    out["value"] = best
    out["sample"] = (f"{sample}, N={N}, maxdim={maxdim}, fp64, oracle/lgn_oracle.py (materialised restatement of the "
                     f"reference CPU path), {time.perf_counter() - t_all:.1f}s in total; value = best anomaly-off rate")
    return out


def _time_steps(trainer, batch, steps, warmup, world):
    """Times `steps` steps.  The native step owns static input buffers (HIP-graph replay): the batch is written into them ONCE,
    before the timed region ("inputs already resident in HBM"), and every step runs on those buffers -- a training loop would
    let its data loader fill `trainer.p4 / trainer.mask` the same way.  The module-API harness takes the batch per step, as
    the reference's loop does."""
    if hasattr(trainer, "load_batch"):
        trainer.load_batch(batch)
        step = trainer.step
    else:
        step = lambda: trainer.step(batch)          # noqa: E731
    # Settling, part of the harness set-up and reported in the JSON line (`settle_steps`): the first ~20 ms of load after the graph capture
    # run at lower clocks (measured at cfg2: --warmup 3 --steps 20 reads 0.550 ms per step, --warmup 30 --steps 20 0.531 ms,
    # --warmup 5 --steps 200 0.531 ms).  The W warm-up steps the caller asked for follow, then exactly K timed steps.
    for _ in range(SETTLE_STEPS):
        step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(loss).item(), "non-finite loss"
    return elapsed


def small_batch_legs(dev, cfg, G, NativeTrainStep):
    """cfg2's network at 256 / 128 / 64 jets -- the per-GPU shares of BASELINE's cfg3 (512 jets over 2 / 4 / 8 GPUs) -- 20 timed steps
    each, as ``configs.b256 / b128 / b64``: the single-process step (lgn_step_train_f64, one graph launch) and, where an `nccl` group
    of ONE rank can be set up on this GPU, the data-parallel branch a rank of an N-GPU job runs ([fwd + bwd | all-reduce | L1 + Adam]
    in one graph; the all-reduce of a single rank moves no data: its launch is in the figure, the exchange is not)."""
    import socket
    legs = {}
    own_group = False
    if dist.is_available() and not dist.is_initialized():
        try:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            own_group = True
        except Exception as exc:      # noqa: BLE001  (a secondary figure: reported, never fatal)
            legs["dp_branch_error"] = f"{type(exc).__name__}: {exc}"[:300]
    try:
        for jets in (256, 128, 64):
            name = f"b{jets}"
            try:
                p4, labels = synthetic_jets(jets, cfg["N"], seed=0)
                b = {"p4": p4.to(dev), "labels": labels.to(dev)}
                e2, d2 = G._models(cfg["N"], cfg["ch_enc"], cfg["ch_dec"], dev, seed=0, maxdim=cfg["maxdim"])
                t2 = NativeTrainStep(e2, d2, batch_size=jets, lr=5e-4, l1_lambda=1e-8, use_graph=True)
                el = _time_steps(t2, b, 20, 5, 1)
                legs[name] = {"value": jets * 20 / el, "unit": "jets/s", "ms_per_step": 1e3 * el / 20, "steps": 20, "warmup": 5,
                              "workload": f"cfg2's networks, {jets} jets of 30 particles on one GPU (cfg3's share on {cfg['B'] // jets} GPUs)"}
                del t2, e2, d2
                if own_group:
                    e2, d2 = G._models(cfg["N"], cfg["ch_enc"], cfg["ch_dec"], dev, seed=0, maxdim=cfg["maxdim"])
                    t2 = NativeTrainStep(e2, d2, batch_size=jets, lr=5e-4, l1_lambda=1e-8, use_graph=True, force_collective=True)
                    el = _time_steps(t2, b, 20, 5, 1)
                    legs[name]["dp_branch"] = {"ms_per_step": 1e3 * el / 20, "graph_launches_per_step": t2.launches_per_step,
                                               "note": "lgn_step_fwd_bwd_f64 | RCCL all-reduce (world size 1) | lgn_step_finalize_f64"}
                    del t2, e2, d2
            except RuntimeError as exc:
                legs[name] = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:300]}
    finally:
        if own_group:
            dist.destroy_process_group()
    return legs


# all-reduce(SUM) of the flat gradient buffer (63.5 k parameters + the per-jet loss terms = 512 KB) over xGMI: latency-bound at
# this size; NOT measured here (one GPU per box) -- an assumption, stated in the line
ASSUMED_ALLREDUCE_US = 25.0


def predict_strong_scaling(global_batch, one_gpu_rate, configs):
    """BASELINE's cfg3 (cfg2's 512 jets split over N GPUs) PREDICTED from the one-GPU small-batch steps: a rank's step is the
    data-parallel branch at 512 / N jets (configs.bNNN.dp_branch, else the single-process step) plus an all-reduce of 512 KB.
    The driver's SCALE run is the measurement; this object says what the builder expects it to read and why."""
    pred = {"global_batch": global_batch, "assumed_allreduce_us": ASSUMED_ALLREDUCE_US, "one_gpu_jets_per_s": one_gpu_rate,
            "note": "prediction, not a measurement: per-rank step time measured on ONE GPU at the per-GPU share of the batch + an assumed "
                    "latency-bound all-reduce; a 64-jet step is 25 dependent launches at their one-workgroup latency, so strong scaling of "
                    "a 512-jet batch is bounded by it (>= 6x at 8 GPUs would need <= 87 us per 64-jet step incl. the all-reduce)"}
    for n in (2, 4, 8):
        leg = configs.get(f"b{global_batch // n}") or {}
        ms = (leg.get("dp_branch") or {}).get("ms_per_step") or leg.get("ms_per_step")
        if not ms:
            pred[f"gpus_{n}"] = None
            continue
        t = ms + ASSUMED_ALLREDUCE_US * 1e-3
        pred[f"gpus_{n}"] = {"jets_per_gpu": global_batch // n, "ms_per_step": t, "value": global_batch / (t * 1e-3),
                             "speedup_vs_one_gpu": global_batch / (t * 1e-3) / one_gpu_rate}
    return pred


def _spawn_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them), rank 0 inherits stdout and prints the
    JSON line.  The parent touches no GPU.  Returns the worst exit code; if a rank fails the others are terminated (by PID)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst, live = 0, list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in live:              # a failed rank leaves the others waiting in a collective
                    q.terminate()
        time.sleep(0.05)
    return worst


def _dry_run(args, world, rank):
    """LGN_BENCH_DRY=1 (CPU tests of the launch plumbing): gloo rendezvous, the same barrier / max-over-ranks timing
    shape as the real run around a trivial CPU step, rank 0 prints the line marked ``"dry_run": true``."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.ones(4, dtype=torch.float64) * (rank + 1)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = x.clone()
        if world > 1:
            dist.all_reduce(y, op=dist.ReduceOp.SUM)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "sum": float(y[0]),
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "elapsed": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2", help="BASELINE.json configuration (default cfg2)")
    ap.add_argument("--batch", type=int, default=None, help="jets per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total jets per step, split over the ranks (strong scaling; default: the config's batch, so that "
                         "--gpus 8 on cfg2 is BASELINE's cfg3)")
    ap.add_argument("--weak", action="store_true", help="(the default) weak scaling: the config's batch on every GPU")
    ap.add_argument("--strong", action="store_true", help="strong scaling as the primary figure: the config's batch split over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the module_api and decoder_pairwise legs")
    ap.add_argument("--harness", choices=["native", "native-nograph", "module", "modular", "captured"], default=None,
                    help="native: one C call per step replayed from a HIP graph (default); module: the reference's loop on the "
                         "nn.Module API")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process only starts the ranks (it never initialises a GPU) and passes their exit code on
        raise SystemExit(_spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if sum(x is not None for x in (args.batch, args.global_batch)) + int(args.weak) + int(args.strong) > 1:
        raise SystemExit("--batch, --global-batch, --weak and --strong are exclusive")
    if os.environ.get("LGN_BENCH_DRY") == "1":
        return _dry_run(args, world, rank)
    # ONE line on stdout, whatever the libraries underneath print there (RCCL writes a version banner to fd 1 when a communicator is
    # created): fd 1 points at stderr for the whole run, the JSON line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if not args.strong and args.global_batch is None:
        per_gpu, scaling = (args.batch or cfg["B"]), "weak"
    else:
        gb = args.global_batch if args.global_batch is not None else cfg["B"]
        if gb % world:
            raise SystemExit(f"global batch {gb} is not divisible by {world} ranks")
        per_gpu, scaling = gb // world, "strong"
    N = cfg["N"]
    harness = args.harness or "native"
    harness = "module" if harness == "modular" else harness

    import __graft_entry__ as G
    from lgn.step import CapturedModuleStep, NativeTrainStep, ReferenceLoopStep

    def build(which, jets=None):
        enc, dec = G._models(N, cfg["ch_enc"], cfg["ch_dec"], dev, seed=0, maxdim=cfg["maxdim"])   # identical replicas on every rank
        if which == "module":
            return enc, ReferenceLoopStep(enc, dec, lr=5e-4, l1_lambda=1e-8)
        if which == "captured":
            return enc, CapturedModuleStep(enc, dec, batch_size=jets or per_gpu, lr=5e-4, l1_lambda=1e-8)
        return enc, NativeTrainStep(enc, dec, batch_size=jets or per_gpu, lr=5e-4, l1_lambda=1e-8, use_graph=which == "native")

    def timed(tr, b):
        e = _time_steps(tr, b, args.steps, args.warmup, world)
        if world > 1:                                      # MAX over ranks
            t = torch.tensor([e], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        return e

    enc, trainer = build(harness)
    p4, labels = synthetic_jets(per_gpu, N, seed=rank)     # per-rank shard, resident in HBM
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    elapsed = timed(trainer, batch)
    other = None
    if world > 1 and harness != "module" and args.batch is None and args.global_batch is None and cfg["B"] % world == 0:
        # the OTHER scaling figure next to the primary one, timed in the same run (all ranks take part): strong = the config's batch
        # split over the ranks, weak = the config's batch on every rank
        del trainer
        jets = cfg["B"] // world if scaling == "weak" else cfg["B"]
        _, tw = build(harness, jets)
        pw, lw = synthetic_jets(jets, N, seed=rank)
        ew = timed(tw, {"p4": pw.to(dev), "labels": lw.to(dev)})
        other = {"value": jets * world * args.steps / ew, "unit": "jets/s", "ms_per_step": 1e3 * ew / args.steps,
                 "jets_per_gpu": jets, "global_batch": jets * world, "scaling": "strong" if scaling == "weak" else "weak"}
        del tw
        trainer = None

    if rank == 0:
        mode = (f"weak scaling: {per_gpu} jets on each of {world} GPU(s)" if scaling == "weak" else
                f"strong scaling: {per_gpu * world} jets per step split over {world} GPU(s) = {per_gpu} per GPU")
        out = {
            "metric": "jets/sec fwd+bwd, 30-particle maxdim=2 bs=512" if args.config == "cfg2" else f"jets/sec fwd+bwd, {args.config}",
            "value": per_gpu * world * args.steps / elapsed,
            "unit": "jets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfg['text']}, tau-latent 1s/8v, min&max, chamfer + 1e-8 L1, fwd+bwd+Adam, zero-padded "
                                   f"Nobj~U{{10..{N}}}, batch resident in HBM (written into the step's input buffers before the "
                                   f"timed region); {mode}",
                       "name": args.config, "jets_per_gpu": per_gpu, "global_batch": per_gpu * world, "particles": N,
                       "maxdim": cfg["maxdim"],
                       "parallelism": f"dp{world}" + (" (one RCCL all-reduce of the flat gradient per step)" if world > 1 else ""),
                       "harness": {"native": "NativeTrainStep: lgn_step_train_f64 (one process) or lgn_step_fwd_bwd_f64 | all-reduce | "
                                             "lgn_step_finalize_f64 (data parallel), replayed from a HIP graph",
                                   "native-nograph": "NativeTrainStep without graph capture",
                                   "module": "ReferenceLoopStep: reference loop on the nn.Module API",
                                   "captured": "CapturedModuleStep: module API + ChamferLoss under autograd + native L1 / Adam, "
                                               "captured into one HIP graph (what configurations outside the whole-step call get)"}[harness]},
        }
        if other is not None:
            out[other["scaling"] + "_scaling"] = other
        if cfg["maxdim"] == 2:
            dom = time_dominant_kernel(enc, batch)
            achieved = dom["flops"] / (dom["us"] * 1e-6) / 1e12
            traffic, traffic_src = None, None   # HBM bytes per launch of the dominant kernel, from the PMC passes recorded under profiles/
            if args.config == "cfg2" and per_gpu == 512:
                traffic_src = f"profiles/{PROFILE_ROUND}_pmc_cfg2.json"
                try:
                    with open(os.path.join(ROOT, traffic_src)) as fh:
                        traffic = json.load(fh)["kernels"]["lgn::" + dom["kernel"]]["derived"]["hbm_bytes"]
                except (OSError, KeyError) as exc:
                    print(f"bench.py: no PMC record of {dom['kernel']} in {traffic_src} ({type(exc).__name__}: {exc}); traffic = null",
                          file=sys.stderr)
            fw = dom["forward"]
            step_flops = WHOLE_STEP_FLOPS_PER_JET[args.config]
            ach_step = out["value"] * step_flops / 1e12
            out["roofline"] = {
                "bound": "mfma", "pipe": "fp64 datapath: v_fma_f64 and v_mfma_f64 share it on MI355X (measured, "
                                         "csrc/probes/mfma_rate_probe.hip: an MFMA wave and an FMA wave on one SIMD take the SUM "
                                         "of their times) -- one 78.6 TFLOP/s budget for the kernel's vector and matrix flops",
                "kernel": dom["kernel"], "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                "traffic_note": "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction; an ESTIMATE: the guide "
                                "calibrates the x2 on 16 B/lane loads, this kernel issues 8 B/lane), separate rocprofv3 "
                                f"--pmc passes recorded in {traffic_src}; HBM is not the bound",
                "us_per_launch": dom["us"], "timing": dom.get("timing"), "algorithmic_flops_per_launch": dom["flops"],
                "forward_kernel": {"kernel": fw["kernel"], "us_per_launch": fw["us"], "algorithmic_flops_per_launch": fw["flops"],
                                   "achieved": fw["flops"] / (fw["us"] * 1e-6) / 1e12,
                                   "frac": fw["flops"] / (fw["us"] * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS},
                "note": "the decoder levels run the separable O(N C) form (SURVEY a-14: an algorithmic change, "
                        "not counted as roofline gain); this kernel is an encoder level and is unaffected",
                "whole_step": {"achieved": ach_step, "frac": ach_step / FP64_VECTOR_PEAK_TFLOPS, "algorithmic_flops_per_jet": step_flops,
                               "note": "SURVEY 8(d) algorithmic flops per jet (decoder counted as pair sweeps, which the default "
                                       "step does not execute) x measured jets/s; the honest figure is decoder_pairwise.whole_step"}}
            if N <= 40 and world == 1 and getattr(trainer, "decoder", None) is not None:
                ks, covered = price_step_kernels(enc, trainer.decoder, batch, out["ms_per_step"])
                out["roofline"]["kernels"] = ks
                out["roofline"]["kernels_note"] = (f"every kernel family >= 3 % of the step, timed in isolation (graph of 20 launches) and "
                                                   f"priced with SURVEY 8(d) flops; level + CGMLP kernels together = {covered:.2f} of the step.  "
                                                   f"Not a flops kernel and not in the list: the step's tail (step_tail_kernel: every deferred "
                                                   f"reduction -- ~107 MB of partial rows at 512 jets, HBM-bound -- radial finalisation, L1 + Adam, "
                                                   f"loss: one launch, ~5 % of the step; its duration is in profiles/{PROFILE_ROUND}_cfg2_kernel_stats.csv)")
        if cfg["maxdim"] != 2:
            # table-driven levels: the dominant launch (per-node backward of the widest level) priced with SURVEY 8(d)'s counting
            # rules, and the whole-step algorithmic rate (cfg5 fwd+bwd = 109.9 MFLOP per jet) next to it
            flops_per_jet = 109.9e6
            ach_step = out["value"] * flops_per_jet / 1e12
            dom = time_dominant_kernel_generic(enc, per_gpu, N)
            traffic = None
            if args.config == "cfg5" and per_gpu == 512 and dom is not None:
                try:
                    with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_cfg5.json")) as fh:
                        ks = json.load(fh)["kernels"]
                    traffic = ks["lgn::" + dom["kernel"].replace("<Kind", "<lgn::cgs::Kind")]["derived"].get("hbm_bytes")
                except (OSError, KeyError) as exc:
                    print(f"bench.py: no PMC record of {dom['kernel']} in profiles/{PROFILE_ROUND}_pmc_cfg5.json ({type(exc).__name__}: "
                          f"{exc}); traffic = null", file=sys.stderr)
            if dom is not None:
                ach = dom["flops"] / (dom["us"] * 1e-6) / 1e12
                out["roofline"] = {"bound": "mfma", "pipe": "fp64 vector datapath (no matrix instructions in this kernel; schema has hbm|mfma only)",
                                   "kernel": dom["kernel"], "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                                   "traffic_note": f"2 x FETCH_SIZE + WRITE_SIZE of the rocprofv3 --pmc passes in profiles/{PROFILE_ROUND}_pmc_cfg5.json",
                                   "us_per_launch": dom["us"], "algorithmic_flops_per_launch": dom["flops"],
                                   "note": "us_per_launch is the C-ABI call (kernel + weight packing + partial-row reduction + unpacking); "
                                           f"the kernel alone: profiles/{PROFILE_ROUND}_cfg5_kernel_stats.csv",
                                   "whole_step": {"achieved": ach_step, "frac": ach_step / FP64_VECTOR_PEAK_TFLOPS,
                                                  "algorithmic_flops_per_jet": flops_per_jet,
                                                  "note": "SURVEY 8(d) algorithmic flops per jet x measured jets/s"}}
            else:
                out["roofline"] = {"bound": "mfma", "kernel": "whole step (table-driven levels)", "achieved": ach_step,
                                   "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_step / FP64_VECTOR_PEAK_TFLOPS,
                                   "traffic": None, "algorithmic_flops_per_jet": flops_per_jet}
        if world == 1 and not args.no_extras:
            if harness != "module":
                del trainer
                _, mod = build("module")
                e2 = _time_steps(mod, batch, args.steps, args.warmup, 1)
                out["module_api"] = {"value": per_gpu * args.steps / e2, "unit": "jets/s", "ms_per_step": 1e3 * e2 / args.steps,
                                     "harness": "ReferenceLoopStep (lgn/step.py): enc(batch) -> dec(latent) -> lgn.losses.ChamferLoss + "
                                                "l1_norm() -> loss.backward() -> 2 x torch.optim.Adam, fused whole-network "
                                                "native calls under autograd, no graph capture"}
                del mod
                try:        # a secondary figure: whatever goes wrong here is reported, the contract line above stands
                    _, cap = build("captured")
                    e3 = _time_steps(cap, batch, args.steps, args.warmup, 1)
                    out["module_api_captured"] = {"value": per_gpu * args.steps / e3, "unit": "jets/s", "ms_per_step": 1e3 * e3 / args.steps,
                                                  "harness": "CapturedModuleStep (lgn/step.py): the same module-API step (ChamferLoss, "
                                                             "backward(), native L1 + Adam) captured into one HIP graph -- the route of "
                                                             "configurations the whole-step call does not take (jet_features, extra scalars, ...)"}
                    del cap
                except RuntimeError as exc:
                    out["module_api_captured"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:300]}
            if harness == "native" and cfg["maxdim"] == 2:
                os.environ["LGN_AMD_DEC_PAIRWISE"] = "1"
                try:
                    _, pw = build("native")
                    e3 = _time_steps(pw, batch, args.steps, args.warmup, 1)
                    pw_rate = per_gpu * args.steps / e3
                    out["roofline"]["decoder_pairwise"] = {
                        "value": pw_rate, "unit": "jets/s", "ms_per_step": 1e3 * e3 / args.steps,
                        "whole_step": {"achieved": pw_rate * WHOLE_STEP_FLOPS_PER_JET[args.config] / 1e12,
                                       "frac": pw_rate * WHOLE_STEP_FLOPS_PER_JET[args.config] / 1e12 / FP64_VECTOR_PEAK_TFLOPS},
                        "note": "same step with LGN_AMD_DEC_PAIRWISE=1: decoder levels as O(N^2) pair sweeps, the reference's "
                                "formulation whose flops SURVEY 8(d) counts"}
                    del pw
                finally:
                    del os.environ["LGN_AMD_DEC_PAIRWISE"]
        if world == 1 and not args.no_extras and args.config == "cfg2" and harness == "native" and per_gpu == cfg["B"]:
            # the other single-GPU BASELINE configs, 20 timed steps each of the same native step (< 0.2 s of GPU time): driver-timed
            # figures for cfg4 / cfg5 next to the headline (python bench.py --config cfgN gives the full line of each)
            out["configs"] = {}
            for name in ("cfg4", "cfg5"):
                c2 = CONFIGS[name]
                try:
                    e2, d2 = G._models(c2["N"], c2["ch_enc"], c2["ch_dec"], dev, seed=0, maxdim=c2["maxdim"])
                    t2 = NativeTrainStep(e2, d2, batch_size=c2["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True)
                    q4, l4 = synthetic_jets(c2["B"], c2["N"], seed=0)
                    el = _time_steps(t2, {"p4": q4.to(dev), "labels": l4.to(dev)}, 20, 5, 1)
                    rate = c2["B"] * 20 / el
                    ach = rate * WHOLE_STEP_FLOPS_PER_JET[name] / 1e12
                    out["configs"][name] = {"value": rate, "unit": "jets/s", "ms_per_step": 1e3 * el / 20, "steps": 20, "warmup": 5,
                                            "workload": c2["text"], "whole_step": {"achieved": ach, "frac": ach / FP64_VECTOR_PEAK_TFLOPS,
                                                                                   "algorithmic_flops_per_jet": WHOLE_STEP_FLOPS_PER_JET[name]}}
                    del t2, e2, d2
                except RuntimeError as exc:
                    out["configs"][name] = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:300]}
        if world == 1 and not args.no_extras and args.config == "cfg2" and harness == "native" and per_gpu == cfg["B"]:
            out["configs"].update(small_batch_legs(dev, cfg, G, NativeTrainStep))
            out["predicted_strong_scaling"] = predict_strong_scaling(cfg["B"], out["value"], out["configs"])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
