#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the LGN message-passing hot path on MI355X.

Metric (BASELINE.json): jets/sec, forward + backward, 30-particle jets, maxdim=2, bs=512 per GPU, at
1/2/4/8 MI355X.  One "step" = one pass of the hot path over one batch of synthetic jets already
resident in HBM: encoder -> decoder -> get_real('sum') -> Chamfer + 1e-8 L1 -> backward ->
(gradient all-reduce over RCCL when N > 1) -> Adam, i.e. the inner loop of the reference's
utils/train.py:280-343.  fp64 throughout (the reference's precision).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (the fused level BACKWARD of the widest encoder level; its forward twin is reported
                  next to it), timed live with events on the launch stream, priced with SURVEY 8(d)'s algorithmic flops
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

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_PART = 30
BATCH = 512
CH_ENC = (3, 3, 4, 4)
CH_DEC = (4, 4, 3, 3)
FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X fp64 vector == fp64 matrix peak (MI355X_MICROARCH.md: FP32 157.3 / 2)


def synthetic_jets(B, N, seed):
    """p3 ~ N(0,1), E = sqrt(|p3|^2 + 1e-6), 'overall_max' normalisation; Nobj ~ U{10..N} zero padded."""
    g = torch.Generator().manual_seed(seed)
    p3 = torch.randn(B, N, 3, dtype=torch.float64, generator=g)
    e = torch.sqrt((p3 * p3).sum(-1, keepdim=True) + 1e-6)
    p4 = torch.cat([e, p3], -1)
    p4 = p4 / (p4.abs().amax(dim=-1, keepdim=True).amax(dim=-2, keepdim=True) + 1e-16)
    nobj = torch.randint(min(10, N), N + 1, (B,), generator=g)
    labels = (torch.arange(N).unsqueeze(0) < nobj.unsqueeze(1)).to(torch.uint8)
    return p4 * labels.unsqueeze(-1).to(p4.dtype), labels


def level_fwd_flops(N, C, CO, decoder):
    """Algorithmic flops of one fused level forward per jet, SURVEY 8(d) counting rules."""
    edge = N * N * ((40 + 30 * C) if decoder else (25 + 120 + 160 * C + 30 * C))
    pairs_d1d2 = 4 * 1 + 4 * 4 + 1 * 1 + 1 * 4          # (11,00) (11,11) (00,00) (00,11)
    nnz = 4 + 4 + 1 + 4
    aggregate = N * N * C * 8 * pairs_d1d2 + N * C * 4 * nnz
    power = N * C * (6 * pairs_d1d2 + 4 * nnz)
    catmix = N * 8 * (1 + 4) * CO * 5 * C
    return edge + aggregate + power + catmix


def time_dominant_kernel(enc, batch, reps=20):
    """Average duration of the dominant kernel -- the fused BACKWARD of the widest encoder level, the largest single
    launch of the step (profiles/) -- and of the matching fused forward, measured with events on the stream they are
    launched on (torch's current stream).  Both go through the C ABI with preallocated buffers; for N <= 40 the
    backward is ONE kernel (level_bwd3_kernel), its partial-row reductions are separate launches and not timed here."""
    import ctypes as C
    from lgn import _native as Nn
    lvl = max(range(enc.num_cg_levels), key=lambda l: enc.num_channels[l] * enc.num_channels[l + 1])
    Cc, CO = enc.num_channels[lvl], enc.num_channels[lvl + 1]
    dev = enc.device
    B, N = batch["p4"].shape[:2]
    g = torch.Generator(device="cpu").manual_seed(1)
    s = torch.randn(2, B, N, Cc, dtype=torch.float64, generator=g).to(dev)
    v = torch.randn(2, B, N, Cc, 4, dtype=torch.float64, generator=g).to(dev)
    rad = tuple(t.detach().contiguous() for t in enc.rad_funcs.rad_funcs[lvl].flat_params())
    mix = enc.lgn_cg.node_levels[lvl].cat_mix.mix_reps
    wm0, wm1 = mix.weight((0, 0)).detach().contiguous(), mix.weight((1, 1)).detach().contiguous()
    p = batch["p4"].to(dev).contiguous()
    mask = batch["labels"].to(dev).contiguous()

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            fn()
        stop.record()
        torch.cuda.synchronize()
        return start.elapsed_time(stop) * 1e3 / reps

    # forward (the loop also allocates 4 output tensors per call from torch's caching allocator: no device sync)
    us_fwd = timed(lambda: Nn.level_fwd(False, s, v, p, mask, rad, wm0, wm1))
    ag0, ag1, so, vo = Nn.level_fwd(False, s, v, p, mask, rad, wm0, wm1)
    gs = torch.randn(so.shape, dtype=torch.float64, generator=g).to(dev)
    gv = torch.randn(vo.shape, dtype=torch.float64, generator=g).to(dev)
    L = Nn.lib()
    rm, rr = C.c_int(), C.c_int()
    Nn._check(L.lgn_level_bwd_partial_rows(B, N, 0, C.byref(rm), C.byref(rr)), "lgn_level_bwd_partial_rows")
    part_mix = torch.empty(rm.value, 4 * CO * 5 * Cc, device=dev, dtype=torch.float64)
    part_rad = torch.empty(rr.value, L.lgn_level_rad_partial_len(Cc, 0), device=dev, dtype=torch.float64)
    g_ag = torch.empty(B, N, 20 * Cc, device=dev, dtype=torch.float64)
    g_s_in, g_v_in = torch.empty_like(s), torch.empty_like(v)
    a, b, c, w0, b0, w1, b1 = rad
    P = Nn.ptr

    def bwd():
        rc = L.lgn_level_bwd_f64(B, N, Cc, CO, 0, P(s), P(v), P(p), P(mask), P(a), P(b), P(c), P(w0), P(b0), P(w1), P(b1),
                                 P(wm0), P(wm1), P(ag0), P(ag1), P(gs), P(gv), P(g_ag), P(g_s_in), P(g_v_in), P(None),
                                 P(part_mix), P(part_rad), Nn.stream_ptr())
        Nn._check(rc, "lgn_level_bwd_f64")

    us_bwd = timed(bwd)
    fwd_flops = B * level_fwd_flops(N, Cc, CO, False)
    single = N <= 40
    return {"kernel": f"level_bwd3_kernel<{Cc}, false, false>" if single else "level_bwd (nodes2 + rad2 + mix kernels)",
            "level": lvl, "us": us_bwd, "flops": 2 * fwd_flops,          # backward = 2 x forward
            "forward": {"kernel": f"level_fwd2_kernel<{Cc}, false, false>", "us": us_fwd, "flops": fwd_flops}}


def cpu_baseline(seconds_budget=25.0):
    """Oracle (port of the reference's CPU path) on the host cores, bounded sample of the same workload."""
    from oracle import lgn_oracle as O
    # the GPU box gives one GPU a share of 16 host cores; more threads only add contention on these small ops
    ncores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(ncores)
    ce = O.NetConfig(num_particles=N_PART, num_channels=CH_ENC)
    cd = O.NetConfig(num_particles=N_PART, num_channels=CH_DEC)
    torch.manual_seed(0)
    Pe = {k: v.requires_grad_(True) for k, v in O.init_encoder_params(ce).items()}
    Pd = {k: v.requires_grad_(True) for k, v in O.init_decoder_params(cd, (2, 16)).items()}
    bs = 32
    p4, labels = synthetic_jets(bs, N_PART, seed=0)

    def one():
        for P in (Pe, Pd):
            for v in P.values():
                v.grad = None
        loss, _ = O.autoencoder_loss(Pe, Pd, ce, cd, p4, labels, l1_lambda=1e-8)
        loss.backward()

    tw = time.perf_counter()
    one()                                   # warm-up (allocator, CG tables)
    tw = time.perf_counter() - tw
    t0 = time.perf_counter(); n = 0
    while True:
        one(); n += 1
        if time.perf_counter() - t0 + tw > seconds_budget or n >= 16:
            break
    dt = time.perf_counter() - t0
    return {"value": bs * n / dt, "unit": "jets/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} fwd+bwd steps of bs={bs}, N={N_PART}, maxdim=2, fp64, oracle/lgn_oracle.py (materialised "
                      f"restatement of the reference CPU path), {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="jets per GPU (default: BASELINE cfg2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--harness", choices=["native", "native-nograph", "modular"], default="native",
                    help="native: one C call per step replayed from a HIP graph (default); modular: nn.Module + autograd path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    enc, dec = G._models(N_PART, CH_ENC, CH_DEC, dev, seed=0)      # identical replicas on every rank
    if args.harness == "modular":
        trainer = TrainStep(enc, dec, lr=5e-4, l1_lambda=1e-8)
    else:
        trainer = NativeTrainStep(enc, dec, batch_size=args.batch, lr=5e-4, l1_lambda=1e-8,
                                  use_graph=args.harness == "native")
    p4, labels = synthetic_jets(args.batch, N_PART, seed=rank)     # per-rank shard, resident in HBM
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}

    for _ in range(args.warmup):
        trainer.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss).item(), "non-finite loss"

    if rank == 0:
        dom = time_dominant_kernel(enc, batch)
        achieved = dom["flops"] / (dom["us"] * 1e-6) / 1e12
        traffic = None          # HBM bytes per launch of the dominant kernel, from the PMC passes recorded under profiles/
        try:
            with open(os.path.join(ROOT, "profiles", "r01_v8_traffic.json")) as fh:
                traffic = json.load(fh).get(dom["kernel"], {}).get("hbm_bytes_per_launch")
        except OSError:
            pass
        out = {
            "metric": "jets/sec fwd+bwd, 30-particle maxdim=2 bs=512",
            "value": args.batch * world * args.steps / elapsed,
            "unit": "jets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2: synthetic 30-particle jets, maxdim=2, enc 3-3-4-4 / dec 4-4-3-3, tau-latent 1s/8v, "
                                   "min&max, chamfer + 1e-8 L1, fwd+bwd+Adam, zero-padded Nobj~U{10..30}",
                       "jets_per_gpu": args.batch, "global_batch": args.batch * world, "particles": N_PART,
                       "parallelism": f"dp{world}" + (" (one RCCL all-reduce of the flat gradient per step)" if world > 1 else ""),
                       "harness": args.harness},
            "roofline": {"bound": "mfma", "pipe": "fp64 vector ALU (same peak as fp64 MFMA on MI355X; the kernel is "
                                                   "FMA-bound, neither HBM- nor matrix-core-bound)",
                         "kernel": dom["kernel"], "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction), separate rocprofv3 "
                                         "--pmc passes recorded in profiles/r01_v8_traffic.json; 0.5 TB/s, HBM is not the bound",
                         "us_per_launch": dom["us"], "algorithmic_flops_per_launch": dom["flops"],
                         "forward_kernel": {"kernel": dom["forward"]["kernel"], "us_per_launch": dom["forward"]["us"],
                                            "algorithmic_flops_per_launch": dom["forward"]["flops"],
                                            "achieved": dom["forward"]["flops"] / (dom["forward"]["us"] * 1e-6) / 1e12,
                                            "frac": dom["forward"]["flops"] / (dom["forward"]["us"] * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS},
                         "note": "the decoder levels run the separable O(N C) form (SURVEY a-14: an algorithmic change, "
                                 "not counted as roofline gain); this kernel is an encoder level and is unaffected"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
