"""GPU parity tests (run with -m gpu on an MI355X): every native entry point is called through the
C ABI (ctypes -> liblgn_amd.so) and compared with the oracle on the same seeded inputs, and the full
networks are compared with the golden vectors captured from the reference.
Tolerances: fp64 arithmetic, relative to max|ref|; 1e-11 forward, 1e-9 gradients (sums over up to
B*N*N = 460k terms in a different order than the CPU BLAS)."""
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-11
GRAD_TOL = 1e-9


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def O():
    from oracle import lgn_oracle
    return lgn_oracle


def _rand_level_params(O, C, CO, decoder, g):
    cfg = O.NetConfig(num_channels=(C, CO))
    P = {}
    torch.manual_seed(int(torch.randint(0, 10000, (1,), generator=g)))
    O._init_radial(P, cfg, decoder)
    tau0 = {(0, 0): C, (1, 1): C}
    plans = O.build_level_plans(cfg, tau0)
    O._init_levels(P, cfg, plans)
    # make every parameter O(1) so that all gradient paths are exercised
    for k in P:
        if "cat_mix" in k:
            P[k] = torch.randn(P[k].shape, dtype=torch.float64, generator=g) * 0.3
    return cfg, plans, P


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("C,CO,N,B", [(3, 3, 30, 3), (3, 4, 30, 2), (4, 4, 30, 2), (4, 3, 30, 2), (4, 4, 7, 2),
                                      (2, 5, 33, 2), (4, 4, 70, 1), (1, 1, 1, 1),
                                      (4, 4, 48, 2), (3, 4, 50, 1), (4, 3, 41, 2), (2, 2, 63, 1),    # 40 < N < 64
                                      (4, 4, 150, 1), (3, 3, 150, 2), (5, 6, 100, 1), (8, 8, 70, 1),   # chunked receivers
                                      (2, 3, 270, 1), (4, 4, 30, 66)])   # N > 256 (symmetric sweep off: > 64 groups); jet split 4
def test_level_fwd_bwd(dev, O, decoder, C, CO, N, B):
    """N <= 40: one-kernel backward (level_bwd3); 40 < N: mix + ONE pair sweep for node gradients and radial sums
    (level_bwd_sweep_enc: four waves per jet, receivers in chunks when the jet's g_ag does not fit beside a second workgroup)
    resp. mix + separable decoder backward; N >= 64 with a small batch: the 8-wave 'wide' sweeps."""
    _level_case(dev, O, decoder, C, CO, N, B)


@pytest.mark.parametrize("ordered", [False, True])
@pytest.mark.parametrize("C,CO,N,B", [(4, 4, 30, 260), (3, 4, 30, 260), (3, 3, 33, 260), (2, 5, 40, 257), (4, 3, 7, 300), (1, 2, 1, 513),
                                      (5, 6, 26, 260), (8, 8, 13, 260)])
def test_level_backward_whole_jet_workgroups(dev, O, monkeypatch, ordered, C, CO, N, B):
    """Batches of more than 256 jets give every jet ONE workgroup, and the encoder's one-kernel backward then runs its
    radial-parameter GEMM once per UNORDERED pair tile (R(i, j) = R(j, i): the owner of the higher group adds both directed edges'
    gradients; level_bwd3.hip, SYM) -- 8, 9 and 10 groups of 4 particles (even / odd counts, partly empty last groups), 1 and 2
    groups, a single particle.  LGN_AMD_BWD_ORDERED=1: the per-ordered-tile form on the same launch shape."""
    if ordered:
        monkeypatch.setenv("LGN_AMD_BWD_ORDERED", "1")
    _level_case(dev, O, False, C, CO, N, B)


def test_level_backward_symmetric_and_ordered_sweeps_agree_on_random_shapes(dev, O, monkeypatch):
    """The two forms of the encoder level backward (unordered pair tiles with both directed edges' radial gradient / one GEMM per
    ordered tile) on the same inputs, over random jet sizes 2 .. 150, channel counts 1 .. 4 and paddings: every output to 1e-11 of
    its own scale (the forms differ in summation order only)."""
    from lgn import _native as Nn
    g = torch.Generator().manual_seed(12345)
    for trial in range(14):
        N = int(torch.randint(2, 41, (1,), generator=g)) if trial < 10 else int(torch.randint(41, 151, (1,), generator=g))
        C = int(torch.randint(1, 5, (1,), generator=g)); CO = int(torch.randint(1, 7, (1,), generator=g))
        B = 257 + int(torch.randint(0, 40, (1,), generator=g)) if N <= 40 else int(torch.randint(1, 4, (1,), generator=g))
        cfg, plans, P = _rand_level_params(O, C, CO, False, g)
        pre = "rad_funcs.rad_funcs.0."
        rad = tuple(P[pre + n].to(dev).contiguous() for n in ["a", "b", "c", "linear.0.weight", "linear.0.bias", "linear.1.weight", "linear.1.bias"])
        wm0 = P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(0, 0)"].to(dev).contiguous()
        wm1 = P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(1, 1)"].to(dev).contiguous()
        s = torch.randn(2, B, N, C, dtype=torch.float64, generator=g).to(dev)
        v = torch.randn(2, B, N, C, 4, dtype=torch.float64, generator=g).to(dev)
        gs = torch.randn(2, B, N, CO, dtype=torch.float64, generator=g).to(dev)
        gv = torch.randn(2, B, N, CO, 4, dtype=torch.float64, generator=g).to(dev)
        p4, labels = O.synthetic_jets(B, N, seed=100 + trial, pad=N > 3)
        p, mask = p4.to(dev), labels.to(dev)
        ag0, ag1, so, vo = Nn.level_fwd(False, s, v, p, mask, rad, wm0, wm1)
        outs = []
        for ordered in (False, True):
            if ordered:
                monkeypatch.setenv("LGN_AMD_BWD_ORDERED", "1")
            else:
                monkeypatch.delenv("LGN_AMD_BWD_ORDERED", raising=False)
            outs.append(Nn.level_bwd(False, s, v, p, mask, rad, wm0, wm1, ag0, ag1, gs, gv, None))
        monkeypatch.delenv("LGN_AMD_BWD_ORDERED", raising=False)
        (g_s, g_v, g_w0, g_w1, rg), (h_s, h_v, h_w0, h_w1, rh) = outs
        what = f"trial {trial}: N={N} C={C} CO={CO} B={B}"
        for x, y, name in ((g_s, h_s, "g_s"), (g_v, h_v, "g_v"), (g_w0, h_w0, "g_wm0"), (g_w1, h_w1, "g_wm1")):
            U.assert_close(x, y, 1e-11, f"{what} {name}")
        for x, y in zip(rg, rh):
            if y.abs().max() > 0:
                U.assert_close(x, y, 1e-10, f"{what} radial gradient")
            else:
                assert x.abs().max() == 0


@pytest.mark.parametrize("C,CO,N,B", [(4, 4, 150, 1), (3, 3, 150, 2), (4, 4, 48, 2), (3, 4, 50, 1), (4, 3, 41, 2), (2, 2, 63, 1), (4, 4, 70, 1),
                                      (5, 6, 100, 1), (2, 2, 260, 1)])      # (N > 256: more than 64 groups of four particles)
def test_level_backward_large_jets_ordered_pair_tiles(dev, O, monkeypatch, C, CO, N, B):
    """N > 40: the one-sweep backward with LGN_AMD_BWD_ORDERED=1 -- the radial-parameter GEMM per ordered pair tile, round-robin
    group ownership -- beside the default symmetric sweep that test_level_fwd_bwd runs (unordered tiles, groups dealt by cost;
    C > 4 always takes the ordered form)."""
    monkeypatch.setenv("LGN_AMD_BWD_ORDERED", "1")
    _level_case(dev, O, False, C, CO, N, B)


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("C,CO,N,B", [(4, 4, 30, 2), (3, 4, 30, 3), (4, 3, 7, 1), (2, 5, 33, 2), (4, 4, 48, 2), (3, 4, 70, 1)])
def test_level_three_kernel_backward_small_jets(dev, O, monkeypatch, decoder, C, CO, N, B):
    """LGN_AMD_LEVEL_V2=1 (read per call) selects the three-kernel backward (mix + nodes2 + rad2: two pair sweeps) -- for small jets
    instead of the one-kernel backward, for large encoder jets instead of mix + the one-sweep kernel (level_bwd_sweep_enc)."""
    monkeypatch.setenv("LGN_AMD_LEVEL_V2", "1")
    _level_case(dev, O, decoder, C, CO, N, B)


@pytest.mark.parametrize("decoder", [False, True])
def test_level_large_batch_of_large_jets_matches_single_jet(dev, O, decoder):
    """N >= 64 with B > 320 takes the non-'wide' launch shapes (4-wave sweeps, chunked forward under the 78 KB budget);
    the oracle cannot hold such a batch, so the property used is batch independence: jets 0, 137 and 329 of a B = 330,
    N = 70 batch must equal the same jets computed in batches of their own (B = 1: the 'wide' path, which the oracle
    pins in test_level_fwd_bwd), forward and input/position gradients; parameter gradients must be the sum over jets
    (checked on a B = 322 batch made of two copies of a 161-jet batch: exactly twice its gradients)."""
    from lgn import _native as Nn
    C, CO, N, B = 4, 4, 70, 330
    g = torch.Generator().manual_seed(7 + int(decoder))
    cfg, plans, P = _rand_level_params(O, C, CO, decoder, g)
    pre = "rad_funcs.rad_funcs.0."
    names = ["a", "b", "c", "linear.0.weight", "linear.0.bias", "linear.1.weight", "linear.1.bias"]
    rad = tuple(P[pre + n].to(dev).contiguous() for n in names)
    if decoder:
        rad = (None, None, None, None, rad[4], None, rad[6])
    wm0 = P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(0, 0)"].to(dev).contiguous()
    wm1 = P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(1, 1)"].to(dev).contiguous()

    def run(s, v, p, mask, gs, gv):
        ag0, ag1, so, vo = Nn.level_fwd(decoder, s, v, p, mask, rad, wm0, wm1)
        g_p = torch.zeros_like(p) if decoder else None
        g_s, g_v, g_w0, g_w1, rg = Nn.level_bwd(decoder, s, v, p, mask, rad, wm0, wm1, ag0, ag1, gs, gv, g_p)
        return so, vo, g_s, g_v, g_p, g_w0, g_w1, rg

    def batch(Bn, seed):
        gg = torch.Generator().manual_seed(seed)
        s = torch.randn(2, Bn, N, C, dtype=torch.float64, generator=gg).to(dev)
        v = torch.randn(2, Bn, N, C, 4, dtype=torch.float64, generator=gg).to(dev)
        gs = torch.randn(2, Bn, N, CO, dtype=torch.float64, generator=gg).to(dev)
        gv = torch.randn(2, Bn, N, CO, 4, dtype=torch.float64, generator=gg).to(dev)
        if decoder:
            p, mask = torch.randn(2, Bn, N, 4, dtype=torch.float64, generator=gg).to(dev), None
        else:
            p4, labels = O.synthetic_jets(Bn, N, seed=seed, pad=True)
            p, mask = p4.to(dev), labels.to(dev)
        return s, v, p, mask, gs, gv

    s, v, p, mask, gs, gv = batch(B, 3)
    big = run(s, v, p, mask, gs, gv)
    for k in (0, 137, 329):
        sl = slice(k, k + 1)
        one = run(s[:, sl].contiguous(), v[:, sl].contiguous(), (p[:, sl] if decoder else p[sl]).contiguous(),
                  None if decoder else mask[sl].contiguous(), gs[:, sl].contiguous(), gv[:, sl].contiguous())
        for i, what in enumerate(("s_out", "v_out", "g_s_in", "g_v_in")):
            U.assert_close(big[i][:, sl], one[i], 1e-12, f"jet {k} {what}")
        if decoder:
            U.assert_close(big[4][:, sl], one[4], 1e-12, f"jet {k} g_p")
    half = batch(161, 5)
    bdim = (1, 1, 1 if decoder else 0, 0, 1, 1)            # batch axis of (s, v, p, mask, gs, gv)
    dbl = tuple(None if t is None else torch.cat([t, t], ax).contiguous() for t, ax in zip(half, bdim))
    r1, r2 = run(*half), run(*dbl)
    U.assert_close(r2[5], 2 * r1[5], 1e-11, "g_wm0 additivity")
    U.assert_close(r2[6], 2 * r1[6], 1e-11, "g_wm1 additivity")
    for a, b in zip(r2[7], r1[7]):
        if b.abs().max() > 0:
            U.assert_close(a, 2 * b, 1e-10, "radial gradient additivity")


@pytest.mark.parametrize("C,CO,N,B", [(4, 4, 30, 2), (3, 4, 30, 2), (4, 3, 7, 1), (4, 4, 70, 1)])
def test_level_decoder_pair_sweep(dev, O, monkeypatch, C, CO, N, B):
    """The decoder levels normally run the separable O(N C) form (edge mask == 0 -> constant radial weights);
    LGN_AMD_DEC_PAIRWISE=1 keeps the O(N^2) pair sweep of the reference's formulation.  Both must match the oracle."""
    monkeypatch.setenv("LGN_AMD_DEC_PAIRWISE", "1")
    _level_case(dev, O, True, C, CO, N, B)


def _level_case(dev, O, decoder, C, CO, N, B):
    from lgn import ops
    g = torch.Generator().manual_seed(100 * C + 10 * CO + N + int(decoder))
    cfg, plans, P = _rand_level_params(O, C, CO, decoder, g)
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    node = {(1, 1): torch.randn(2, B, N, C, 4, dtype=torch.float64, generator=g).requires_grad_(True),
            (0, 0): torch.randn(2, B, N, C, 1, dtype=torch.float64, generator=g).requires_grad_(True)}
    if decoder:
        p = torch.randn(2, B, N, 4, dtype=torch.float64, generator=g).requires_grad_(True)
        zonal, norms, _ = O.zonal_rel(p, p, "canonical")
        mask = None
        emask = torch.zeros(2, B, N, N, dtype=torch.float64)
    else:
        p4, labels = O.synthetic_jets(B, N, seed=N + C, pad=N > 4)
        p, mask = p4, labels
        zonal, norms, _ = O.zonal_rel(p, p, "cartesian")
        em = mask.unsqueeze(1) * mask.unsqueeze(2)
        emask = em * (norms != 0).byte()
    rad = O.radial_filters(P, cfg, 0, norms, emask, decoder)
    edge = {k: O.scalar_times_irrep(rad[k], zonal[k]) for k in rad}
    out = O.node_level(P, O.get_cg(2), cfg, 0, plans[0], node, edge)
    cot = {k: torch.randn(v.shape, dtype=torch.float64, generator=g) for k, v in out.items()}
    sum((out[k] * cot[k]).sum() for k in out).backward()

    # native
    d = lambda t: t.detach().to(dev).requires_grad_(t.requires_grad)  # noqa: E731
    s_in = d(node[(0, 0)].squeeze(-1).detach().requires_grad_(True))
    v_in = d(node[(1, 1)])
    pd = d(p)
    pre = "rad_funcs.rad_funcs.0."
    names = ["a", "b", "c", "linear.0.weight", "linear.0.bias", "linear.1.weight", "linear.1.bias"]
    radp = [d(P[pre + n]) for n in names]
    wm0 = d(P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(0, 0)"])
    wm1 = d(P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(1, 1)"])
    s_out, v_out = ops.LevelFn.apply(decoder, s_in, v_in, pd, None if decoder else mask.to(dev), *radp, wm0, wm1)
    U.assert_close(s_out.unsqueeze(-1), out[(0, 0)], FWD_TOL, "s_out")
    U.assert_close(v_out, out[(1, 1)], FWD_TOL, "v_out")
    ((s_out.unsqueeze(-1) * cot[(0, 0)].to(dev)).sum() + (v_out * cot[(1, 1)].to(dev)).sum()).backward()
    U.assert_close(s_in.grad.unsqueeze(-1), node[(0, 0)].grad, GRAD_TOL, "g_s_in")
    U.assert_close(v_in.grad, node[(1, 1)].grad, GRAD_TOL, "g_v_in")
    if decoder:
        U.assert_close(pd.grad, p.grad, GRAD_TOL, "g_p")
    for n, t in zip(names, radp):
        ref = P[pre + n].grad
        ref = torch.zeros_like(P[pre + n]) if ref is None else ref
        if ref.abs().max() == 0:
            assert t.grad is None or t.grad.abs().max() == 0, f"{n} must have exactly zero gradient"
        else:
            U.assert_close(t.grad, ref, GRAD_TOL, "g_" + n)
    U.assert_close(wm0.grad, P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(0, 0)"].grad, GRAD_TOL, "g_wm0")
    U.assert_close(wm1.grad, P["lgn_cg.node_levels.0.cat_mix.mix_reps.weights.(1, 1)"].grad, GRAD_TOL, "g_wm1")


def _level_mlp_inputs(O, decoder, C, CO, N, B, act, g, seed):
    """Random level + CGMLP parameters (every parameter O(1)), a jet batch and cotangents; returns what both sides need."""
    cfg = O.NetConfig(num_channels=(C, CO), activation=act)
    P = {}
    torch.manual_seed(seed)
    O._init_radial(P, cfg, decoder)
    plans = O.build_level_plans(cfg, {(0, 0): C, (1, 1): C})
    O._init_levels(P, cfg, plans)
    for k in P:
        if "cat_mix" in k:
            P[k] = torch.randn(P[k].shape, dtype=torch.float64, generator=g) * 0.3
    node = {(1, 1): torch.randn(2, B, N, C, 4, dtype=torch.float64, generator=g),
            (0, 0): torch.randn(2, B, N, C, 1, dtype=torch.float64, generator=g)}
    if decoder:
        p, mask = torch.randn(2, B, N, 4, dtype=torch.float64, generator=g), None
    else:
        p, mask = O.synthetic_jets(B, N, seed=seed, pad=N > 4)
    cot = {(0, 0): torch.randn(2, B, N, CO, 1, dtype=torch.float64, generator=g),
           (1, 1): torch.randn(2, B, N, CO, 4, dtype=torch.float64, generator=g)}
    return cfg, plans, P, node, p, mask, cot


_RAD_NAMES = ["a", "b", "c", "linear.0.weight", "linear.0.bias", "linear.1.weight", "linear.1.bias"]
_MIX = "lgn_cg.node_levels.0.cat_mix.mix_reps.weights."


def _level_mlp_native(dev, decoder, act, P, node, p, mask, cot):
    """level, then its CGMLP (ops.LevelFn, ops.CGMLPFn: the pair of lgn/models/lgn_cg.py:164-172); returns outputs and every gradient on
    the CPU."""
    from lgn import ops, _native as Nn
    d = lambda t: t.detach().to(dev).requires_grad_(True)  # noqa: E731
    s_in, v_in, pd = d(node[(0, 0)].squeeze(-1)), d(node[(1, 1)]), d(p)
    radp = [d(P["rad_funcs.rad_funcs.0." + n]) for n in _RAD_NAMES]
    wm0, wm1 = d(P[_MIX + "(0, 0)"]), d(P[_MIX + "(1, 1)"])
    flat = []
    for i in range(7):
        flat += [d(P[f"lgn_cg.mlp_levels.0.linear.{i}.weight"]), d(P[f"lgn_cg.mlp_levels.0.linear.{i}.bias"])]
    s_pre, v_out = ops.LevelFn.apply(decoder, s_in, v_in, pd, None if mask is None else mask.to(dev), *radp, wm0, wm1)
    s_out = ops.CGMLPFn.apply(Nn.activation_id(act), s_pre, *flat)
    ((s_out.unsqueeze(-1) * cot[(0, 0)].to(dev)).sum() + (v_out * cot[(1, 1)].to(dev)).sum()).backward()
    z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad  # noqa: E731
    grads = {"s_in": s_in.grad.unsqueeze(-1), "v_in": v_in.grad, "p": pd.grad if decoder else None, "wm0": wm0.grad, "wm1": wm1.grad}
    grads.update({"rad." + n: z(t) for n, t in zip(_RAD_NAMES, radp)})
    for i in range(7):
        grads[f"mlp.{i}.weight"], grads[f"mlp.{i}.bias"] = flat[2 * i].grad, flat[2 * i + 1].grad
    return s_out.unsqueeze(-1), v_out, grads


def _level_mlp_oracle(O, decoder, cfg, plans, P, node, p, mask, cot):
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    node = {k: v.clone().requires_grad_(True) for k, v in node.items()}
    p = p.clone().requires_grad_(True) if decoder else p
    B, N = node[(0, 0)].shape[1:3]
    if decoder:
        zonal, norms, _ = O.zonal_rel(p, p, "canonical")
        emask = torch.zeros(2, B, N, N, dtype=torch.float64)
    else:
        zonal, norms, _ = O.zonal_rel(p, p, "cartesian")
        emask = (mask.unsqueeze(1) * mask.unsqueeze(2)) * (norms != 0).byte()
    rad = O.radial_filters(P, cfg, 0, norms, emask, decoder)
    edge = {k: O.scalar_times_irrep(rad[k], zonal[k]) for k in rad}
    out = O.cg_mlp(P, cfg, 0, O.node_level(P, O.get_cg(2), cfg, 0, plans[0], node, edge))
    sum((out[k] * cot[k]).sum() for k in out).backward()
    z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad  # noqa: E731
    grads = {"s_in": node[(0, 0)].grad, "v_in": node[(1, 1)].grad, "p": p.grad if decoder else None,
             "wm0": P[_MIX + "(0, 0)"].grad, "wm1": P[_MIX + "(1, 1)"].grad}
    grads.update({"rad." + n: z(P["rad_funcs.rad_funcs.0." + n]) for n in _RAD_NAMES})
    for i in range(7):
        grads[f"mlp.{i}.weight"] = P[f"lgn_cg.mlp_levels.0.linear.{i}.weight"].grad
        grads[f"mlp.{i}.bias"] = P[f"lgn_cg.mlp_levels.0.linear.{i}.bias"].grad
    return out[(0, 0)], out[(1, 1)], grads


def _assert_grads(got, ref, tol, tag=""):
    for k, r in ref.items():
        if r is None:
            continue
        if r.abs().max() == 0:
            assert got[k].abs().max() == 0, f"{tag}{k} must have exactly zero gradient"
        else:
            U.assert_close(got[k], r, tol, tag + "g_" + k)


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("C,CO,N,B", [(3, 3, 30, 3), (3, 4, 30, 2), (4, 4, 30, 2), (4, 3, 30, 2),      # the four BASELINE level shapes
                                      (4, 4, 7, 2), (3, 4, 7, 1), (4, 3, 33, 2), (3, 3, 33, 1),          # one tile; two passes
                                      (4, 4, 16, 2), (4, 4, 17, 1), (3, 3, 32, 2), (4, 4, 40, 1),        # tile / pass boundaries
                                      (2, 2, 30, 2), (1, 1, 5, 1), (4, 2, 13, 3), (2, 4, 30, 1),         # narrow MLPs (H = 12, 24)
                                      (5, 4, 30, 2), (4, 5, 30, 1), (4, 4, 48, 1),                       # C > 4, N > 40
                                      (4, 4, 30, 66), (3, 3, 30, 70)])                                  # jet split 4 (65 .. 128 jets)
def test_level_mlp_fwd_bwd(dev, O, decoder, C, CO, N, B):
    """LGNNodeLevel followed by its CGMLP against the oracle, over the level shapes of the BASELINE configs, tile / pass boundaries of
    the level kernels, narrow CGMLPs and channel counts on either side of the C <= 4 kernels (small B: jets split over workgroups)."""
    g = torch.Generator().manual_seed(1000 * C + 100 * CO + N + int(decoder))
    cfg, plans, P, node, p, mask, cot = _level_mlp_inputs(O, decoder, C, CO, N, B, "leakyrelu", g, seed=N + 7 * C)
    s_ref, v_ref, g_ref = _level_mlp_oracle(O, decoder, cfg, plans, P, node, p, mask, cot)
    s_out, v_out, g_got = _level_mlp_native(dev, decoder, "leakyrelu", P, node, p, mask, cot)
    U.assert_close(s_out, s_ref, FWD_TOL, "s_out (after the MLP)")
    U.assert_close(v_out, v_ref, FWD_TOL, "v_out")
    _assert_grads(g_got, g_ref, GRAD_TOL)


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("act", ["leakyrelu", "relu", "elu", "sigmoid", "logsigmoid", "atan"])
def test_level_mlp_activations(dev, O, decoder, act):
    """Every activation of get_activation_fn through the level + CGMLP pair."""
    C, CO, N, B = 4, 3, 30, 2
    g = torch.Generator().manual_seed(77 + int(decoder))
    cfg, plans, P, node, p, mask, cot = _level_mlp_inputs(O, decoder, C, CO, N, B, act, g, seed=11)
    s_ref, v_ref, g_ref = _level_mlp_oracle(O, decoder, cfg, plans, P, node, p, mask, cot)
    s_out, v_out, g_got = _level_mlp_native(dev, decoder, act, P, node, p, mask, cot)
    U.assert_close(s_out, s_ref, FWD_TOL, f"{act} s_out")
    U.assert_close(v_out, v_ref, FWD_TOL, f"{act} v_out")
    _assert_grads(g_got, g_ref, GRAD_TOL, act + " ")


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("C,CO", [(3, 4), (4, 4), (4, 3)])
def test_level_mlp_full_batch_matches_small_batches_and_v1(dev, O, monkeypatch, decoder, C, CO):
    """B = 512 (one workgroup per jet, two per CU; the CGMLP on the chain kernels of mlp_chain.hip: the BASELINE launch shapes) cannot be
    held by the oracle.  Properties instead: (1) jets of the 512-batch equal the same jets run in a batch of 3 (the jet-split level
    launch and the 16-row CGMLP workgroups, which the oracle pins in test_level_mlp_fwd_bwd); (2) every output and gradient equals the
    12-wave CGMLP kernels' (LGN_AMD_MLP_V1=1) to rounding; (3) the run is bitwise reproducible."""
    N, B = 30, 512
    g = torch.Generator().manual_seed(5 + C + int(decoder))
    cfg, plans, P, node, p, mask, cot = _level_mlp_inputs(O, decoder, C, CO, N, B, "leakyrelu", g, seed=3)
    s1, v1, g1 = _level_mlp_native(dev, decoder, "leakyrelu", P, node, p, mask, cot)
    s2, v2, g2 = _level_mlp_native(dev, decoder, "leakyrelu", P, node, p, mask, cot)
    assert torch.equal(s1, s2) and torch.equal(v1, v2) and all(torch.equal(g1[k], g2[k]) for k in g1 if g1[k] is not None)
    sl = [0, 255, 511]
    sub = lambda t, ax: t.index_select(ax, torch.tensor(sl)).contiguous()  # noqa: E731
    node3 = {k: sub(v, 1) for k, v in node.items()}
    cot3 = {k: sub(v, 1) for k, v in cot.items()}
    p3, m3 = (sub(p, 1), None) if decoder else (sub(p, 0), sub(mask, 0))
    s3, v3, g3 = _level_mlp_native(dev, decoder, "leakyrelu", P, node3, p3, m3, cot3)
    idx = torch.tensor(sl, device=dev)
    U.assert_close(s1.index_select(1, idx), s3, 1e-13, "jets of the full batch: s_out")
    U.assert_close(v1.index_select(1, idx), v3, 1e-13, "jets of the full batch: v_out")
    U.assert_close(g1["s_in"].index_select(1, idx), g3["s_in"], 1e-12, "jets of the full batch: g_s_in")
    U.assert_close(g1["v_in"].index_select(1, idx), g3["v_in"], 1e-12, "jets of the full batch: g_v_in")
    monkeypatch.setenv("LGN_AMD_MLP_V1", "1")
    su, vu, gu = _level_mlp_native(dev, decoder, "leakyrelu", P, node, p, mask, cot)
    U.assert_close(s1, su, 1e-13, "chain vs 12-wave CGMLP kernels: s_out")
    U.assert_close(v1, vu, 1e-13, "chain vs 12-wave CGMLP kernels: v_out")
    _assert_grads(g1, gu, 1e-11, "chain vs 12-wave CGMLP kernels: ")


@pytest.mark.parametrize("act", ["relu", "elu", "sigmoid", "logsigmoid", "atan"])
def test_cgmlp_activations_vs_reference_golden(dev, act):
    """The CGMLP kernels with every non-default activation against vectors of the reference's CGMLP (g9)."""
    from lgn import ops, _native as Nn
    z = U.load("g9_activations.npz")
    P = U.params_from(z, f"{act}.param")
    s = U.rep_from(z, "in")[(0, 0)].squeeze(-1).to(dev).requires_grad_(True)
    flat = []
    for i in range(7):
        flat += [P[f"linear.{i}.weight"].to(dev).requires_grad_(True), P[f"linear.{i}.bias"].to(dev).requires_grad_(True)]
    y = ops.CGMLPFn.apply(Nn.activation_id(act), s, *flat)
    U.assert_close(y.unsqueeze(-1), U.rep_from(z, f"{act}.out")[(0, 0)], FWD_TOL, f"{act} out")
    (y.unsqueeze(-1) * torch.from_numpy(z["cot"]).to(dev)).sum().backward()
    U.assert_close(s.grad.unsqueeze(-1), z[f"{act}.grad_in"], GRAD_TOL, f"{act} g_in")
    for i in range(7):
        U.assert_close(flat[2 * i].grad, z[f"{act}.grad.linear.{i}.weight"], GRAD_TOL, f"{act} g_w{i}")
        U.assert_close(flat[2 * i + 1].grad, z[f"{act}.grad.linear.{i}.bias"], GRAD_TOL, f"{act} g_b{i}")


@pytest.mark.parametrize("C,B,N,act", [(3, 2, 30, "leakyrelu"), (4, 3, 30, "leakyrelu"), (4, 1, 150, "leakyrelu"), (2, 1, 5, "leakyrelu"),
                                       (6, 1, 40, "leakyrelu"), (6, 3, 37, "leakyrelu"), (5, 2, 30, "leakyrelu"), (8, 1, 70, "leakyrelu"),
                                       # >= 8192 rows: 64-row workgroups (fewer rows: 16-row ones, H <= 48)
                                       (4, 300, 30, "leakyrelu"), (3, 275, 30, "leakyrelu"), (2, 280, 30, "leakyrelu"), (1, 300, 30, "relu"),
                                       # ... and 48 < H <= 96: the chain forward, the wide backward
                                       (6, 300, 30, "leakyrelu"), (5, 275, 30, "leakyrelu"), (8, 280, 30, "elu"), (7, 272, 30, "leakyrelu"),
                                       # the other activations through every kernel family: 16-row and 64-row H <= 48, one-pass and
                                       # two-pass wide
                                       (4, 3, 30, "elu"), (4, 300, 30, "sigmoid"), (6, 3, 37, "atan"), (5, 2, 30, "logsigmoid"),
                                       (8, 1, 70, "relu"), (3, 2, 30, "atan")])
def test_cgmlp(dev, O, C, B, N, act):
    from lgn import ops, _native as Nn
    g = torch.Generator().manual_seed(C * 7 + N)
    cfg = O.NetConfig(num_channels=(C, C), activation=act)
    P = {}
    torch.manual_seed(C)
    plans = O.build_level_plans(cfg, {(0, 0): C, (1, 1): C})
    O._init_levels(P, cfg, plans)
    P = {k: v.requires_grad_(True) for k, v in P.items() if "mlp" in k}
    s = torch.randn(2, B, N, C, 1, dtype=torch.float64, generator=g).requires_grad_(True)
    node = {(1, 1): torch.zeros(2, B, N, C, 4, dtype=torch.float64), (0, 0): s}
    out = O.cg_mlp(P, cfg, 0, node)[(0, 0)]
    cot = torch.randn(out.shape, dtype=torch.float64, generator=g)
    (out * cot).sum().backward()

    sd = s.detach().squeeze(-1).to(dev).requires_grad_(True)
    flat = []
    for i in range(7):
        flat += [P[f"lgn_cg.mlp_levels.0.linear.{i}.weight"].detach().to(dev).requires_grad_(True),
                 P[f"lgn_cg.mlp_levels.0.linear.{i}.bias"].detach().to(dev).requires_grad_(True)]
    y = ops.CGMLPFn.apply(Nn.activation_id(act), sd, *flat)
    U.assert_close(y.unsqueeze(-1), out, FWD_TOL, "mlp out")
    (y.unsqueeze(-1) * cot.to(dev)).sum().backward()
    U.assert_close(sd.grad.unsqueeze(-1), s.grad, GRAD_TOL, "mlp g_in")
    for i in range(7):
        U.assert_close(flat[2 * i].grad, P[f"lgn_cg.mlp_levels.0.linear.{i}.weight"].grad, GRAD_TOL, f"g_w{i}")
        U.assert_close(flat[2 * i + 1].grad, P[f"lgn_cg.mlp_levels.0.linear.{i}.bias"].grad, GRAD_TOL, f"g_b{i}")


@pytest.mark.parametrize("C,B,N,width,depth", [
    # mlp_width other than 6 (lgn/models/lgn_levels.py:124-189: hidden width = mlp_width x 2C).  Few rows: the 16-row workgroups
    (3, 2, 30, 4, 6), (3, 2, 30, 5, 6), (4, 3, 30, 4, 6), (4, 3, 30, 5, 6), (4, 3, 30, 7, 6), (2, 1, 5, 1, 6), (1, 2, 30, 9, 6),
    (4, 2, 30, 12, 6), (8, 1, 30, 3, 6), (6, 2, 30, 5, 6), (3, 2, 30, 8, 6), (5, 1, 33, 9, 6), (7, 2, 20, 2, 6), (3, 2, 30, 16, 6),
    # >= 8 192 rows: 64-row workgroups -- H = 24 / 30 / 32 / 40 (C = 3, 4: the widths VERDICT r5 names), H = 48 at C != 4 (the
    # unrolled-k 12-wave kernel, NOT the chain kernel, which is built for H = 6 x 2C), H = 56 / 60 / 80 / 96 (wide kernels)
    (3, 275, 30, 4, 6), (3, 275, 30, 5, 6), (4, 275, 30, 4, 6), (4, 275, 30, 5, 6), (4, 280, 30, 7, 6), (3, 275, 30, 8, 6),
    (6, 275, 30, 4, 6), (2, 280, 30, 12, 6), (6, 275, 30, 5, 6), (8, 275, 30, 5, 6), (4, 275, 30, 12, 6), (1, 300, 30, 7, 6),
    (2, 275, 30, 6, 6), (1, 275, 30, 6, 6),
    # ... and together with mlp_depth 3 .. 5
    (3, 275, 30, 5, 3), (4, 275, 30, 4, 4), (4, 3, 30, 7, 5), (3, 2, 30, 5, 4)])
def test_cgmlp_widths(dev, O, C, B, N, width, depth):
    """CGMLP forward + all gradients against the oracle at hidden widths that are NOT 6 x 2C (every BASELINE config and every
    other test uses mlp_width = 6): tile-padding paths of mlp_mfma.hip / mlp_mfma_wide.hip (H not a multiple of 16 or of 4)."""
    from lgn import ops, _native as Nn
    g = torch.Generator().manual_seed(C * 7 + N + width)
    cfg = O.NetConfig(num_channels=(C, C), mlp_width=width, mlp_depth=depth)
    P = {}
    torch.manual_seed(C + width)
    plans = O.build_level_plans(cfg, {(0, 0): C, (1, 1): C})
    O._init_levels(P, cfg, plans)
    P = {k: v.requires_grad_(True) for k, v in P.items() if "mlp" in k}
    nlin = depth + 1
    assert tuple(P["lgn_cg.mlp_levels.0.linear.0.weight"].shape) == (width * 2 * C, 2 * C)
    s = torch.randn(2, B, N, C, 1, dtype=torch.float64, generator=g).requires_grad_(True)
    node = {(1, 1): torch.zeros(2, B, N, C, 4, dtype=torch.float64), (0, 0): s}
    out = O.cg_mlp(P, cfg, 0, node)[(0, 0)]
    cot = torch.randn(out.shape, dtype=torch.float64, generator=g)
    (out * cot).sum().backward()
    sd = s.detach().squeeze(-1).to(dev).requires_grad_(True)
    flat = []
    for i in range(nlin):
        flat += [P[f"lgn_cg.mlp_levels.0.linear.{i}.weight"].detach().to(dev).requires_grad_(True),
                 P[f"lgn_cg.mlp_levels.0.linear.{i}.bias"].detach().to(dev).requires_grad_(True)]
    y = ops.CGMLPFn.apply(Nn.activation_id("leakyrelu"), sd, *flat)
    U.assert_close(y.unsqueeze(-1), out, FWD_TOL, "mlp out")
    (y.unsqueeze(-1) * cot.to(dev)).sum().backward()
    U.assert_close(sd.grad.unsqueeze(-1), s.grad, GRAD_TOL, "mlp g_in")
    for i in range(nlin):
        U.assert_close(flat[2 * i].grad, P[f"lgn_cg.mlp_levels.0.linear.{i}.weight"].grad, GRAD_TOL, f"g_w{i}")
        U.assert_close(flat[2 * i + 1].grad, P[f"lgn_cg.mlp_levels.0.linear.{i}.bias"].grad, GRAD_TOL, f"g_b{i}")


@pytest.mark.parametrize("tag,maxdim", [("cg2_2", 2), ("cg3_5", 3), ("cg3_2", 3)])
def test_cg_product_vs_reference_golden(dev, O, tag, maxdim):
    """lgn.cg_lib.cg_product / CGProduct (lgn_cg_product_fwd/bwd_f64, csrc/cg_product.hip) against the reference's own vectors (g4:
    aggregate = node (x) edge summed over neighbours, power = node (x) node; values, key order, channel order) and -- gradients --
    against the oracle's autograd on the same inputs, aggregate with the edge-like operand on either side."""
    from lgn.cg_lib import CGDict, CGProduct, cg_product
    z = U.load("g4_ops.npz")
    cg = CGDict(maxdim=maxdim, device=dev)
    node, edge = U.rep_from(z, tag + ".node"), U.rep_from(z, tag + ".edge")
    gn = {k: v.to(dev).requires_grad_(True) for k, v in node.items()}
    ge = {k: v.to(dev).requires_grad_(True) for k, v in edge.items()}
    agg = cg_product(cg, gn, ge, maxdim, aggregate=True)
    U.assert_rep_close(dict(agg.items()), U.rep_from(z, tag + ".aggregate"), FWD_TOL, "aggregate")
    power = CGProduct(maxdim=maxdim, cg_dict=cg)(U_gvec(gn), U_gvec(gn))
    pref = U.rep_from(z, tag + ".power")
    assert list(power.keys()) == list(pref.keys())
    for k, ref in pref.items():          # ((1,1) (x) (1,1) -> (0,2), (2,0) is antisymmetric: v (x) v gives rounding noise there, 1e-17 in the reference)
        if ref.abs().max() < 1e-13:
            assert power[k].abs().max() < 1e-13, f"power {k} must vanish"
        else:
            U.assert_close(power[k], ref, FWD_TOL, f"power {k}")
    agg2 = cg_product(cg, ge, gn, maxdim, aggregate=True)            # edge-like operand first
    # gradients of one scalar of all three products, against the oracle's autograd
    g = torch.Generator().manual_seed(len(tag) + maxdim)
    on = {k: v.clone().requires_grad_(True) for k, v in node.items()}
    oe = {k: v.clone().requires_grad_(True) for k, v in edge.items()}
    ocg = O.get_cg(maxdim)
    refs = [O.cg_product(ocg, on, oe, maxdim, aggregate=True), O.cg_product(ocg, on, on, maxdim, aggregate=False),
            _cg_product_edge_first(CGDict(maxdim=maxdim), oe, on, maxdim)]
    U.assert_rep_close(dict(agg2.items()), dict(refs[2].items()), FWD_TOL, "aggregate, edge-like operand first")
    tot_ref, tot = 0.0, 0.0
    for got, ref in zip((agg, power, agg2), refs):
        for k in ref.keys():
            cot = torch.randn(ref[k].shape, dtype=torch.float64, generator=g)
            tot_ref = tot_ref + (ref[k] * cot).sum()
            tot = tot + (got[k] * cot.to(dev)).sum()
    tot_ref.backward()
    tot.backward()
    for k in node:
        U.assert_close(gn[k].grad, on[k].grad, GRAD_TOL, f"d node {k}")
    for k in edge:
        U.assert_close(ge[k].grad, oe[k].grad, GRAD_TOL, f"d edge {k}")


def _cg_product_edge_first(cg, edge, node, maxdim):
    """cg_product(edge, node, aggregate=True) written out in torch (the oracle takes the node-like operand first only):
    z[b,i,c,m1 d2 + m2] = sum_j edge[b,i,j,c,m1] node[b,j,c,m2] (complex), then the pair's stacked CG matrix."""
    out = {}
    for (k1, n1), e in edge.items():
        for (k2, n2), x in node.items():
            keys = [(k, n) for k in range(abs(k1 - k2), min(maxdim, k1 + k2 + 1), 2) for n in range(abs(n1 - n2), min(maxdim, n1 + n2 + 1), 2)]
            mat = torch.cat([cg[((k1, n1), (k2, n2))][key] for key in keys], -2)
            er, ei, xr, xi = e[0], e[1], x[0].unsqueeze(1), x[1].unsqueeze(1)             # e (B,N,N,C,d1); x (B,1,N,C,d2)
            zr = (er.unsqueeze(-1) * xr.unsqueeze(-2) - ei.unsqueeze(-1) * xi.unsqueeze(-2)).sum(2).flatten(-2)
            zi = (er.unsqueeze(-1) * xi.unsqueeze(-2) + ei.unsqueeze(-1) * xr.unsqueeze(-2)).sum(2).flatten(-2)
            dec = torch.stack([zr @ mat.t(), zi @ mat.t()])
            for key, piece in zip(keys, torch.split(dec, [(k + 1) * (n + 1) for k, n in keys], dim=-1)):
                out.setdefault(key, []).append(piece)
    return {key: torch.cat(v, dim=-2) for key, v in out.items()}


def U_gvec(rep):
    from lgn.g_lib import GVec
    return GVec(rep)


@pytest.mark.parametrize("rows,Ci,Co,d", [((2, 30), 1, 3, 4), ((2, 30), 4, 8, 4), ((5, 1), 16, 30, 4), ((3, 7), 4, 1, 1),
                                          ((700,), 3, 2, 4)])
def test_mixreps(dev, O, rows, Ci, Co, d):
    from lgn import ops
    g = torch.Generator().manual_seed(Ci * 31 + Co)
    w = torch.randn(2, Co, Ci, dtype=torch.float64, generator=g).requires_grad_(True)
    x = torch.randn((2,) + rows + (Ci, d), dtype=torch.float64, generator=g).requires_grad_(True)
    y = O.mix_weight_vec(w, x)
    cot = torch.randn(y.shape, dtype=torch.float64, generator=g)
    (y * cot).sum().backward()
    wd, xd = w.detach().to(dev).requires_grad_(True), x.detach().to(dev).requires_grad_(True)
    yd = ops.MixFn.apply(wd, xd)
    U.assert_close(yd, y, FWD_TOL, "mix y")
    (yd * cot.to(dev)).sum().backward()
    U.assert_close(wd.grad, w.grad, GRAD_TOL, "mix g_w")
    U.assert_close(xd.grad, x.grad, GRAD_TOL, "mix g_x")


@pytest.mark.parametrize("B,N,M", [(2, 7, 19), (1, 1, 5), (3, 40, 9)])
def test_chamfer_kernel_unequal_sets(dev, B, N, M):
    """lgn_chamfer_f64 with N != M (the module refuses it like the reference, whose elementwise sum of the two minima arrays needs
    N == M): sum_i min_j + sum_j min_i, halved, and its gradients, against torch."""
    from lgn import _native as Nn
    from lgn.losses import ChamferLoss
    g = torch.Generator().manual_seed(N + M)
    x = torch.randn(B, N, 4, dtype=torch.float64, generator=g).requires_grad_(True)
    y = torch.randn(B, M, 4, dtype=torch.float64, generator=g).requires_grad_(True)
    d = ((x.unsqueeze(-2) - y.unsqueeze(-3)) ** 2).sum(-1)
    ref = 0.5 * (d.min(dim=-1).values.sum(-1) + d.min(dim=-2).values.sum(-1))
    ref.sum().backward()
    part, gx, gy = Nn.chamfer(x.detach().to(dev), y.detach().to(dev))
    U.assert_close(part, ref, 1e-13, "per-jet terms")
    U.assert_close(gx, x.grad, 1e-12, "d / d x")
    U.assert_close(gy, y.grad, 1e-12, "d / d y")
    with pytest.raises(RuntimeError, match="must match the size"):
        ChamferLoss(device=dev)(x.detach().to(dev), y.detach().to(dev))


@pytest.mark.parametrize("B,N,M,jet", [(4, 30, 30, False), (3, 12, 12, True), (2, 150, 150, True), (1, 1, 1, False), (300, 30, 30, False)])
def test_chamfer_loss_module(dev, O, B, N, M, jet):
    """lgn.losses.ChamferLoss (one kernel: loss + both gradients) against the reference's formula restated in torch fp64
    (oracle: chamfer_loss; utils/losses/chamfer_loss/chamfer_loss.py:16-31), including zero-padded rows (exact ties: first minimum),
    the jet_features MSE term, and the gradient w.r.t. the target."""
    from lgn.losses import ChamferLoss
    g = torch.Generator().manual_seed(100 * N + M + B)
    x = torch.randn(B, N, 4, dtype=torch.float64, generator=g)
    y = torch.randn(B, M, 4, dtype=torch.float64, generator=g)
    if N > 4 and M > 4:
        x[0, N - 2:] = 0.0                  # padded particles: identical rows -> tied distances
        y[0, M - 3:] = 0.0
        y[-1, 0] = x[-1, 1]                 # an exact zero distance
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ref = O.chamfer_loss(xr, yr)
    if jet:
        ref = ref + torch.nn.MSELoss()(xr.sum(dim=-2), yr.sum(dim=-2))
    (3.0 * ref).backward()
    xd, yd = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
    loss = ChamferLoss(device=dev)(xd, yd, jet_features=jet)
    (3.0 * loss).backward()
    U.assert_close(loss, ref, 1e-13, "chamfer")
    U.assert_close(xd.grad, xr.grad, 1e-12, "d loss / d x")
    U.assert_close(yd.grad, yr.grad, 1e-12, "d loss / d y")
    with torch.no_grad():                    # no graph: forward only, target without gradient
        U.assert_close(ChamferLoss(device=dev)(x.to(dev), y.to(dev), jet), ref, 1e-13, "chamfer (no grad)")
    # leading batch axes are flattened as the reference's broadcasting cdist would treat them
    if B % 2 == 0:
        l2 = ChamferLoss(device=dev)(x.to(dev).view(2, B // 2, N, 4), y.to(dev).view(2, B // 2, M, 4))
        U.assert_close(l2, O.chamfer_loss(x, y), 1e-13, "chamfer, two batch axes")


def _build(meta, dev):
    import __graft_entry__ as G
    enc, dec = G._models(meta["N"], meta["ch_enc"], meta["ch_dec"], dev, seed=meta["seed"], maxdim=meta["maxdim"],
                         map_to_latent=meta.get("map_to_latent", "min&max"), activation=meta.get("activation", "leakyrelu"),
                         jet_features=meta.get("jet_features", False), tau_input_scalars=1 + meta.get("extra_scalars", 0),
                         mlp_depth=meta.get("mlp_depth", 6), num_basis_fn=meta.get("num_basis_fn", 10), mlp_width=meta.get("mlp_width", 6))
    return enc, dec


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("maxdim,full,C,CO,N,B", [(3, True, 4, 6, 30, 2), (3, False, 4, 4, 30, 2), (3, True, 6, 4, 13, 1),
                                                  (2, False, 3, 4, 30, 2), (3, True, 2, 3, 5, 1),
                                                  # natural dispatch beyond the channel-outermost moments kernels (N <= 32) and the
                                                  # tile-blocked separable decoder kernels (N <= 64)
                                                  (3, True, 4, 6, 40, 1), (3, False, 4, 4, 48, 2), (3, True, 3, 4, 70, 1)])
def test_generic_level_fwd_bwd(dev, O, decoder, maxdim, full, C, CO, N, B):
    """Table-driven level (moments + sparse CG + CatMix) for arbitrary irreps vs the oracle's cg_product /
    CatMixReps, forward and all gradients.  `full`: the node carries all five maxdim=3 irreps."""
    _generic_level_case(dev, O, decoder, maxdim, full, C, CO, N, B)


@pytest.mark.parametrize("flag", ["LGN_AMD_DEC_PAIRWISE", "LGN_AMD_MOMENTS_V1"])
@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("maxdim,full,C,CO,N,B", [(3, True, 4, 6, 30, 2), (3, False, 4, 4, 30, 2)])
def test_generic_level_alternative_moments_kernels(dev, O, monkeypatch, flag, decoder, maxdim, full, C, CO, N, B):
    """The same level with LGN_AMD_DEC_PAIRWISE=1 (decoder moments as O(N^2) pair sweeps instead of the separable jet-level
    sums) and with LGN_AMD_MOMENTS_V1=1 (the all-channels-in-flight kernels that serve N > 32)."""
    if flag == "LGN_AMD_DEC_PAIRWISE" and not decoder:
        pytest.skip("decoder-only switch")
    monkeypatch.setenv(flag, "1")
    _generic_level_case(dev, O, decoder, maxdim, full, C, CO, N, B)


def _generic_level_case(dev, O, decoder, maxdim, full, C, CO, N, B):
    from lgn import ops, _native as Nn
    from lgn.cg_lib import CGDict
    from lgn.plan import build_level_plans, build_local_tables, param_key_order
    g = torch.Generator().manual_seed(1000 * maxdim + 10 * C + CO + N + int(decoder) + int(full))
    node_order = [(1, 1), (2, 0), (0, 2), (2, 2), (0, 0)] if full else [(1, 1), (0, 0)]
    cfg = O.NetConfig(num_channels=(C, CO), maxdim=maxdim)
    tau_in = {k: C for k in ([(0, 0), (0, 2), (1, 1), (2, 0), (2, 2)] if full else [(0, 0), (1, 1)])}
    plan_o = O.build_level_plans(cfg, tau_in)[0]
    plan = build_level_plans([C, CO], [maxdim], [1], True, tau_in, node_order)[0]
    P = {}
    torch.manual_seed(int(torch.randint(0, 10000, (1,), generator=g)))
    O._init_radial(P, cfg, decoder)
    for k, n_in in plan_o.tau_cat.items():
        P[f"lgn_cg.node_levels.0.cat_mix.mix_reps.weights.{k}"] = torch.randn(2, CO, n_in, dtype=torch.float64, generator=g) * 0.3
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    node = {r: torch.randn(2, B, N, C, (r[0] + 1) * (r[1] + 1), dtype=torch.float64, generator=g).requires_grad_(True)
            for r in node_order}
    if decoder:
        p = torch.randn(2, B, N, 4, dtype=torch.float64, generator=g).requires_grad_(True)
        zonal, norms, _ = O.zonal_rel(p, p, "canonical")
        mask, emask = None, torch.zeros(2, B, N, N, dtype=torch.float64)
    else:
        p, mask = O.synthetic_jets(B, N, seed=N + C, pad=N > 4)
        zonal, norms, _ = O.zonal_rel(p, p, "cartesian")
        emask = (mask.unsqueeze(1) * mask.unsqueeze(2)) * (norms != 0).byte()
    rad = O.radial_filters(P, cfg, 0, norms, emask, decoder)
    edge = {k: O.scalar_times_irrep(rad[k], zonal[k]) for k in rad}
    out = O.node_level(P, O.get_cg(maxdim), cfg, 0, plan_o, node, edge)
    cot = {k: torch.randn(v.shape, dtype=torch.float64, generator=g) for k, v in out.items()}
    sum((out[k] * cot[k]).sum() for k in out).backward()

    d = lambda t: t.detach().to(dev).requires_grad_(t.requires_grad)  # noqa: E731
    tables = Nn.DeviceTables(build_local_tables(plan, CGDict(maxdim=maxdim)), dev)
    assert set(tables.meta["out_irreps"]) == set(out.keys())
    feats = {r: d(node[r]) for r in node_order}
    X = torch.cat([feats[r] for r in node_order], dim=-1)
    pd = d(p)
    pre = "rad_funcs.rad_funcs.0."
    names = ["a", "b", "c", "linear.0.weight", "linear.0.bias", "linear.1.weight", "linear.1.bias"]
    radp = [d(P[pre + n]) for n in names]
    wmix = [d(P[f"lgn_cg.node_levels.0.cat_mix.mix_reps.weights.{r}"]) for r in tables.meta["out_irreps"]]
    Y = ops.GenericLevelFn.apply(decoder, tables, CO, X, pd, None if decoder else mask.to(dev), *radp, *wmix)
    parts = dict(zip(tables.meta["out_irreps"], torch.split(Y, [(r[0] + 1) * (r[1] + 1) for r in tables.meta["out_irreps"]], dim=-1)))
    for r in out:
        U.assert_close(parts[r], out[r], FWD_TOL, f"out {r}")
    sum((parts[r] * cot[r].to(dev)).sum() for r in out).backward()
    for r in node_order:
        U.assert_close(feats[r].grad, node[r].grad, GRAD_TOL, f"g_node {r}")
    if decoder:
        U.assert_close(pd.grad, p.grad, GRAD_TOL, "g_p")
    for n, t in zip(names, radp):
        ref = P[pre + n].grad
        ref = torch.zeros_like(P[pre + n]) if ref is None else ref
        if ref.abs().max() == 0:
            assert t.grad is None or t.grad.abs().max() == 0, f"{n} must have exactly zero gradient"
        else:
            U.assert_close(t.grad, ref, GRAD_TOL, "g_" + n)
    for r, w in zip(tables.meta["out_irreps"], wmix):
        U.assert_close(w.grad, P[f"lgn_cg.node_levels.0.cat_mix.mix_reps.weights.{r}"].grad, GRAD_TOL, f"g_wmix {r}")


@pytest.mark.parametrize("name", ["g1_e2e_maxdim2.npz", "g3_e2e_n150.npz", "g2_e2e_maxdim3.npz", "g6_e2e_mix.npz",
                                  "g7_e2e_meanmax.npz", "g9_e2e_elu.npz", "g10_e2e_jetfeat.npz", "g11_e2e_mlpdepth4.npz",
                                  "g11_e2e_mlpdepth3_maxdim3.npz", "g12_e2e_n150_maxdim3.npz", "g13_e2e_basis5.npz",
                                  "g13_e2e_basis5_maxdim3.npz", "g14_e2e_mlpwidth4.npz", "g14_e2e_mlpwidth5.npz", "g14_e2e_mlpwidth7.npz",
                                  "g14_e2e_mlpwidth5_maxdim3.npz", "g15_e2e_basis12.npz", "g15_e2e_basis20_maxdim3.npz"])
@pytest.mark.parametrize("fused", [True, False])
def test_end_to_end_vs_reference_golden(dev, O, name, fused):
    """Full encoder -> decoder -> Chamfer forward/backward against vectors captured from the reference, through the
    module API: ``fused`` = one native call per network and direction (lgn_encoder_* / lgn_decoder_*, taken when the
    configuration allows it), else one native call per operator under autograd."""
    z = U.load(name)
    m = U.meta(z)
    enc, dec = _build(m, dev)
    enc.use_fused = dec.use_fused = fused
    # same seed => same initial weights as the reference; load the fixture weights anyway (checkpoint path)
    enc.load_state_dict({k: v for k, v in U.params_from(z, "enc").items()})
    dec.load_state_dict({k: v for k, v in U.params_from(z, "dec").items()})
    p4 = torch.from_numpy(z["p4"]); labels = torch.from_numpy(z["labels"])
    batch = {"p4": p4, "labels": labels}
    if "scalars" in z.files:                 # g10: jet_features + data['scalars'] (lgn_encoder.py:372-411)
        batch["scalars"] = torch.from_numpy(z["scalars"])

    latent, nodes_all = enc(batch, covariance_test=True)
    U.assert_rep_close(dict(latent.items()), U.rep_from(z, "latent"), FWD_TOL, "latent")
    n_enc = len(nodes_all)
    for i, rep in enumerate(nodes_all):
        U.assert_rep_close(dict(rep.items()), U.rep_from(z, f"enc_nodes.{i}"), FWD_TOL, f"enc_nodes[{i}]")
    gen, nodes_all = dec(latent, covariance_test=True, nodes_all=nodes_all)
    for i, rep in enumerate(nodes_all[n_enc:]):
        U.assert_rep_close(dict(rep.items()), U.rep_from(z, f"dec_nodes.{i}"), FWD_TOL, f"dec_nodes[{i}]")

    if fused and name.startswith("g15"):
        # num_basis_fn 12 / 20 (24 / 40 bells): more than the one group of 20 the whole-network calls read -- the per-operator path sums
        # the table-driven moments over the groups (lgn/ops.py: GenericLevelFn), at maxdim 2 as well
        assert not enc._fused_ok() and not dec._fused_ok(), "more than 20 bells must take the per-operator path"
    elif fused:
        # every OTHER fixture's configuration IS covered by the whole-network native calls (since round 4: g10 jet features + extra input
        # scalars, g7 mean+max pooling, g6 the learned 'mix' latent map) -- the training forward below must take them
        # (g12, 150 particles x 6 channels at maxdim 3: the decoder's input stage fits a CU's LDS since round 6 -- its input-mixing terms
        # share the vector-gradient rows; g13, num_basis_fn = 5: the flat parameter block stores the radial tensors 20 bells wide)
        assert enc._fused_ok() and dec._fused_ok(), "expected the one-call-per-network native path"
    rec = dec(enc(batch))
    U.assert_close(rec, z["recon"], FWD_TOL, "recon")
    loss = O.chamfer_loss(rec[0] + rec[1], p4.to(dev))
    U.assert_close(loss, z["loss_chamfer"], FWD_TOL, "chamfer")
    loss.backward()
    # Every gradient tensor is held to GRAD_TOL relative to ITS OWN largest entry.  The named exceptions are tensors whose gradient
    # is the survivor of a cancellation -- rounding noise of the sums they come from is absolute: they are held to GRAD_TOL of the
    # step's largest gradient / 100 instead.  (g9 / ELU: the encoder's input mixing weight gets 5e-12 against 3e-9.)
    top = max(float(abs(z[k]).max()) for k in z.files if k.startswith("grad."))
    floor = 1e-2 * top
    scaled = _SCALED_GRADS.get(name, ())
    routed = []        # tensors that did NOT take the strict per-tensor check: printed (pytest -s / -rA) so that a drift stays visible
    for pre, mod in (("enc", enc), ("dec", dec)):
        assert [n for n, _ in mod.named_parameters()] == ["flat_params"]
        for k, got in mod.named_grads():
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"])
            if ref.abs().max() == 0:
                assert got.abs().max() == 0, f"{pre}.{k} must have exactly zero gradient"
            elif f"{pre}.{k}" in scaled:
                U.assert_close_scaled(got, ref, GRAD_TOL, floor, f"grad {pre}.{k}")
                routed.append((f"{pre}.{k}", "_SCALED_GRADS", (got.detach().cpu() - ref).abs().max().item(), ref.abs().max().item()))
            elif (got.detach().cpu() - ref).abs().max().item() > ROUNDING * top:
                # (a tensor 7 orders below the step's largest gradient may differ by a few roundings OF THAT LARGEST gradient -- 64 eps --
                # whatever that is relative to itself: g6 / g7 through the whole-network calls, 4e-21 absolute on tensors of 5e-13)
                U.assert_close(got, ref, GRAD_TOL, f"grad {pre}.{k}")
            else:
                err, own = (got.detach().cpu() - ref).abs().max().item(), ref.abs().max().item()
                if err > GRAD_TOL * own:        # under the absolute floor but NOT within the per-tensor tolerance: the floor decided
                    routed.append((f"{pre}.{k}", "ROUNDING floor", err, own))
    for what, route, err, own in routed:
        print(f"[{name} fused={fused}] grad {what}: passed by {route}: abs err {err:.3e}, own max {own:.3e} (rel {err / own:.3e}), "
              f"step max {top:.3e}")
    # the floor is for tensors orders of magnitude below the step's scale -- a tensor that needs it while being within 1e4 of the
    # largest gradient is a regression, not rounding
    assert all(own < 1e-4 * top for _, route, _, own in routed if route == "ROUNDING floor"), routed


# gradient tensors checked against the step's gradient scale instead of their own (see test_end_to_end_vs_reference_golden)
_SCALED_GRADS = {"g9_e2e_elu.npz": ("enc.input_func_node.weights.(0, 0)", "enc.input_func_node.weights.(1, 1)")}
ROUNDING = 64 * 2.2e-16       # absolute differences below this x the step's largest gradient entry are rounding of that entry's scale


@pytest.mark.parametrize("name,B,N,maxdim,che,chd", [("cfg2", 512, 30, 2, (3, 3, 4, 4), (4, 4, 3, 3)),
                                                     ("cfg4", 256, 150, 2, (3, 3, 4, 4), (4, 4, 3, 3)),
                                                     ("cfg5", 512, 30, 3, (4, 4, 6, 6), (6, 6, 4, 4))])
def test_full_size_batch_properties(dev, O, name, B, N, maxdim, che, chd):
    """BASELINE sizes (cfg2: bs=512 N=30; cfg4: bs=256 N=150; cfg5: bs=512 maxdim=3): jets are independent graphs, so
    (1) any jet's output inside the big batch equals its output in a batch of its own, (2) permuting the jets permutes the
    outputs, and (3) parameter gradients of the batch equal the sum of gradients of its two halves (the loss is a sum)."""
    import __graft_entry__ as G
    enc, dec = G._models(N, che, chd, dev, seed=5, maxdim=maxdim)
    p4, labels = O.synthetic_jets(B, N, seed=9, pad=True)
    p4d = p4.to(dev)

    def run(idx):
        for m in (enc, dec):
            m.zero_grad()
        rec = dec(enc({"p4": p4[idx], "labels": labels[idx]}))
        loss = O.chamfer_loss(rec[0] + rec[1], p4d[idx.to(dev)])
        loss.backward()
        grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten()
                           for m in (enc, dec) for p in m.parameters()])
        return rec.detach(), loss.detach(), grads

    all_idx = torch.arange(B)
    rec, loss, grads = run(all_idx)
    assert torch.isfinite(rec).all() and torch.isfinite(grads).all()
    sub = torch.tensor([0, 17, B // 2 - 1, B - 1])
    rec_s, _, _ = run(sub)
    assert (rec[:, sub.to(dev)] - rec_s).abs().max().item() <= 1e-13 * rec.abs().max().item()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    rec_p, loss_p, grads_p = run(perm)
    assert (rec[:, perm.to(dev)] - rec_p).abs().max().item() <= 1e-13 * rec.abs().max().item()
    U.assert_close(loss_p, loss, 1e-12, "loss under jet permutation")
    U.assert_close(grads_p, grads, 1e-9, "grads under jet permutation")
    _, l1, g1 = run(all_idx[:B // 2])
    _, l2, g2 = run(all_idx[B // 2:])
    U.assert_close(l1 + l2, loss, 1e-12, "loss additivity")
    U.assert_close(g1 + g2, grads, 1e-9, "gradient additivity over jets")


def test_bad_arguments_fail_loudly(dev):
    """Host-side argument checks: negative return code surfaced as RuntimeError, no launch."""
    from lgn import _native as N
    s = torch.zeros(2, 1, 4, 9, device=dev, dtype=torch.float64)      # C = 9 unsupported
    v = torch.zeros(2, 1, 4, 9, 4, device=dev, dtype=torch.float64)
    p = torch.zeros(2, 1, 4, 4, device=dev, dtype=torch.float64)
    b = torch.zeros(9, device=dev, dtype=torch.float64)
    w = torch.zeros(2, 2, 45, device=dev, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="unsupported"):
        N.level_fwd(True, s, v, p, None, (None, None, None, None, b, None, b), w, w)
    with pytest.raises(RuntimeError, match="CPU"):
        N.mixreps_fwd(torch.zeros(2, 1, 1, dtype=torch.float64), torch.zeros(2, 3, 1, 1, dtype=torch.float64))


@pytest.mark.parametrize("decoder", [False, True])
def test_static_local_kernels_match_runtime_table_kernels(dev, decoder):
    """The compile-time-table kernels of generic_local_static.hip (tile-blocked layouts; what the native maxdim-3 sequencers
    run) against the run-time-table kernels of generic_local.hip (oracle-tested above), level by level, through the C ABI:
    both level kinds, every padded output width COT in {4, 6, 8} (the lane-sum butterflies of 8 / 12 / 16 values), channel
    counts 2..8, and a node count that leaves the last 64-node tile partly empty."""
    import ctypes
    import __graft_entry__ as G
    from lgn import _native as Nn
    from lgn.plan import static_kind
    enc, dec = G._models(30, (2, 4, 7, 8), (8, 6, 5, 3), dev, seed=3, maxdim=3)
    net = dec if decoder else enc
    B, N = 5, 30
    M = B * N
    tiles = (M + 63) // 64
    Mp = tiles * 64
    g = torch.Generator().manual_seed(11 + int(decoder))

    def to_tb(t2):          # [2][M][C][K] -> [tile][C][K][2][64]
        z = torch.zeros(2, Mp, t2.shape[2], t2.shape[3], device=dev, dtype=torch.float64)
        z[:, :M] = t2
        return z.reshape(2, tiles, 64, t2.shape[2], t2.shape[3]).permute(1, 3, 4, 0, 2).contiguous()

    def from_tb(tb, K):     # [tile][C][K][2][64] -> [2][M][C][K]
        return tb.permute(3, 0, 4, 1, 2).reshape(2, Mp, tb.shape[1], K)[:, :M]

    seen = set()
    for lvl in range(3):
        tables, plan = net.level_tables(lvl), net.plans[lvl]
        C, CO, Q, Qo = plan.channels_in, plan.channels_out, tables.meta["Q"], tables.meta["Qout"]
        kind = static_kind(tables.meta)
        assert kind in (1, 2), "every maxdim-3 level must match one of the two generated table sets"
        seen.add((kind, 4 if CO <= 4 else 6 if CO <= 6 else 8))
        X = torch.randn(2, B, N, C, Q, dtype=torch.float64, generator=g).to(dev)
        Um = torch.randn(B, N, C, Q, 5, 2, dtype=torch.float64, generator=g).to(dev)
        gout = torch.randn(2, B, N, CO, Qo, dtype=torch.float64, generator=g).to(dev)
        mix = net.lgn_cg.node_levels[lvl].cat_mix.mix_reps
        wcat = torch.cat([mix.weight(r).detach().reshape(-1) for r in tables.meta["out_irreps"]]).contiguous()
        wcat = wcat + 0.3 * torch.randn(wcat.shape, dtype=torch.float64, generator=g).to(dev)     # (the init gain is tiny)
        ref_out = Nn.local_fwd(tables, CO, X, Um, wcat)
        ref_gU, ref_gX, ref_gw = Nn.local_bwd(tables, CO, X, Um, wcat, gout)

        XT = to_tb(X.reshape(2, M, C, Q))
        UT = to_tb(Um.reshape(M, C, Q * 5, 2).permute(3, 0, 1, 2))
        goT = to_tb(gout.reshape(2, M, CO, Qo))
        outT = torch.zeros(tiles, CO, Qo, 2, 64, device=dev, dtype=torch.float64)
        w0 = (ctypes.c_int * 5)(*tables.meta["ints"]["out_w0"])
        npk = Nn.lib().lgn_local_static_packed_doubles(kind, C, CO)
        wp = torch.empty(npk, device=dev, dtype=torch.float64)
        Nn._check(Nn.lib().lgn_local_fwd_static_f64(kind, M, C, CO, Nn.ptr(XT), Nn.ptr(UT), Nn.ptr(wcat), w0, Nn.ptr(wp), Nn.ptr(outT),
                                                    None, -1, Nn.stream_ptr()), "lgn_local_fwd_static_f64")
        U.assert_close(from_tb(outT, Qo).reshape(2, B, N, CO, Qo), ref_out, 1e-12, f"level {lvl} (kind {kind}, {C}->{CO}) forward")
        gUT, gXT = torch.full_like(UT, float("nan")), torch.full_like(XT, float("nan"))      # every entry must be written
        part = torch.empty(tiles, npk, device=dev, dtype=torch.float64)
        gpk = torch.empty(npk, device=dev, dtype=torch.float64)
        gw = torch.zeros_like(wcat)
        Nn._check(Nn.lib().lgn_local_bwd_static_f64(kind, M, C, CO, Nn.ptr(XT), Nn.ptr(UT), Nn.ptr(wcat), w0, Nn.ptr(wp), Nn.ptr(goT),
                                                    Nn.ptr(gUT), Nn.ptr(gXT), Nn.ptr(part), Nn.ptr(gpk), Nn.ptr(gw), Nn.stream_ptr()),
                  "lgn_local_bwd_static_f64")
        got_gU = from_tb(gUT, Q * 5).reshape(2, M, C, Q, 5).permute(1, 2, 3, 4, 0).reshape(B, N, C, Q, 5, 2)
        U.assert_close(got_gU, ref_gU, 1e-12, f"level {lvl} (kind {kind}, {C}->{CO}) d U")
        U.assert_close(from_tb(gXT, Q).reshape(2, B, N, C, Q), ref_gX, 1e-12, f"level {lvl} d X")
        U.assert_close(gw, ref_gw, 1e-12, f"level {lvl} d W")
    assert seen == ({(1, 6), (2, 6), (2, 4)} if decoder else {(1, 4), (2, 8)}), seen


@pytest.mark.parametrize("tag,kw", [
    ("mlp_false", dict(mlp=False)),
    ("scale_elu", dict(scale=0.5, activation="elu")),
    ("jet_features_maxdim3", dict(jet_features=True, maxdim=3)),
    ("sigmoid_maxdim3_wide", dict(activation="sigmoid", maxdim=3, ch_enc=(4, 4, 6, 6), ch_dec=(6, 6, 4, 4), N=30, B=2)),
    ("logsigmoid_n150", dict(activation="logsigmoid", N=150, B=1, ch_enc=(3, 3, 4, 4), ch_dec=(4, 4, 3, 3)))])
def test_module_option_combinations_vs_oracle(dev, O, tag, kw):
    """Constructor options no reference fixture combines (mlp=False, scale with a non-default activation, jet features at maxdim 3,
    non-default activations in the wide CGMLP kernels and at N = 150): module API on the GPU against the oracle, forward and every
    parameter gradient (gradient tensors far below the step's gradient scale are held to 1 % of that scale)."""
    from lgn.models import LGNEncoder, LGNDecoder
    N, B, maxdim = kw.get("N", 12), kw.get("B", 3), kw.get("maxdim", 2)
    ch_enc, ch_dec = kw.get("ch_enc", (2, 2, 3, 3)), kw.get("ch_dec", (3, 3, 2, 2))
    act, mlp, scale, jf = kw.get("activation", "leakyrelu"), kw.get("mlp", True), kw.get("scale", 1.0), kw.get("jet_features", False)
    torch.manual_seed(11)
    common = dict(maxdim=[maxdim], max_zf=[1], weight_init="randn", level_gain=[1.0], num_basis_fn=10, activation=act, mlp=mlp,
                  mlp_depth=6, mlp_width=6, device=dev, dtype=torch.float64)
    enc = LGNEncoder(num_input_particles=N, tau_input_scalars=1, tau_input_vectors=1, map_to_latent="min&max", tau_latent_scalars=1,
                     tau_latent_vectors=8, num_channels=list(ch_enc), scale=scale, jet_features=jf, **common)
    dec = LGNDecoder(tau_latent_scalars=2, tau_latent_vectors=16, num_output_particles=N, tau_output_scalars=1, tau_output_vectors=1,
                     num_channels=list(ch_dec), cg_dict=enc.cg_dict, **common)
    oc = dict(num_particles=N, maxdim=maxdim, activation=act, mlp=mlp)
    ce = O.NetConfig(num_channels=tuple(ch_enc), jet_features=jf, scale=scale, **oc)
    cd = O.NetConfig(num_channels=tuple(ch_dec), **oc)
    Pe = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    Pd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
    p4, labels = O.synthetic_jets(B, N, seed=5, pad=True)
    rec = dec(enc({"p4": p4, "labels": labels}))
    loss = O.chamfer_loss(rec[0] + rec[1], p4.to(dev))
    loss.backward()
    rec_o = O.decoder_forward(Pd, cd, O.encoder_forward(Pe, ce, p4, labels))
    loss_o = O.chamfer_loss(rec_o[0] + rec_o[1], p4)
    loss_o.backward()
    U.assert_close(rec, rec_o.detach(), FWD_TOL, f"{tag} recon")
    U.assert_close(loss.detach(), loss_o.detach(), FWD_TOL, f"{tag} loss")
    grads = [(f"{pre}.{k}", g, P[k].grad) for pre, mod, P in (("enc", enc, Pe), ("dec", dec, Pd)) for k, g in mod.named_grads()]
    # strict per-tensor tolerance; the named cancellation survivors (see _SCALED_GRADS) against the step's gradient scale
    floor = 1e-2 * max(float(r.abs().max()) for _, _, r in grads if r is not None)
    scaled = _SCALED_OPTION_GRADS.get(tag, ())
    for name, g, r in grads:
        if r is None or r.abs().max() == 0:
            assert g.abs().max() == 0, f"{tag} {name}: must have exactly zero gradient"
        elif name in scaled:
            U.assert_close_scaled(g, r, GRAD_TOL, floor, f"{tag} grad {name}")
        else:
            U.assert_close(g, r, GRAD_TOL, f"{tag} grad {name}")


# (ELU again: the encoder's input mixing weights, the survivors of a cancellation -- 1.2e-9 of their own size, 1e-12 of the step's)
_IN0 = ("enc.input_func_node.weights.(0, 0)", "enc.input_func_node.weights.(1, 1)")
_SCALED_OPTION_GRADS = {"scale_elu": _IN0, "sigmoid_maxdim3_wide": _IN0}
