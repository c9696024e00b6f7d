"""CPU-side tests of the product's host logic (no GPU compute): the C-ABI library loads and exports
every symbol include/lgn_amd.h declares, CG tables / layout plans / module tree match the reference's
fixtures, and the product refuses to run without a GPU."""
import os
import re

import pytest
import torch

import _util as U

ROOT = U.ROOT


def test_library_exports_every_declared_symbol():
    from lgn import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as G
        G.build()
    lib = _native.lib()
    header = open(os.path.join(ROOT, "include", "lgn_amd.h")).read()
    declared = set(re.findall(r"\b(lgn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(lib, sym), f"liblgn_amd.so does not export {sym}"
    assert set(_native.EXPORTED_SYMBOLS) == declared
    assert lib.lgn_abi_version() == _native.ABI_VERSION


def test_host_side_argument_errors_need_no_gpu():
    from lgn import _native
    lib = _native.lib()
    rc = lib.lgn_level_fwd_f64(1, 1, 1, 1, 0, *([None] * 18))
    assert rc < 0 and b"null" in lib.lgn_last_error()
    rc = lib.lgn_reduce_partials_f64(None, 0, 0, None, 0, None)
    assert rc < 0


def test_cg_tables_match_reference_dump():
    from lgn.cg_lib import CGDict
    z = U.load("g5_tables.npz")
    cg = CGDict(maxdim=3)
    n = 0
    for (r1, r2), entry in cg.items():
        for r, mat in entry.items():
            U.assert_close(mat, z[f"cg.{r1}.{r2}.{r}"], 2e-15, f"cg {r1}x{r2}->{r}")
            n += 1
    assert n == len([k for k in z.files if k.startswith("cg.")])


def _models(meta):
    import __graft_entry__ as G
    return G._models(meta["N"], meta["ch_enc"], meta["ch_dec"], torch.device("cpu"), seed=meta["seed"])


def test_module_tree_and_init_match_reference():
    z = U.load("g1_e2e_maxdim2.npz")
    enc, dec = _models(U.meta(z))
    for mod, pre in ((enc, "enc"), (dec, "dec")):
        ref = U.params_from(z, pre)
        sd = mod.state_dict()
        assert list(sd.keys()) == list(ref.keys()), "state_dict keys / order differ from the reference"
        for k in ref:
            assert tuple(sd[k].shape) == tuple(ref[k].shape)
            assert torch.equal(sd[k], ref[k]), f"same seed must give the reference's initial {k}"
        mod.load_state_dict(ref)
    assert enc.num_learnable_parameters == 34146 and dec.num_learnable_parameters == 29342
    assert enc.cg_dict is dec.cg_dict and enc.maxdim == 2
    assert float(enc.l1_norm() + dec.l1_norm()) == pytest.approx(float(z["l1_norm"]), rel=1e-13)


def test_layout_plan_matches_reference_key_orders():
    from lgn.plan import MAXDIM2_BLOCKS, build_level_plans, param_key_order
    z = U.load("g1_e2e_maxdim2.npz")
    enc, dec = _models(U.meta(z))
    import json
    for i, plan in enumerate(enc.plans):
        want = [tuple(k) for k in json.loads(str(z[f"enc_nodes.{i}.__keys__"]))]
        assert plan.node_order == want
        assert plan.cat_blocks == MAXDIM2_BLOCKS
    assert enc.plans[-1].out_order == [tuple(k) for k in json.loads(str(z["enc_nodes.3.__keys__"]))]
    # maxdim=3 bookkeeping (SURVEY 8 a-3'): node key order and the number of contributing pairs
    z3 = U.load("g2_e2e_maxdim3.npz")
    tau0 = {(0, 0): 4, (1, 1): 4}
    plans = build_level_plans([4, 4, 6, 6], [3, 3, 3], [1, 1, 1], True, tau0, param_key_order(list(tau0)))
    for i, plan in enumerate(plans):
        assert plan.node_order == [tuple(k) for k in json.loads(str(z3[f"enc_nodes.{i}.__keys__"]))]
    assert plans[-1].out_order == [tuple(k) for k in json.loads(str(z3["enc_nodes.3.__keys__"]))]
    assert [b[1:] for b in plans[1].cat_blocks[(1, 1)] if b[0] == "ag"] == \
        [((1, 1), (0, 0)), ((2, 0), (1, 1)), ((0, 2), (1, 1)), ((2, 2), (1, 1)), ((0, 0), (1, 1))]
    assert len([b for b in plans[1].cat_blocks[(2, 2)] if b[0] == "sq"]) == 10
    for key, ref in U.params_from(z3, "enc").items():
        mm = re.match(r"lgn_cg\.node_levels\.(\d)\.cat_mix\.mix_reps\.weights\.\((\d), (\d)\)", key)
        if mm:
            lvl, k = int(mm.group(1)), (int(mm.group(2)), int(mm.group(3)))
            assert ref.shape[2] == plans[lvl].tau_cat[k] == len(plans[lvl].cat_blocks[k]) * plans[lvl].channels_in


def test_no_cpu_fallback():
    z = U.load("g1_e2e_maxdim2.npz")
    enc, dec = _models(U.meta(z))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.from_numpy(z["p4"]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(U.rep_from(z, "latent"))


def test_glue_matches_reference_geometry_and_pooling():
    """The PyTorch glue around the kernels (basis changes, pooling) against the reference's vectors."""
    from lgn import ops
    z = U.load("g4_ops.npz")
    p = torch.from_numpy(z["geo.p"]); pc = torch.from_numpy(z["geo.pc"])
    U.assert_close(ops.cart_to_canonical_real(p).unsqueeze(-2), z["geo.p_to_rep"], 1e-15)
    U.assert_close(ops.cart_to_canonical_cplx(pc), z["geo.p_cplx_to_rep"], 1e-15)
    U.assert_close(ops.canonical_to_cart(pc), z["geo.rep_to_p"], 1e-15)
    U.assert_close(ops.normsq4(p), z["geo.normsq4"], 1e-15)
    lat = U.rep_from(z, "pool.in")
    for method in ("min", "max", "min&max", "mean", "sum", "min+max"):
        U.assert_rep_close(ops.aggregate_latent(method, lat), U.rep_from(z, f"pool.{method}"), 1e-15, method)


def test_lorentz_D_matches_reference_matrices():
    """Representation matrices used by the equivariance harness vs the reference's LorentzD (3 rotations, 3 boosts)."""
    from lgn.cg_lib import CGDict
    from lgn.models.autotest import lorentz_D, cartesian_lorentz
    z = U.load("g5_tables.npz")
    cg = CGDict(maxdim=3)
    for i in range(6):
        ang = [complex(a[0], a[1]) for a in z[f"lorentzD.{i}.angles"]]
        for k in range(3):
            for n in range(3):
                U.assert_close(lorentz_D((k, n), *ang, cg), z[f"lorentzD.{i}.({k}, {n})"], 1e-13, f"D{i} ({k},{n})")
    # a boost along z with rapidity 2 in Cartesian coordinates; metric preserved
    R = cartesian_lorentz(lorentz_D((1, 1), 0, 0, 2.0j, cg))
    eta = torch.diag(torch.tensor([1.0, -1, -1, -1], dtype=torch.float64))
    U.assert_close(R @ eta @ R.t(), eta, 1e-13, "R eta R^T")
    assert R[0, 0].item() == pytest.approx(3.7621956910836314, rel=1e-13)


def test_flat_parameter_block_keeps_the_reference_state_dict_surface():
    """One flat nn.Parameter per network (lgn/models/common.py): parameters() is a single leaf, state_dict() /
    load_state_dict() keep the reference's keys, order and strictness, views follow .to() / deepcopy / re-homing."""
    import copy
    import __graft_entry__ as G
    enc, dec = G._models(30, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0)
    for mod, n in ((enc, 34146), (dec, 29342)):
        assert [k for k, _ in mod.named_parameters()] == ["flat_params"] and mod.flat_params.numel() == n
        assert mod.num_learnable_parameters == n
        sd = mod.state_dict()
        assert sum(v.numel() for v in sd.values()) == n
        assert all(v.untyped_storage().data_ptr() == mod.flat_params.untyped_storage().data_ptr() for v in sd.values())
        new = {k: torch.full_like(v, float(i)) for i, (k, v) in enumerate(sd.items())}
        mod.load_state_dict(new)
        for i, (k, v) in enumerate(mod.named_parameter_views()):
            assert (v == float(i)).all(), k
        assert mod.lgn_cg.mlp_levels[1].linear[3].weight.eq(float(list(sd).index("lgn_cg.mlp_levels.1.linear.3.weight"))).all()
        with pytest.raises(RuntimeError, match="Missing key"):
            mod.load_state_dict({k: v for k, v in list(new.items())[1:]})
        with pytest.raises(RuntimeError, match="Unexpected key"):
            mod.load_state_dict(dict(new, extra=torch.zeros(1)))
        with pytest.raises(RuntimeError, match="size mismatch"):
            mod.load_state_dict(dict(new, **{list(new)[0]: torch.zeros(3)}))
        twin = copy.deepcopy(mod)
        twin.flat_params.data.zero_()
        assert twin.input_func_node.weight((1, 1)).abs().max() == 0 and mod.input_func_node.weight((1, 1)).abs().max() > 0
        mod.double()                                   # nn.Module._apply path: views must follow the (possibly new) storage
        mod.flat_params.data.fill_(2.5)
        assert (mod.rad_funcs.rad_funcs[0].a == 2.5).all()
        assert float(mod.l1_norm()) == 2.5 * n
        opt = torch.optim.Adam(mod.parameters(), 1e-3)   # what utils/initialize.py:156-158 builds
        mod.l1_norm().backward()
        opt.step()
        assert all(g is not None and (g == 1).all() for _, g in mod.named_grads())
        assert (mod.rad_funcs.rad_funcs[0].a < 2.5).all()


def test_networks_nested_in_a_wrapper_round_trip_state_dict_and_deepcopy_after_use():
    """(i) An autoencoder wrapper holding both networks: ``parent.state_dict()`` emits the reference keys under the child
    prefixes and ``parent.load_state_dict`` takes them back (nn.Module walks children through _load_from_state_dict, not
    through the networks' own load_state_dict), with strict-mode errors intact.  (ii) deepcopy of a network that has already
    run on the GPU: its native cache holds ctypes descriptors (pointers: not copyable) and must be skipped, the copy's views
    must point into the copy's own flat block."""
    import copy
    import ctypes
    import __graft_entry__ as G

    class Wrapper(torch.nn.Module):
        def __init__(self, e, d):
            super().__init__()
            self.e, self.d = e, d

    a = Wrapper(*G._models(30, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0))
    b = Wrapper(*G._models(30, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=5))
    sd = a.state_dict()
    assert list(sd)[0] == "e.rad_funcs.rad_funcs.0.a" and "e.flat_params" not in sd and len(sd) == len(a.e.state_dict()) + len(a.d.state_dict())
    assert not torch.equal(a.e.flat_params, b.e.flat_params)
    res = b.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(a.e.flat_params, b.e.flat_params) and torch.equal(a.d.flat_params, b.d.flat_params)
    bad = dict(sd)
    gone = list(bad)[3]
    del bad[gone]
    bad["d.bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Missing key"):
        b.load_state_dict(bad)
    res = b.load_state_dict(bad, strict=False)
    assert res.missing_keys == [gone] and res.unexpected_keys == ["d.bogus"]
    with pytest.raises(RuntimeError, match="size mismatch"):
        b.load_state_dict(dict(sd, **{gone: torch.zeros(7)}))
    assert list(a.e.state_dict(None, "pre.", False))[0] == "pre.rad_funcs.rad_funcs.0.a"      # positional form
    # what a first GPU forward leaves behind (NetHandle: byref + descriptor with pointer fields; DeviceTables likewise)
    a.e.__dict__["_native_cache"] = {4: ctypes.byref(ctypes.c_int(3))}
    a.e.__dict__["_level_tables"] = {0: ctypes.pointer(ctypes.c_int(3))}
    twin = copy.deepcopy(a)
    assert "_native_cache" not in twin.e.__dict__ and "_level_tables" not in twin.e.__dict__
    assert twin.e.flat_params.data_ptr() != a.e.flat_params.data_ptr() and torch.equal(twin.e.flat_params, a.e.flat_params)
    assert twin.e._p_views[0].data_ptr() == twin.e.flat_params.data_ptr()
    twin.e.flat_params.data.zero_()
    assert twin.e.input_func_node.weight((1, 1)).abs().max() == 0 and a.e.input_func_node.weight((1, 1)).abs().max() > 0


def test_native_step_refuses_configurations_it_does_not_implement():
    """lgn_step_fwd_bwd_f64 is the maxdim = 2 closed form; a maxdim = 3 (or non min&max) network must be refused on the
    host instead of being read with the wrong weight layout (checked before any GPU requirement).  (maxdim = 3 networks are
    covered by the table-driven native step since round 2.)"""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep
    enc, _ = G._models(30, (4, 4, 6, 6), (6, 6, 4, 4), torch.device("cpu"), seed=0, maxdim=3)
    _, dec = G._models(30, (4, 4, 6, 6), (6, 6, 4, 4), torch.device("cpu"), seed=0, maxdim=2)
    with pytest.raises(NotImplementedError, match="same kind"):       # mixed fused / table-driven networks
        NativeTrainStep(enc, dec, batch_size=4)
    enc, dec = G._models(30, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0)
    from lgn.nn import RadPolyTrig
    r11 = RadPolyTrig(1, 11, 3)                                         # more bells than one group of the kernels' 20: whole groups, zero padded
    kp = r11.kernel_params()
    assert r11.kernel_width == 40 and [tuple(t.shape) for t in kp] == [(1, 1, 1, 40)] * 3 + [(6, 40), (6,), (6, 40), (6,)]
    assert all(float(t[..., 22:].abs().max()) == 0 for t in (kp[0], kp[1], kp[2], kp[3], kp[5])) and torch.equal(kp[0][..., :22], r11.a)
    assert RadPolyTrig(1, 20, 3).kernel_params()[0].shape[-1] == 40 and RadPolyTrig(1, 10, 3).kernel_width == 20
    with pytest.raises(ValueError, match="num_basis_fn"):
        RadPolyTrig(1, 0, 3)
    r5 = RadPolyTrig(1, 5, 3)                                           # fewer: zero-padded to the kernels' 20 on the per-operator path
    kp = r5.kernel_params()
    assert [tuple(t.shape) for t in kp] == [(1, 1, 1, 20)] * 3 + [(6, 20), (6,), (6, 20), (6,)]
    assert all(float(t[..., 10:].abs().max()) == 0 for t in (kp[0], kp[1], kp[2], kp[3], kp[5]))
    kp[3].sum().backward()                                              # the padding is differentiable: gradients reach the 10 real columns
    assert tuple(r5.linear[0].weight.grad.shape) == (6, 10)
    from lgn.nn import CGMLP
    with pytest.raises(NotImplementedError, match="mlp_width"):         # hidden width 7 x 16 = 112 > 96: refused at construction
        CGMLP(8, num_hidden=6, layer_width_mul=7)
    assert CGMLP(8, num_hidden=6, layer_width_mul=6).width == 96 and CGMLP(3, num_hidden=6, layer_width_mul=5).width == 30
    enc.mlp_depth = 2                                                   # the native CGMLP kernels are built for mlp_depth 3 .. 6
    enc.__dict__.pop("_native_kind", None)
    with pytest.raises(NotImplementedError, match="mlp_depth"):
        NativeTrainStep(enc, dec, batch_size=4)
    # min / max / mean poolings, their '&' / '+' combinations and the learned 'mix' map are native (lgn_net_desc.latent_pool, round 4);
    # 'sum' (an extra axis in the reference) is not
    from lgn import _native as Nn
    assert Nn.pool_code("min&max") == 2 | (1 << 6) and Nn.pool_blocks(0) == Nn.pool_blocks(Nn.pool_code("min&max")) == 2
    assert Nn.pool_code("Mean+Max") == 2 | (1 << 3) | (2 << 4) | (1 << 6) and Nn.pool_blocks(Nn.pool_code("mean+max")) == 1
    assert Nn.pool_blocks(Nn.pool_code("mean&min&max")) == 3
    assert Nn.pool_code("mix") == 1 | (3 << 4) and Nn.pool_blocks(Nn.pool_code("mix")) == 1
    assert Nn.pool_code("sum") is None and Nn.pool_code("mean&min+max") is None and Nn.pool_code("mix&max") is None
    assert Nn.pool_code("min&max&mean&min&max") is None
    enc, dec = G._models(12, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0, map_to_latent="mean+max")
    assert enc._fused_ok()
    enc, dec = G._models(12, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0, map_to_latent="mix")
    assert enc._fused_ok() and tuple(enc.mix_reps.weight((1, 1)).shape) == (2, 8, 12 * 4)
    for latent in ("sum",):
        enc, dec = G._models(12, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0, map_to_latent=latent)
        assert not enc._fused_ok()
        with pytest.raises(NotImplementedError, match="map_to_latent"):
            NativeTrainStep(enc, dec, batch_size=4)
    # jet features / extra input scalars: the per-network native calls take them (lgn_net_desc.n_in_scalars, round 4) and, for
    # maxdim = 2 networks, the whole-step call (round 6: lgn_net_desc.dec_N -- the encoder has one node more here); table-driven
    # networks with jet features are refused at plan time (-> CapturedModuleStep)
    enc, dec = G._models(12, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0, jet_features=True)
    assert enc.num_input_particles == 13 and enc.tau_input_scalars == 2 and enc._fused_ok()
    assert tuple(enc.input_func_node.weight((0, 0)).shape) == (2, 3, 2)
    enc, dec = G._models(12, (3, 3, 4, 4), (4, 4, 3, 3), torch.device("cpu"), seed=0, jet_features=True, maxdim=3)
    with pytest.raises(NotImplementedError, match="jet_features"):
        NativeTrainStep(enc, dec, batch_size=4)


def test_chamfer_loss_module_has_no_cpu_path():
    """lgn.losses.ChamferLoss keeps the reference's constructor / forward signature (utils/losses/chamfer_loss/chamfer_loss.py:7-16)
    and fails loudly without a GPU instead of computing on the host."""
    import inspect
    from lgn.losses import ChamferLoss
    assert list(inspect.signature(ChamferLoss.__init__).parameters) == ["self", "device"]
    assert list(inspect.signature(ChamferLoss.forward).parameters) == ["self", "x", "y", "jet_features"]
    loss = ChamferLoss(device=torch.device("cpu"))
    x = torch.randn(2, 5, 4, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loss(x, x)
    with pytest.raises(ValueError, match="4-vectors"):
        loss(x[..., :3], x[..., :3])


def test_generated_static_tables_are_up_to_date_and_match_the_runtime_matcher():
    """csrc/cg_static_tables.hpp is generated from lgn.plan.build_local_tables (tools/gen_static_tables.py); the committed
    file must equal a fresh rendering, and plan.static_kind must recognise exactly the levels it was generated from."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_static_tables", os.path.join(ROOT, "tools", "gen_static_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(gen.OUT) as fh:
        assert fh.read() == gen.render(), "run tools/gen_static_tables.py"
    from lgn.plan import canonical_static_tables, static_kind
    for kind, tab in canonical_static_tables().items():
        assert static_kind(tab) == kind


def test_fewer_radial_bells_are_stored_padded_and_seen_unpadded():
    """num_basis_fn < 10 (lgn/nn/position_levels.py:44-64): the flat parameter block STORES a, b, c and the radial Linear weights 20
    bells wide (what every kernel reads), zero padded; names, shapes, state_dict, load_state_dict, the parameter count and deepcopy see
    the reference's tensors (narrow views of the stored blocks)."""
    import copy
    import __graft_entry__ as G
    enc, dec = G._models(12, (2, 3, 4), (4, 3, 2), torch.device("cpu"), seed=3, num_basis_fn=5)
    ref, _ = G._models(12, (2, 3, 4), (4, 3, 2), torch.device("cpu"), seed=3, num_basis_fn=10)
    sd = enc.state_dict()
    assert tuple(sd["rad_funcs.rad_funcs.0.a"].shape) == (1, 1, 1, 10)
    assert tuple(sd["rad_funcs.rad_funcs.1.linear.0.weight"].shape) == (6, 10) and tuple(sd["rad_funcs.rad_funcs.1.linear.0.bias"].shape) == (6,)
    assert list(sd) == [k for k in ref.state_dict()], "same keys, same order as with the default 10"
    true_count = sum(v.numel() for v in sd.values())
    assert enc.num_learnable_parameters == true_count
    nrad = 2 * (3 * 10 + 4 * 10 + 6 * 10 + 3 * 10 + 2 * 2 * 10)       # padding columns: per level 3 x 10 (a, b, c) + 2 x 2C x 10 (weights)
    assert enc.flat_params.numel() == true_count + (3 * 10 * 2 + 2 * (4 + 6) * 10), (enc.flat_params.numel(), true_count, nrad)
    rf = enc.rad_funcs.rad_funcs[0]
    kp = rf.kernel_params()
    assert [tuple(t.shape) for t in kp] == [(1, 1, 1, 20)] * 3 + [(4, 20), (4,), (4, 20), (4,)]
    assert all(float(t[..., 10:].abs().max()) == 0 for t in (kp[0], kp[1], kp[2], kp[3], kp[5]))
    assert kp[0].data_ptr() == rf.a.data_ptr() and torch.equal(kp[3][:, :10], rf.linear[0].weight)
    # a round trip through state_dict leaves the padding at zero and restores the values
    enc2 = copy.deepcopy(enc)
    with torch.no_grad():
        enc2.flat_params.add_(1.0)                      # (dirties the padding too)
    enc2.load_state_dict(sd)
    for (k, v), (_, w) in zip(enc2.state_dict().items(), sd.items()):
        assert torch.equal(v, w), k
    assert enc2.flat_params.data_ptr() != enc.flat_params.data_ptr()
    # gradients: named_grads are the narrow views of the flat gradient
    enc.flat_params.grad = torch.arange(enc.flat_params.numel(), dtype=torch.float64)
    g = dict(enc.named_grads())
    assert tuple(g["rad_funcs.rad_funcs.0.a"].shape) == (1, 1, 1, 10) and tuple(g["rad_funcs.rad_funcs.0.linear.1.weight"].shape) == (4, 10)
    from lgn.ops import native_kind
    assert native_kind(enc) == "fused" and native_kind(dec) == "fused"
    # more than 20 bells (num_basis_fn 12): stored as two groups of 20, seen 24 wide; only the per-operator path sums over groups
    big, bigd = G._models(12, (2, 3, 4), (4, 3, 2), torch.device("cpu"), seed=3, num_basis_fn=12)
    sdb = big.state_dict()
    assert tuple(sdb["rad_funcs.rad_funcs.0.a"].shape) == (1, 1, 1, 24) and tuple(sdb["rad_funcs.rad_funcs.1.linear.1.weight"].shape) == (6, 24)
    assert big.num_learnable_parameters == sum(v.numel() for v in sdb.values())
    assert big.rad_funcs.rad_funcs[1].kernel_params()[3].shape == (6, 40) and big.rad_funcs.rad_funcs[1].kernel_params()[3].data_ptr() == \
        big.rad_funcs.rad_funcs[1].linear[0].weight.data_ptr()
    assert native_kind(big) is None and native_kind(bigd) is None
    from lgn.step import NativeTrainStep
    with pytest.raises(NotImplementedError, match="num_basis_fn"):
        NativeTrainStep(big, bigd, batch_size=2)
