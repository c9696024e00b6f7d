"""Pins the oracle (oracle/lgn_oracle.py) to golden vectors produced by the reference
(tests/golden/gen_golden.py).  CPU only.  Tolerances are relative to max|ref| (fp64)."""
import pytest
import torch

import _util as U
from oracle import lgn_oracle as O

TOL = 1e-12


def _cfgs(m):
    common = dict(num_particles=m["N"], maxdim=m["maxdim"], map_to_latent=m.get("map_to_latent", "min&max"),
                  activation=m.get("activation", "leakyrelu"), mlp_depth=m.get("mlp_depth", 6), num_basis_fn=m.get("num_basis_fn", 10), mlp_width=m.get("mlp_width", 6))
    return (O.NetConfig(num_channels=tuple(m["ch_enc"]), jet_features=m.get("jet_features", False),
                        tau_input_scalars=1 + m.get("extra_scalars", 0), **common),
            O.NetConfig(num_channels=tuple(m["ch_dec"]), **common))


@pytest.mark.parametrize("name", ["g1_e2e_maxdim2.npz", "g2_e2e_maxdim3.npz", "g3_e2e_n150.npz", "g6_e2e_mix.npz",
                                  "g7_e2e_meanmax.npz", "g9_e2e_elu.npz", "g10_e2e_jetfeat.npz", "g11_e2e_mlpdepth4.npz",
                                  "g11_e2e_mlpdepth3_maxdim3.npz", "g12_e2e_n150_maxdim3.npz", "g13_e2e_basis5.npz",
                                  "g13_e2e_basis5_maxdim3.npz", "g14_e2e_mlpwidth4.npz", "g14_e2e_mlpwidth5.npz", "g14_e2e_mlpwidth7.npz",
                                  "g14_e2e_mlpwidth5_maxdim3.npz", "g15_e2e_basis12.npz", "g15_e2e_basis20_maxdim3.npz"])
def test_end_to_end_forward_backward(name):
    z = U.load(name)
    m = U.meta(z)
    ce, cd = _cfgs(m)
    Pe = {k: v.clone().requires_grad_(True) for k, v in U.params_from(z, "enc").items()}
    Pd = {k: v.clone().requires_grad_(True) for k, v in U.params_from(z, "dec").items()}
    p4 = torch.from_numpy(z["p4"]); labels = torch.from_numpy(z["labels"])
    xs = torch.from_numpy(z["scalars"]) if "scalars" in z.files else None      # data['scalars'] (g10)

    lat, enc_nodes = O.encoder_forward(Pe, ce, p4, labels, covariance_test=True, extra_scalars=xs)
    U.assert_rep_close(lat, U.rep_from(z, "latent"), TOL, "latent")
    for i, rep in enumerate(enc_nodes):
        U.assert_rep_close(rep, U.rep_from(z, f"enc_nodes.{i}"), TOL, f"enc_nodes[{i}]")
    gen, dec_nodes = O.decoder_forward(Pd, cd, lat, covariance_test=True)
    for i, rep in enumerate(dec_nodes):
        U.assert_rep_close(rep, U.rep_from(z, f"dec_nodes.{i}"), TOL, f"dec_nodes[{i}]")

    rec = O.decoder_forward(Pd, cd, O.encoder_forward(Pe, ce, p4, labels, extra_scalars=xs))
    U.assert_close(rec, z["recon"], TOL, "recon")
    real = O.get_real(rec, "sum")
    U.assert_close(real, z["recon_real"], TOL, "recon_real")
    loss = O.chamfer_loss(real, p4)
    U.assert_close(loss, z["loss_chamfer"], TOL, "chamfer")
    l1 = O.l1_norm(Pe) + O.l1_norm(Pd)
    U.assert_close(l1, z["l1_norm"], TOL, "l1")
    U.assert_close(loss + 1e-8 * l1, z["loss_total"], TOL, "total")

    loss.backward()
    for pre, P in (("enc", Pe), ("dec", Pd)):
        for k, p in P.items():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"])
            if ref.abs().max() == 0:
                assert g.abs().max() == 0, f"grad {pre}.{k} must be exactly zero"
            else:
                U.assert_close(g, ref, 1e-9, f"grad {pre}.{k}")


def test_dead_gradients_match_survey():
    """parameters that receive exactly zero data-gradient."""
    z = U.load("g1_e2e_maxdim2.npz")
    dead = ["grad.enc.mix_reps.weights.(0, 0)", "grad.dec.latent_to_graph.weights.(0, 0)",
            "grad.dec.mix_to_output.weights.(0, 0)", "grad.dec.rad_funcs.rad_funcs.0.a",
            "grad.dec.rad_funcs.rad_funcs.1.linear.0.weight", "grad.enc.lgn_cg.mlp_levels.2.linear.3.weight"]
    for k in dead:
        assert abs(z[k]).max() == 0, k


@pytest.mark.parametrize("tag,maxdim", [("cg2_2", 2), ("cg3_5", 3), ("cg3_2", 3)])
def test_cg_product(tag, maxdim):
    z = U.load("g4_ops.npz")
    cg = O.get_cg(maxdim)
    node = U.rep_from(z, tag + ".node"); edge = U.rep_from(z, tag + ".edge")
    U.assert_rep_close(O.cg_product(cg, node, edge, maxdim, aggregate=True), U.rep_from(z, tag + ".aggregate"), TOL, "agg")
    U.assert_rep_close(O.cg_product(cg, node, node, maxdim, aggregate=False), U.rep_from(z, tag + ".power"), TOL, "pow")


def test_cg_tables_and_closed_form():
    z = U.load("g5_tables.npz")
    cg = O.get_cg(3)
    n = 0
    for (r1, r2), entry in cg.table.items():
        for r, mat in entry.items():
            U.assert_close(mat, z[f"cg.{r1}.{r2}.{r}"], 1e-14, f"cg {r1}x{r2}->{r}")
            n += 1
    assert n == 369 or n > 0
    # SURVEY a-2: (1,1)x(1,1)->(0,0) = 1/2 [e00 + e13 - e22 + e31]
    m = cg[((1, 1), (1, 1))][(0, 0)].reshape(4, 4)
    expect = torch.zeros(4, 4, dtype=torch.float64)
    expect[0, 0] = .5; expect[1, 3] = .5; expect[2, 2] = -.5; expect[3, 1] = .5
    U.assert_close(m, expect, 1e-15, "closed form")


def test_geometry():
    z = U.load("g4_ops.npz")
    p = torch.from_numpy(z["geo.p"]); pc = torch.from_numpy(z["geo.pc"])
    U.assert_close(O.p_to_rep(p), z["geo.p_to_rep"], TOL)
    U.assert_close(O.p_cplx_to_rep(pc), z["geo.p_cplx_to_rep"], TOL)
    U.assert_close(O.rep_to_p(pc), z["geo.rep_to_p"], TOL)
    U.assert_close(O.normsq4(p), z["geo.normsq4"], TOL)
    U.assert_close(O.repdot_planar(pc, pc), z["geo.repdot"], TOL)
    zf, nrm, nsq = O.zonal_rel(p, p, "cartesian")
    U.assert_rep_close(zf, U.rep_from(z, "geo.zf_cart"), TOL)
    U.assert_close(nrm, z["geo.zf_cart.norm"], TOL); U.assert_close(nsq, z["geo.zf_cart.normsq"], TOL)
    pcc = O.p_cplx_to_rep(pc)
    zf, nrm, nsq = O.zonal_rel(pcc, pcc, "canonical")
    U.assert_rep_close(zf, U.rep_from(z, "geo.zf_canon"), TOL)
    U.assert_close(nrm, z["geo.zf_canon.norm"], TOL); U.assert_close(nsq, z["geo.zf_canon.normsq"], TOL)


@pytest.mark.parametrize("decoder", [False, True])
def test_radial_filters(decoder):
    z = U.load("g4_ops.npz")
    tag = "rad_dec" if decoder else "rad_enc"
    P = {"rad_funcs." + k: v.clone().requires_grad_(True) for k, v in U.params_from(z, tag + ".param").items()}
    cfg = O.NetConfig(num_channels=(3, 4, 4))
    norms = torch.from_numpy(z[tag + ".norms"]); mask = torch.from_numpy(z[tag + ".mask"])
    tot = 0
    for lvl in range(2):
        out = O.radial_filters(P, cfg, lvl, norms, mask, decoder)
        for k, v in out.items():
            U.assert_close(v, z[f"{tag}.out.{lvl}.{k}"], TOL, f"{tag} {lvl} {k}")
            tot = tot + (v * torch.from_numpy(z[f"{tag}.cot.{lvl}.{k}"])).sum()
    tot.backward()
    for k, p in P.items():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        U.assert_close(g, z[f"{tag}.grad.{k[len('rad_funcs.'):]}"], 1e-11, f"grad {k}")


def test_cgmlp():
    z = U.load("g4_ops.npz")
    P = {"lgn_cg.mlp_levels.0." + k: v for k, v in U.params_from(z, "mlp.param").items()}
    cfg = O.NetConfig(num_channels=(3, 3))
    out = O.cg_mlp(P, cfg, 0, U.rep_from(z, "mlp.in"))
    U.assert_rep_close(out, U.rep_from(z, "mlp.out"), TOL, "cgmlp")


@pytest.mark.parametrize("act", ["relu", "elu", "sigmoid", "logsigmoid", "atan"])
def test_cgmlp_activations(act):
    """Every non-default activation of get_activation_fn (lgn/nn/generic_levels.py:119-135): output and all gradients."""
    z = U.load("g9_activations.npz")
    P = {"lgn_cg.mlp_levels.0." + k: v.clone().requires_grad_(True) for k, v in U.params_from(z, f"{act}.param").items()}
    node = U.rep_from(z, "in")
    node[(0, 0)] = node[(0, 0)].clone().requires_grad_(True)
    s_in = node[(0, 0)]
    out = O.cg_mlp(P, O.NetConfig(num_channels=(4, 4), activation=act), 0, node)
    U.assert_rep_close(out, U.rep_from(z, f"{act}.out"), TOL, f"cgmlp {act}")
    (out[(0, 0)] * torch.from_numpy(z["cot"])).sum().backward()
    U.assert_close(s_in.grad, z[f"{act}.grad_in"], 1e-11, f"{act} grad_in")
    for k, p in P.items():
        U.assert_close(p.grad, z[f"{act}.grad." + k[len("lgn_cg.mlp_levels.0."):]], 1e-11, f"{act} grad {k}")


@pytest.mark.parametrize("method", ["min", "max", "min&max", "mean", "sum", "min+max"])
def test_pooling(method):
    z = U.load("g4_ops.npz")
    out = O.aggregate_latent(method, U.rep_from(z, "pool.in"))
    U.assert_rep_close(out, U.rep_from(z, f"pool.{method}"), TOL, method)


def test_chamfer_and_normalize():
    z = U.load("g4_ops.npz")
    x = torch.from_numpy(z["chamfer.x"]).requires_grad_(True); y = torch.from_numpy(z["chamfer.y"])
    l = O.chamfer_loss(x, y); l.backward()
    U.assert_close(l, z["chamfer.loss"], TOL); U.assert_close(x.grad, z["chamfer.grad_x"], TOL)
    U.assert_close(O.normalize_p4_overall_max(torch.from_numpy(z["normp4.in"])), z["normp4.overall_max"], TOL)


def test_same_seed_same_init_as_reference():
    """The oracle's initialiser consumes the RNG in the reference's order: same seed -> same weights."""
    z = U.load("g1_e2e_maxdim2.npz")
    m = U.meta(z)
    ce, cd = _cfgs(m)
    torch.manual_seed(m["seed"])
    Pe = O.init_encoder_params(ce)
    Pd = O.init_decoder_params(cd, (2, 16))
    ref_e = U.params_from(z, "enc"); ref_d = U.params_from(z, "dec")
    assert set(Pe.keys()) == set(ref_e.keys())   # (state_dict order groups node_levels before mlp_levels)
    assert set(Pd.keys()) == set(ref_d.keys())
    for k in ref_e:
        U.assert_close(Pe[k], ref_e[k], 1e-15, "enc " + k)
    for k in ref_d:
        U.assert_close(Pd[k], ref_d[k], 1e-15, "dec " + k)
