#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by importing the *reference*
implementation (zichunhao/lgn-autoencoder, mounted read-only at /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/tests/golden/gen_golden.py

The outputs are plain data (inputs, parameters, expected outputs, expected gradients, key
orders); no reference source text is stored.  Everything is float64.

Fixtures
  g1_e2e_maxdim2.npz  end-to-end, B=4 N=30 maxdim=2 ch 3344/4433 (two jets zero padded)
  g2_e2e_maxdim3.npz  end-to-end, B=2 N=30 maxdim=3 ch 4466/6644
  g3_e2e_n150.npz     end-to-end, B=1 N=150 maxdim=2
  g4_ops.npz          per-op vectors (cg_product, radial filters, geometry, pooling, chamfer ...)
  g5_tables.npz       CGDict(maxdim=3) coefficient tables (dense) + LorentzD matrices
  g6_e2e_mix.npz      end-to-end, B=3 N=12 maxdim=2 ch 2233/3322, map_to_latent='mix' (learned mixing over particles)
  g7_e2e_meanmax.npz  end-to-end, B=3 N=12, map_to_latent=mean+max
  g9_activations.npz  CGMLP with every non-default activation of get_activation_fn (lgn/nn/generic_levels.py:119-135): output and
                      gradients w.r.t. input and parameters for a fixed cotangent
  g9_e2e_elu.npz      end-to-end, B=3 N=12 maxdim=2 ch 2233/3322, activation='elu'
  g10_e2e_jetfeat.npz end-to-end, B=3 N=12 maxdim=2 ch 2233/3322, jet_features=True (13th node = jet momentum, second input scalar)
                      plus one extra input scalar per node through data['scalars']
  g11_e2e_mlpdepth4.npz / g11_e2e_mlpdepth3_maxdim3.npz  end-to-end with --mlp-depth 4 (maxdim 2, B=3 N=12 ch 2344/4432) and 3 (maxdim 3,
                      B=2 N=10 ch 246/642): CGMLPs of 5 / 4 Linear layers (round 5)
  g14_e2e_mlpwidth{4,5,7}.npz / g14_e2e_mlpwidth5_maxdim3.npz  end-to-end with --mlp-width 4 / 5 / 7 (maxdim 2, B=3 N=12 ch 2344/4432) and 5
                                     (maxdim 3, B=2 N=10 ch 246/642): CGMLP hidden widths other than 6 x 2C
  g13_e2e_basis5.npz / g13_e2e_basis5_maxdim3.npz  end-to-end with --num-basis-fn 5 (10 bells), maxdim 2 (B=3 N=12) and 3 (B=2 N=10)
  g15_e2e_basis12.npz / g15_e2e_basis20_maxdim3.npz  end-to-end with --num-basis-fn 12 (24 bells, maxdim 2, B=3 N=12) and 20 (40 bells,
                      maxdim 3, B=2 N=10): more bells than one group of the kernels' 20 (round 6)
  g12_e2e_n150_maxdim3.npz  end-to-end, B=1 N=150 maxdim=3 ch 4466/6644 (round 5: jets beyond the LDS-resident kernels)
  g8_harness.npz      the reference's own equivariance harness (lgn/models/autotest/lgn_tests.py:292-423) run on the g1 weights
                      (maxdim 2) and the g2 weights (maxdim 3): gamma / theta grids, output and internal-feature deviation
                      tables, permutation results, on fixed zero-padded jets (SURVEY 8c "G6 harness")

`python gen_golden.py g6 g7` regenerates only the named fixtures.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True      # the reference tree is read-only by contract: no __pycache__ beside its sources, however this is run

import numpy as np
import torch

sys.modules.setdefault("jetnet", types.ModuleType("jetnet"))  # EMD wrapper is never instantiated

from lgn.models import LGNEncoder, LGNDecoder  # noqa: E402
from lgn.cg_lib import CGDict  # noqa: E402
from lgn.cg_lib.cg_ops import cg_product  # noqa: E402
from lgn.g_lib import GVec  # noqa: E402
from lgn.g_lib import rotations as rot  # noqa: E402
from lgn.nn import RadialFilters  # noqa: E402
from lgn.models.lgn_encoder import aggregate  # noqa: E402
from lgn.models.lgn_levels import CGMLP  # noqa: E402
from utils.losses.chamfer_loss.chamfer_loss import ChamferLoss  # noqa: E402
from utils.normalize_p4 import normalize_p4  # noqa: E402
from utils.utils import get_real  # noqa: E402

ZF = sys.modules["lgn.cg_lib.zonal_functions"]
OUT = os.path.dirname(os.path.abspath(__file__))
CPU = torch.device("cpu")
F64 = torch.float64


def npy(t):
    return t.detach().cpu().numpy()


def put_rep(store, prefix, rep):
    store[prefix + ".__keys__"] = np.array(json.dumps([list(k) for k in rep.keys()]))
    for k, v in rep.items():
        store[f"{prefix}.{k}"] = npy(v)


def jets(B, N, seed, pad_rows=()):
    g = torch.Generator().manual_seed(seed)
    p3 = torch.randn(B, N, 3, dtype=F64, generator=g)
    e = torch.sqrt((p3 * p3).sum(-1, keepdim=True) + 1e-6)
    p4, _ = normalize_p4(torch.cat([e, p3], -1), "overall_max")
    labels = torch.ones(B, N, dtype=torch.uint8)
    for row, nreal in pad_rows:
        labels[row, nreal:] = 0
    p4 = p4 * labels.unsqueeze(-1).to(F64)
    return p4, labels


def build(N, maxdim, ch_enc, ch_dec, seed, map_to_latent="min&max", activation="leakyrelu", jet_features=False, tau_input_scalars=1,
          mlp_depth=6, num_basis_fn=10, mlp_width=6):
    torch.manual_seed(seed)
    common = dict(maxdim=[maxdim], max_zf=[1], weight_init="randn", level_gain=[1.0], num_basis_fn=num_basis_fn,
                  activation=activation, mlp=True, mlp_depth=mlp_depth, mlp_width=mlp_width, device=CPU, dtype=F64)
    enc = LGNEncoder(num_input_particles=N, tau_input_scalars=tau_input_scalars, tau_input_vectors=1, map_to_latent=map_to_latent,
                     tau_latent_scalars=1, tau_latent_vectors=8, num_channels=list(ch_enc), scale=1.0,
                     jet_features=jet_features, **common)
    mult = len(map_to_latent.split("&"))          # '&' concatenates the pooled features (utils/initialize.py:118-120)
    dec = LGNDecoder(tau_latent_scalars=1 * mult, tau_latent_vectors=8 * mult, num_output_particles=N, tau_output_scalars=1,
                     tau_output_vectors=1, num_channels=list(ch_dec), cg_dict=enc.cg_dict, **common)
    return enc, dec


def e2e(name, B, N, maxdim, ch_enc, ch_dec, seed, pad_rows=(), map_to_latent="min&max", activation="leakyrelu", jet_features=False,
        extra_scalars=0, mlp_depth=6, num_basis_fn=10, mlp_width=6):
    enc, dec = build(N, maxdim, ch_enc, ch_dec, seed, map_to_latent, activation, jet_features, 1 + extra_scalars, mlp_depth, num_basis_fn,
                     mlp_width)
    p4, labels = jets(B, N, seed + 100, pad_rows)
    meta = dict(B=B, N=N, maxdim=maxdim, ch_enc=list(ch_enc), ch_dec=list(ch_dec), seed=seed, l1_lambda=1e-8, map_to_latent=map_to_latent)
    if activation != "leakyrelu":
        meta["activation"] = activation
    if jet_features or extra_scalars:
        meta["jet_features"], meta["extra_scalars"] = bool(jet_features), extra_scalars
    if mlp_depth != 6:
        meta["mlp_depth"] = mlp_depth
    if num_basis_fn != 10:
        meta["num_basis_fn"] = num_basis_fn
    if mlp_width != 6:
        meta["mlp_width"] = mlp_width
    store = {"p4": npy(p4), "labels": npy(labels), "meta": np.array(json.dumps(meta))}
    for k, v in enc.state_dict().items():
        store["enc." + k] = npy(v)
    for k, v in dec.state_dict().items():
        store["dec." + k] = npy(v)

    batch = {"p4": p4, "labels": labels}
    if extra_scalars:       # data['scalars'] (lgn_encoder.py:403-408): one row per node the encoder works on (incl. the jet node)
        gs = torch.Generator().manual_seed(seed + 200)
        batch["scalars"] = torch.randn(B, N + int(jet_features), extra_scalars, dtype=F64, generator=gs)
        store["scalars"] = npy(batch["scalars"])
    latent, nodes_all = enc(batch, covariance_test=True)
    n_enc = len(nodes_all)
    put_rep(store, "latent", latent)
    gen, nodes_all = dec(latent, covariance_test=True, nodes_all=nodes_all)
    for i, rep in enumerate(nodes_all[:n_enc]):
        put_rep(store, f"enc_nodes.{i}", rep)
    for i, rep in enumerate(nodes_all[n_enc:]):
        put_rep(store, f"dec_nodes.{i}", rep)

    # training-step forward/backward (utils/train.py:283-343), Chamfer only and Chamfer + L1
    enc.zero_grad(); dec.zero_grad()
    latent = enc(batch)
    recon = dec(latent)
    real = get_real(recon, "sum")
    chamfer = ChamferLoss(device=CPU)(real, p4)
    l1 = enc.l1_norm() + dec.l1_norm()
    store["recon"] = npy(recon)
    store["recon_real"] = npy(real)
    store["loss_chamfer"] = npy(chamfer)
    store["l1_norm"] = npy(l1)
    store["loss_total"] = npy(chamfer + 1e-8 * l1)
    chamfer.backward()
    for pre, mod in (("enc", enc), ("dec", dec)):
        for k, p in mod.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            store[f"grad.{pre}.{k}"] = npy(g)
    np.savez_compressed(os.path.join(OUT, name), **store)
    print(name, "chamfer", float(chamfer), "n arrays", len(store))


def rand_rep(keys, batch, C, g):
    return GVec({k: torch.randn((2,) + tuple(batch) + (C, (k[0] + 1) * (k[1] + 1)), dtype=F64, generator=g) for k in keys})


def ops():
    g = torch.Generator().manual_seed(7)
    store = {}
    B, N, C = 2, 5, 3
    for maxdim, node_keys in ((2, [(1, 1), (0, 0)]), (3, [(1, 1), (2, 0), (0, 2), (2, 2), (0, 0)]), (3, [(1, 1), (0, 0)])):
        cgd = CGDict(maxdim=maxdim, device=CPU, dtype=F64)
        tag = f"cg{maxdim}_{len(node_keys)}"
        node = rand_rep(node_keys, (B, N), C, g)
        edge = rand_rep([(0, 0), (1, 1)], (B, N, N), C, g)
        put_rep(store, tag + ".node", node)
        put_rep(store, tag + ".edge", edge)
        put_rep(store, tag + ".aggregate", cg_product(cgd, node, edge, maxdim=maxdim, aggregate=True))
        put_rep(store, tag + ".power", cg_product(cgd, node, node, maxdim=maxdim, aggregate=False))

    # geometry
    p = torch.randn(3, 6, 4, dtype=F64, generator=g)
    pc = torch.randn(2, 3, 6, 4, dtype=F64, generator=g)
    store["geo.p"] = npy(p); store["geo.pc"] = npy(pc)
    store["geo.p_to_rep"] = npy(ZF.p_to_rep(p)[(1, 1)])
    store["geo.p_cplx_to_rep"] = npy(ZF.p_cplx_to_rep(pc)[(1, 1)])
    store["geo.rep_to_p"] = npy(ZF.rep_to_p(pc))
    store["geo.normsq4"] = npy(ZF.normsq4(p))
    store["geo.repdot"] = npy(ZF.repdot({(1, 1): pc}, {(1, 1): pc})[(1, 1)])
    cgd2 = CGDict(maxdim=2, device=CPU, dtype=F64)
    zf, nrm, nsq = ZF.zonal_functions_rel(cgd2, p, p, 1, basis="cartesian")
    put_rep(store, "geo.zf_cart", zf); store["geo.zf_cart.norm"] = npy(nrm); store["geo.zf_cart.normsq"] = npy(nsq)
    pcc = ZF.p_cplx_to_rep(pc)[(1, 1)]
    zf, nrm, nsq = ZF.zonal_functions_rel(cgd2, pcc, pcc, 1, basis="canonical")
    put_rep(store, "geo.zf_canon", zf); store["geo.zf_canon.norm"] = npy(nrm); store["geo.zf_canon.normsq"] = npy(nsq)

    # radial filters, encoder (masked) and decoder (bias only), with grads
    for decoder in (False, True):
        torch.manual_seed(11 + int(decoder))
        rf = RadialFilters(max_zf=[1, 1], num_basis_fn=10, num_channels_out=[3, 4], num_levels=2,
                           input_basis="canonical" if decoder else "cartesian", device=CPU, dtype=F64)
        tag = "rad_dec" if decoder else "rad_enc"
        for k, v in rf.state_dict().items():
            store[f"{tag}.param.{k}"] = npy(v)
        if decoder:
            norms = torch.randn(2, 2, 5, 5, dtype=F64, generator=g)
            mask = torch.zeros(2, 2, 5, 5, dtype=F64)
        else:
            norms = torch.randn(2, 5, 5, dtype=F64, generator=g)
            norms[0, 1, 1] = 0.0
            mask = (torch.rand(2, 5, 5, generator=g) > 0.3).to(torch.uint8) * (norms != 0).byte()
        store[f"{tag}.norms"] = npy(norms); store[f"{tag}.mask"] = npy(mask)
        outs = rf(norms, mask)
        tot = 0
        for lvl, gs in enumerate(outs):
            for k, v in gs.items():
                store[f"{tag}.out.{lvl}.{k}"] = npy(v)
                w = torch.randn(v.shape, dtype=F64, generator=g)
                store[f"{tag}.cot.{lvl}.{k}"] = npy(w)
                tot = tot + (v * w).sum()
        tot.backward()
        for k, p_ in rf.named_parameters():
            store[f"{tag}.grad.{k}"] = npy(p_.grad if p_.grad is not None else torch.zeros_like(p_))

    # CGMLP
    torch.manual_seed(21)
    mlp = CGMLP({(0, 0): 3, (1, 1): 3}, activation="leakyrelu", num_hidden=6, layer_width_mul=6, device=CPU, dtype=F64)
    for k, v in mlp.state_dict().items():
        store[f"mlp.param.{k}"] = npy(v)
    node = rand_rep([(1, 1), (0, 0)], (2, 5), 3, g)
    put_rep(store, "mlp.in", node)
    put_rep(store, "mlp.out", mlp(GVec({k: v.clone() for k, v in node.items()})))

    # pooling min&max incl. the scalar max-by-square quirk
    lat = rand_rep([(0, 0), (1, 1)], (3, 7), 4, g)
    put_rep(store, "pool.in", lat)
    for m in ("min", "max", "min&max", "mean", "sum", "min+max"):
        put_rep(store, f"pool.{m}", aggregate(m, lat))

    # loss pieces
    x = torch.randn(3, 6, 4, dtype=F64, generator=g).requires_grad_(True)
    y = torch.randn(3, 6, 4, dtype=F64, generator=g)
    l = ChamferLoss(device=CPU)(x, y)
    l.backward()
    store["chamfer.x"] = npy(x); store["chamfer.y"] = npy(y); store["chamfer.loss"] = npy(l); store["chamfer.grad_x"] = npy(x.grad)
    raw = torch.randn(3, 6, 4, dtype=F64, generator=g)
    store["normp4.in"] = npy(raw); store["normp4.overall_max"] = npy(normalize_p4(raw, "overall_max")[0])
    np.savez_compressed(os.path.join(OUT, "g4_ops.npz"), **store)
    print("g4_ops.npz", len(store))


def activations():
    """CGMLP (lgn/models/lgn_levels.py:124-241) with each activation get_activation_fn knows besides the default."""
    store = {}
    g = torch.Generator().manual_seed(77)
    node = {(1, 1): torch.randn(2, 3, 7, 4, 4, dtype=F64, generator=g), (0, 0): torch.randn(2, 3, 7, 4, 1, dtype=F64, generator=g)}
    cot = torch.randn(2, 3, 7, 4, 1, dtype=F64, generator=g)
    put_rep(store, "in", node)
    store["cot"] = npy(cot)
    for act in ("relu", "elu", "sigmoid", "logsigmoid", "atan"):
        torch.manual_seed(31)
        mlp = CGMLP({(0, 0): 4, (1, 1): 4}, activation=act, num_hidden=6, layer_width_mul=6, device=CPU, dtype=F64)
        for k, v in mlp.state_dict().items():
            store[f"{act}.param.{k}"] = npy(v)
        x = {k: v.clone() for k, v in node.items()}
        x[(0, 0)].requires_grad_(True)
        s_in = x[(0, 0)]
        out = mlp(GVec(x))
        put_rep(store, f"{act}.out", out)
        (out[(0, 0)] * cot).sum().backward()
        store[f"{act}.grad_in"] = npy(s_in.grad)
        for k, p_ in mlp.named_parameters():
            store[f"{act}.grad.{k}"] = npy(p_.grad)
    np.savez_compressed(os.path.join(OUT, "g9_activations.npz"), **store)
    print("g9_activations.npz", len(store))


def tables():
    store = {}
    cgd = CGDict(maxdim=3, device=CPU, dtype=F64)
    for (r1, r2), entry in cgd.items():
        for r, mat in entry.items():
            store[f"cg.{r1}.{r2}.{r}"] = npy(mat)
    # Lorentz D matrices (used by the equivariance harness re-statement): 3 boosts + 3 rotations
    angles = [(0.0, 0.3, 0.0), (0.7, 1.1, -0.4), (3.0, 0.0, 0.0)]
    boosts = [(0.0, 0.5j, 0.0), (0.0, 2.5j, 0.0), (0.0, 9.0j, 0.0)]
    for i, ang in enumerate(angles + boosts):
        for k in range(3):
            for n in range(3):
                D = rot.LorentzD((k, n), *ang, device=CPU, dtype=F64, cg_dict=cgd)
                store[f"lorentzD.{i}.({k}, {n})"] = npy(D)
        store[f"lorentzD.{i}.angles"] = np.array([[complex(a).real, complex(a).imag] for a in ang])
    np.savez_compressed(os.path.join(OUT, "g5_tables.npz"), **store)
    print("g5_tables.npz", len(store))


def harness():
    """The reference's lgn_tests on fixed weights and jets.  unit='TeV': the 'GeV' branch divides data['p4'] in place, once per
    covariance_test call (lgn_tests.py:94-96), so its tables depend on how often the loader's tensors were seen."""
    from lgn.models.autotest.lgn_tests import lgn_tests
    store = {}
    args = types.SimpleNamespace(num_test_batch=-1)
    for tag, (N, maxdim, che, chd, seed) in {"g1": (30, 2, (3, 3, 4, 4), (4, 4, 3, 3), 0),
                                             "g2": (30, 3, (4, 4, 6, 6), (6, 6, 4, 4), 1)}.items():
        enc, dec = build(N, maxdim, che, chd, seed)                  # == the weights stored in the g1 / g2 fixtures
        p4, labels = jets(6, N, 21 + maxdim, pad_rows=((1, 12), (4, 23)))
        store[f"{tag}.p4"], store[f"{tag}.labels"] = npy(p4), npy(labels)
        torch.manual_seed(1234)                                      # randperm of the permutation test
        res = lgn_tests(args, enc, dec, [{"p4": p4.clone(), "labels": labels.clone()}], unit="TeV")
        store[f"{tag}.gammas"] = np.array(res["gammas"], dtype=np.float64)
        store[f"{tag}.thetas"] = np.array(res["thetas"], dtype=np.float64)
        irreps = [(0, 0), (1, 1)]                                    # what get_node_dev measures (autotest/utils.py:22-45)
        for kind in ("boost", "rot"):
            out = res[f"{kind}_dev_output"]
            store[f"{tag}.{kind}_dev_output"] = np.array([[d[w] for w in irreps] for d in out], dtype=np.float64)
            inner = res[f"{kind}_dev_internal"]
            store[f"{tag}.{kind}_dev_internal"] = np.array([[[d[w] for w in irreps] for d in layer] for layer in inner], dtype=np.float64)
        store[f"{tag}.perm_invariance"] = np.array([res["perm_invariance_dev_output"][w] for w in irreps], dtype=np.float64)
        store[f"{tag}.perm_equivariance"] = np.array([res["perm_equivariance_dev_output"][w] for w in irreps], dtype=np.float64)
        print(tag, "harness: max rot dev", store[f"{tag}.rot_dev_output"].max(), "max boost dev", store[f"{tag}.boost_dev_output"].max(),
              "internal shape", store[f"{tag}.rot_dev_internal"].shape)
    store["meta"] = np.array(json.dumps(dict(unit="TeV", irreps=[[0, 0], [1, 1]], perm_seed=1234,
                                             note="tables of the reference's lgn_tests; rows = 26 alphas, internal [alpha][layer][irrep]")))
    np.savez_compressed(os.path.join(OUT, "g8_harness.npz"), **store)
    print("g8_harness.npz", len(store))


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    want = lambda tag: not only or tag in only          # noqa: E731
    if want("g1"):
        e2e("g1_e2e_maxdim2.npz", 4, 30, 2, (3, 3, 4, 4), (4, 4, 3, 3), seed=0, pad_rows=((1, 17), (3, 25)))
    if want("g2"):
        e2e("g2_e2e_maxdim3.npz", 2, 30, 3, (4, 4, 6, 6), (6, 6, 4, 4), seed=1, pad_rows=((1, 21),))
    if want("g3"):
        e2e("g3_e2e_n150.npz", 1, 150, 2, (3, 3, 4, 4), (4, 4, 3, 3), seed=2, pad_rows=((0, 131),))
    if want("g4"):
        ops()
    if want("g5"):
        tables()
    if want("g6"):
        e2e("g6_e2e_mix.npz", 3, 12, 2, (2, 2, 3, 3), (3, 3, 2, 2), seed=3, pad_rows=((1, 8),), map_to_latent="mix")
    if want("g8"):
        harness()
    if want("g7"):
        e2e("g7_e2e_meanmax.npz", 3, 12, 2, (2, 2, 3, 3), (3, 3, 2, 2), seed=4, pad_rows=((2, 9),), map_to_latent="mean+max")
    if want("g9"):
        activations()
        e2e("g9_e2e_elu.npz", 3, 12, 2, (2, 2, 3, 3), (3, 3, 2, 2), seed=5, pad_rows=((1, 7),), activation="elu")
    if want("g11"):       # --mlp-depth other than the default 6 (lgn/models/lgn_levels.py:151-176): 5 Linear layers at maxdim 2 (H <= 48
        # kernels), 4 at maxdim 3 (H = 72: the wide kernels)
        e2e("g11_e2e_mlpdepth4.npz", 3, 12, 2, (2, 3, 4, 4), (4, 4, 3, 2), seed=7, pad_rows=((1, 8),), mlp_depth=4)
        e2e("g11_e2e_mlpdepth3_maxdim3.npz", 2, 10, 3, (2, 4, 6), (6, 4, 2), seed=8, pad_rows=((0, 7),), mlp_depth=3)
    if want("g13"):       # --num-basis-fn 5 (lgn/nn/position_levels.py:44-64): 10 Lorentzian bells instead of 20, maxdim 2 and maxdim 3
        e2e("g13_e2e_basis5.npz", 3, 12, 2, (2, 3, 4, 4), (4, 3, 3, 2), seed=10, pad_rows=((2, 9),), num_basis_fn=5)
        e2e("g13_e2e_basis5_maxdim3.npz", 2, 10, 3, (2, 3, 4), (4, 3, 2), seed=11, pad_rows=((1, 6),), num_basis_fn=5)
    if want("g15"):       # --num-basis-fn beyond the default 10: 24 bells (12: one whole group of the kernels' 20 + a padded one) at maxdim 2,
        # 40 bells (20: two whole groups) at maxdim 3
        e2e("g15_e2e_basis12.npz", 3, 12, 2, (2, 3, 4, 4), (4, 3, 3, 2), seed=16, pad_rows=((1, 9),), num_basis_fn=12)
        e2e("g15_e2e_basis20_maxdim3.npz", 2, 10, 3, (2, 3, 4), (4, 3, 2), seed=17, pad_rows=((0, 6),), num_basis_fn=20)
    if want("g14"):       # --mlp-width other than the default 6 (lgn/models/lgn_levels.py:124-189): hidden width = mlp_width * 2C -- 16 / 24 /
        # 32 (width 4), 20 / 30 / 40 (width 5: not multiples of the 4-deep matrix instruction), 28 / 42 / 56 (width 7: C = 4 leaves the
        # H <= 48 kernels), and 20 / 40 / 60 at maxdim 3
        e2e("g14_e2e_mlpwidth4.npz", 3, 12, 2, (2, 3, 4, 4), (4, 4, 3, 2), seed=12, pad_rows=((1, 8),), mlp_width=4)
        e2e("g14_e2e_mlpwidth5.npz", 3, 12, 2, (2, 3, 4, 4), (4, 4, 3, 2), seed=13, pad_rows=((0, 10),), mlp_width=5)
        e2e("g14_e2e_mlpwidth7.npz", 3, 12, 2, (2, 3, 4, 4), (4, 4, 3, 2), seed=14, pad_rows=((2, 7),), mlp_width=7)
        e2e("g14_e2e_mlpwidth5_maxdim3.npz", 2, 10, 3, (2, 4, 6), (6, 4, 2), seed=15, pad_rows=((1, 6),), mlp_width=5)
    if want("g12"):       # 150 particles at maxdim 3 (the product of two BASELINE axes; the jet's packed features exceed a CU's LDS)
        e2e("g12_e2e_n150_maxdim3.npz", 1, 150, 3, (4, 4, 6, 6), (6, 6, 4, 4), seed=9, pad_rows=((0, 137),))
    if want("g10"):
        e2e("g10_e2e_jetfeat.npz", 3, 12, 2, (2, 2, 3, 3), (3, 3, 2, 2), seed=6, pad_rows=((2, 9),), jet_features=True, extra_scalars=1)
