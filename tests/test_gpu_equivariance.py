"""Equivariance acceptance test on the GPU (north star: "pass the repo's own equivariance test ... equivariance
error unchanged"), at maxdim 2 (cfg2's model) AND maxdim 3 (cfg5's model: irreps (2,0), (0,2), (2,2) inside).

The harness (lgn/models/autotest) runs on the native modules with the weights of the golden fixtures g1 / g2 and the jets
of g8_harness.npz, and its tables are compared with the tables the REFERENCE's own harness produced for the same weights
and jets on its CPU path (fixture g8, tests/golden/gen_golden.py g8; tests/test_harness_golden.py pins this repo's harness
to them on the oracle).  The native deviations must stay below explicit thresholds and within a small factor of the
reference's own deviations, point by point."""
import numpy as np
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu

CASES = {"g1": ("g1_e2e_maxdim2.npz", 2), "g2": ("g2_e2e_maxdim3.npz", 3)}


@pytest.mark.parametrize("tag", ["g1", "g2"])
def test_equivariance_harness_native_vs_reference_tables(tag):
    import __graft_entry__ as G
    from lgn.models.autotest import lgn_tests, check_equivariance
    dev = torch.device("cuda:0")
    name, maxdim = CASES[tag]
    z, h = U.load(name), U.load("g8_harness.npz")
    m = U.meta(z)
    enc, dec = G._models(m["N"], m["ch_enc"], m["ch_dec"], dev, seed=m["seed"], maxdim=maxdim)
    enc.load_state_dict(U.params_from(z, "enc")); dec.load_state_dict(U.params_from(z, "dec"))
    p4, labels = torch.from_numpy(h[f"{tag}.p4"]), torch.from_numpy(h[f"{tag}.labels"])
    res = lgn_tests(None, enc, dec, [{"p4": p4.clone(), "labels": labels.clone()}], unit="TeV", irreps="all")
    gam = np.asarray(res["gammas"])
    np.testing.assert_allclose(gam, h[f"{tag}.gammas"], rtol=1e-13)

    if maxdim == 2:      # absolute thresholds (DEFAULT_THRESHOLDS); at maxdim 3 the reference itself leaves them at gamma > 2000
        bad = check_equivariance(res)
        assert not bad, "native path violates the equivariance thresholds:\n" + "\n".join(bad)
    else:
        sub = dict(res)
        keep = [i for i, g in enumerate(gam) if g <= 1000.0]
        sub["gammas"] = [res["gammas"][i] for i in keep]
        sub["boost_dev_output"] = [res["boost_dev_output"][i] for i in keep]
        bad = check_equivariance(sub, {"rotation": 5e-9, "boost_gamma_le_10": 1e-8, "boost_gamma_le_1000": 1e-5})
        assert not bad, "native path (maxdim 3) violates the equivariance thresholds:\n" + "\n".join(bad)

    # "equivariance error unchanged": same order of magnitude as the reference's CPU path at every angle / boost.  The
    # deviation is rounding noise that grows like eps * gamma^2 and scatters by an order of magnitude from one gamma to the
    # next, so a point passes if it is within 20x of the reference's value at the same point OR under the smooth envelope.
    irreps = [(0, 0), (1, 1)]
    # (maxdim 2: the native metric scatters between 3e-14 and 3e-12 below gamma = 10 -- |mean(a - b) / mean(b)| of rounding noise --
    # and WHICH boost gets the large value moves with the summation order of the pair sweep: round 6's partner split at small
    # batches moved the maximum from gamma = 5.6 (1.9e-12) to gamma = 3.8 (3.1e-12); the floor was 1e-13 until then)
    floor = 2e-13 if maxdim == 2 else 1e-10      # (maxdim 3: the reference's own rotation table scatters between 1e-12 and 6e-10)
    for kind, xs in (("rot", np.ones(26)), ("boost", gam)):
        ref = h[f"{tag}.{kind}_dev_output"]
        for row, (a, gm) in enumerate(zip(res[f"{kind}_dev_output"], xs)):
            for col, irrep in enumerate(irreps):
                # (envelope: the reference's own table reaches 2.4e-14 gamma^2 at gamma = 7382 for maxdim 2)
                bound = max(20 * max(ref[row][col], floor), floor + (5e-14 if maxdim == 2 else 1e-12) * float(gm) ** 2)
                assert a[irrep] <= bound, f"{tag} {kind} output {irrep} gamma={float(gm):.4g}: native {a[irrep]:.2e} vs reference {ref[row][col]:.2e}"
        # internal features of every layer: the reference's two irreps against its table ...
        ref_i = h[f"{tag}.{kind}_dev_internal"]
        assert len(res[f"{kind}_dev_internal"][0]) == ref_i.shape[1]
        for row, (per_alpha, gm) in enumerate(zip(res[f"{kind}_dev_internal"], xs)):
            for layer, d in enumerate(per_alpha):
                for col, irrep in enumerate(irreps):
                    bound = max(50 * max(ref_i[row][layer][col], floor), 100 * (floor + 1e-12 * float(gm) ** 2))
                    assert d[irrep] <= bound, f"{tag} {kind} internal layer {layer} {irrep} gamma={float(gm):.4g}: {d[irrep]:.2e} vs {ref_i[row][layer][col]:.2e}"
        # ... and EVERY irrep the level carries (maxdim 3: also (2,0), (0,2), (2,2); the reference rotates them but never compares
        # them, and its mean-based metric is 0/0 for the traceless ones) with the max-norm deviation.  Limits = 20x what the CPU path
        # (oracle) shows on the same jets (2.5e-9 for rotations) -- the input scalars sqrt|p^2| of near-massless particles are
        # cancellation noise (layer 0, up to 6e-8 on the GPU box) -- and 2e-7 + 1e-10 gamma^2 for boosts up to gamma = 1000
        for per_alpha, gm in zip(res[f"{kind}_dev_internal_all"], xs):
            for layer, d in enumerate(per_alpha):
                for irrep, v in d.items():
                    if kind == "rot":
                        assert v <= 2e-7, f"{tag} rotation internal layer {layer} {irrep}: {v:.2e}"
                    elif gm <= 1000.0:
                        # (layer 0 = the network INPUT: its scalars sqrt|p^2| are computed from the boosted momenta of nearly
                        # massless particles before any kernel runs, noise ~ 5e-9 gamma^2 on either path)
                        lim = 1e-7 * max(1.0, float(gm) ** 2) if layer == 0 else 2e-7 + 1e-10 * float(gm) ** 2
                        assert v <= lim, f"{tag} boost internal layer {layer} {irrep} gamma={float(gm):.3g}: {v:.2e}"
    if maxdim == 3:
        seen = {irrep for per_alpha in res["rot_dev_internal_all"] for d in per_alpha for irrep in d}
        assert {(2, 0), (0, 2), (2, 2)} <= seen, f"the maxdim-3 internal features were not all checked: {sorted(seen)}"
    assert max(res["perm_invariance_dev_output"].values()) <= 1e-10
    print(tag, "max rot dev native / reference:", max(max(d[w] for w in irreps) for d in res["rot_dev_output"]), h[f"{tag}.rot_dev_output"].max())
    print(tag, "max boost dev (gamma <= 1000) native / reference:",
          max(max(d[w] for w in irreps) for d, g in zip(res["boost_dev_output"], gam) if g <= 1000), h[f"{tag}.boost_dev_output"][gam <= 1000].max())
