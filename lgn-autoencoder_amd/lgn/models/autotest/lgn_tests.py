"""
Re-statement of the reference's equivariance test (README.md:111-117): the autoencoder output and every
internal node feature must transform with the Lorentz-group representation matrices when the input jet is
rotated or boosted, and the output must not depend on the particle order.

What is compared (lgn/models/autotest/lgn_tests.py:179-269): for 26 rapidities alpha in [0, alpha_max]
(gamma = cosh alpha up to 11013) and 26 angles in [0, 2 pi] about `axis`,
    f(R x)   vs   D(R) f(x)        for the output GVec and all `nodes_all` GVecs,
with the deviation |mean(a - b) / mean(b)| per irrep (autotest/utils.py:22-45).  The reference has no
pass/fail threshold; `check_equivariance` adds explicit ones (DEFAULT_THRESHOLDS).

D matrices (lgn/g_lib/rotations.py:84-156): Wigner D^{k/2}(a,b,c) (x) conj(D^{n/2}(-a,b,-c)) taken to the
coupled basis with the CG matrix of (k,0)x(0,n)->(k,n); complex Euler angles give boosts.  Pinned against
the reference's matrices by tests/test_host.py (fixture g5_tables.npz).
"""
import logging
import time
from math import cosh, sqrt

import numpy as np
import torch
from scipy.linalg import expm

SEPARATOR = "=" * 50

# deviation limits on |mean(a-b)/mean(b)|; reference values on its own CPU path: rotation 4e-14..5e-13,
# boost 1e-13 at gamma=1 rising to 3e-6 at gamma=11013, permutation invariance 4e-13.
DEFAULT_THRESHOLDS = {"rotation": 1e-10, "boost_gamma_le_10": 1e-9, "boost_gamma_le_1000": 1e-6, "boost_any": 1e-3,
                      "perm_invariance": 1e-10}


# ---------------------------------------------------------------------------------------------------
# representation matrices
# ---------------------------------------------------------------------------------------------------
def _jy(j: float) -> np.ndarray:
    """Spin-j J_y in the basis m = j, j-1, ..., -j with the reference's phase (rotations.py:66-72)."""
    m = -np.arange(-j, j)
    ladder = np.sqrt((j + m) * (j - m + 1))
    jp, jm = np.diag(ladder, k=1), np.diag(ladder, k=-1)
    return -(jp - jm) / complex(0, 2)


def wigner_D(j: float, alpha, beta, gamma) -> np.ndarray:
    """exp(i alpha m) d^j(beta) exp(i gamma m'), m over -j..j (rotations.py:84-115); angles may be complex."""
    d = expm(1j * beta * _jy(j)) if j > 0 else np.ones((1, 1), dtype=complex)
    m = np.arange(-j, j + 1)
    return np.exp(1j * alpha * m)[:, None] * d * np.exp(1j * gamma * m)[None, :]


def lorentz_D(key, alpha, beta, gamma, cg_dict, dtype=torch.float64, device=None) -> torch.Tensor:
    """(2, d, d) planar representation matrix of irrep (k, n) (rotations.py:119-156)."""
    k, n = key
    d1 = wigner_D(k / 2, alpha, beta, gamma)
    d2 = np.conj(wigner_D(n / 2, -alpha, beta, -gamma))
    big = np.kron(d1, d2)
    cg = cg_dict[((k, 0), (0, n))][(k, n)].detach().cpu().numpy().astype(np.float64)
    D = cg @ big @ cg.T
    out = torch.stack([torch.from_numpy(np.ascontiguousarray(D.real)), torch.from_numpy(np.ascontiguousarray(D.imag))], 0)
    return out.to(dtype=dtype, device=device)


_H = 1.0 / sqrt(2.0)
_U = np.array([[1, 0, 0, 0], [0, _H, -1j * _H, 0], [0, 0, 0, 1], [0, -_H, -1j * _H, 0]], dtype=complex)   # Cartesian -> canonical


def cartesian_lorentz(D11: torch.Tensor) -> torch.Tensor:
    """Real 4x4 Lorentz matrix acting on Cartesian (E,px,py,pz) row vectors as p @ R (lgn_tests.py:23-79):
    R = Re(U^H D U)."""
    D = D11[0].detach().cpu().numpy() + 1j * D11[1].detach().cpu().numpy()
    R = (_U.conj().T @ (D @ _U)).real
    return torch.from_numpy(np.ascontiguousarray(R)).to(dtype=D11.dtype, device=D11.device)


def rotate_rep(rep, alpha, beta, gamma, cg_dict):
    """Apply D to every part of a GVec from the left in the reference's convention (rotations.py:7-51):
    (z_r D_r + z_i D_i, -z_r D_i + z_i D_r), i.e. z times conj(D)."""
    out = {}
    for key, z in rep.items():
        D = lorentz_D(key, alpha, beta, gamma, cg_dict, dtype=z.dtype, device=z.device)
        out[key] = torch.stack([z[0] @ D[0] + z[1] @ D[1], -(z[0] @ D[1]) + z[1] @ D[0]], 0)
    return rep.__class__(out)


# ---------------------------------------------------------------------------------------------------
# the tests
# ---------------------------------------------------------------------------------------------------
def get_output(encoder, decoder, data):
    latent, nodes = encoder(data, covariance_test=True)
    return decoder(latent, covariance_test=True, nodes_all=nodes)


REFERENCE_IRREPS = ((0, 0), (1, 1))        # what the reference's get_node_dev measures, whatever maxdim (autotest/utils.py:22-45)


def node_dev(a, b, eps=1e-16, mode="mean", irreps=REFERENCE_IRREPS):
    """irreps='all': every irrep both GVecs carry -- at maxdim 3 the internal features also have (2,0), (0,2), (2,2), which
    the reference rotates (rotate_rep) but never compares (used with mode='maxnorm' for the extended tables)."""
    ws = [w for w in a.keys() if w in b.keys()] if irreps == "all" else list(irreps)
    if mode == "max":
        return {w: ((a[w] - b[w]) / (b[w] + eps)).abs().max().item() for w in ws}
    if mode == "maxnorm":     # max |a - b| / max |b|: well defined for the traceless irreps, whose MEAN is ~0 (the reference's
        return {w: ((a[w] - b[w]).abs().max() / (b[w].abs().max() + eps)).item() for w in ws}     # metric is 0/0 noise there)
    return {w: abs((a[w] - b[w]).mean().item() / (b[w].mean().item() + eps)) for w in ws}


def _angles(kind, value, axis):
    v = value * 1j if kind == "boost" else value
    return {"x": (v, 0, 0), "y": (0, v, 0)}.get(axis.lower(), (0, 0, v))


@torch.no_grad()
def covariance_test(encoder, decoder, data, test_type, axis="z", alpha_max=None, cg_dict=None, unit="GeV", irreps=REFERENCE_IRREPS):
    cg_dict = encoder.cg_dict if cg_dict is None else cg_dict
    data = dict(data)
    data["p4"] = data["p4"].to(encoder.device, encoder.dtype)
    if unit.lower() == "gev":
        data["p4"] = data["p4"] / 1e3           # TeV: better conditioned after large boosts (lgn_tests.py:94-96)
    kind = "boost" if test_type.lower().startswith("boost") else "rot"
    if kind == "rot" and not test_type.lower().startswith("rot"):
        raise ValueError(f"test_type must be one of 'boost' or 'rotation': {test_type}")
    if alpha_max is None:
        alpha_max = 10.0 if kind == "boost" else 2 * np.pi
    grid = np.arange(0, alpha_max + 0.01, step=alpha_max / 25.0)
    ref_out, ref_nodes = get_output(encoder, decoder, data)
    dev_output, dev_internal, dev_all = [], [], []
    for value in grid:
        ang = _angles("boost" if kind == "boost" else "rot", value, axis)
        R = cartesian_lorentz(lorentz_D((1, 1), *ang, cg_dict, dtype=encoder.dtype, device=encoder.device))
        moved = dict(data)
        moved["p4"] = torch.einsum("...b,ba->...a", data["p4"], R)
        out_in, nodes_in = get_output(encoder, decoder, moved)                  # transform, then network
        out_rot = rotate_rep(ref_out, *ang, cg_dict)                            # network, then transform
        dev_output.append(node_dev(out_in, out_rot))
        moved_nodes = [rotate_rep(b, *ang, cg_dict) for b in ref_nodes]
        dev_internal.append([node_dev(a, b) for a, b in zip(nodes_in, moved_nodes)])
        if irreps == "all":
            dev_all.append([node_dev(a, b, mode="maxnorm", irreps="all") for a, b in zip(nodes_in, moved_nodes)])
    extra = {f"{kind}_dev_internal_all": dev_all} if irreps == "all" else {}
    if kind == "boost":
        return {"gammas": [cosh(x) for x in grid], "boost_dev_output": dev_output, "boost_dev_internal": dev_internal, **extra}
    return {"thetas": grid, "rot_dev_output": dev_output, "rot_dev_internal": dev_internal, **extra}


@torch.no_grad()
def permutation_invariance_test(encoder, decoder, data, *ignore, generator=None):
    """Permute the real (unmasked) particles (lgn_tests.py:140-176).  Returns (invariance, 'equivariance') max
    deviations; only the first is expected to vanish (the decoder orders its outputs by itself)."""
    mask = data["labels"] if "labels" in data else (data["p4"][..., 0] != 0).to(torch.uint8)
    B, N = mask.shape
    perm = torch.arange(N).expand(B, -1).clone()
    for b in range(B):
        n = int(mask[b].long().sum())
        perm[b, :n] = torch.randperm(n, generator=generator)

    def apply(t):
        return torch.stack([t[b, p.to(t.device)] for b, p in enumerate(perm)])

    assert (mask.cpu() == apply(mask.cpu())).all(), "the permutation must stay inside the real particles"
    permuted = {k: apply(v) if k in ("p4", "scalars") else v for k, v in data.items()}
    out_p, _ = get_output(encoder, decoder, permuted)
    out_n, _ = get_output(encoder, decoder, dict(data))
    out_p = {k: v.squeeze() for k, v in out_p.items()}
    out_n = {k: v.squeeze() for k, v in out_n.items()}
    moved = {k: torch.stack((apply(v[0]), apply(v[1])), 0) for k, v in out_n.items()}
    return node_dev(out_p, out_n, mode="max"), node_dev(out_p, moved, mode="max")


def _avg(dicts):
    return {k: sum(d[k] for d in dicts) / len(dicts) for k in dicts[0]}


@torch.no_grad()
def lgn_tests(args, encoder, decoder, dataloader, axis="z", alpha_max=None, theta_max=None, cg_dict=None, unit="GeV",
              irreps=REFERENCE_IRREPS):
    """Same call shape and result keys as the reference's lgn_tests (lgn_tests.py:292-423); prints plain tables.
    irreps='all' adds the tables ``boost_dev_internal_all`` / ``rot_dev_internal_all``: EVERY irrep of every internal GVec with
    the max-norm deviation (extension: see node_dev); the reference-shaped tables are unchanged."""
    t0 = time.time()
    logging.info("Covariance test begins...")
    encoder.eval(); decoder.eval()
    boosts, rots, pinv, pequi = [], [], [], []
    max_batches = getattr(args, "num_test_batch", -1) if args is not None else -1
    for idx, data in enumerate(dataloader):
        boosts.append(covariance_test(encoder, decoder, data, "boost", axis, alpha_max, cg_dict, unit, irreps))
        rots.append(covariance_test(encoder, decoder, data, "rotation", axis, theta_max, cg_dict, unit, irreps))
        a, b = permutation_invariance_test(encoder, decoder, data)
        pinv.append(a); pequi.append(b)
        if max_batches and max_batches > 0 and idx + 1 >= max_batches:
            break
    res = {"gammas": boosts[0]["gammas"], "thetas": rots[0]["thetas"]}
    for name, runs, key in (("boost", boosts, "boost"), ("rot", rots, "rot")):
        n_alpha = len(runs[0][f"{key}_dev_output"])
        res[f"{name}_dev_output"] = [_avg([r[f"{key}_dev_output"][i] for r in runs]) for i in range(n_alpha)]
        n_layers = len(runs[0][f"{key}_dev_internal"][0])
        res[f"{name}_dev_internal"] = [[_avg([r[f"{key}_dev_internal"][i][l] for r in runs]) for l in range(n_layers)]
                                       for i in range(n_alpha)]
        if f"{key}_dev_internal_all" in runs[0]:
            res[f"{name}_dev_internal_all"] = [[_avg([r[f"{key}_dev_internal_all"][i][l] for r in runs]) for l in range(n_layers)]
                                               for i in range(n_alpha)]
    res["perm_invariance_dev_output"] = _avg(pinv)
    res["perm_equivariance_dev_output"] = _avg(pequi)
    print(f"Covariance test completed! Time taken: {round((time.time() - t0) / 60, 2)} min")
    for title, xs, devs, xname in (("Boost", res["gammas"], res["boost_dev_output"], "gamma"),
                                   ("Rotation", res["thetas"], res["rot_dev_output"], "theta")):
        print(SEPARATOR)
        print(f"{title} equivariance test result (output relative error)")
        print(f"{xname:>12s} {'(0,0)':>12s} {'(1,1)':>12s}")
        for x, d in zip(xs, devs):
            print(f"{x:12.4g} {d[(0, 0)]:12.3e} {d[(1, 1)]:12.3e}")
    print(SEPARATOR)
    print(f"Permutation invariance test result: {res['perm_invariance_dev_output']}")
    print(f"Permutation equivariance test result: {res['perm_equivariance_dev_output']}")
    print(SEPARATOR)
    return res


def check_equivariance(results, thresholds=None):
    """Explicit pass/fail on top of lgn_tests' tables.  Returns the list of violations (empty = pass)."""
    th = dict(DEFAULT_THRESHOLDS)
    th.update(thresholds or {})
    bad = []
    for theta, d in zip(results["thetas"], results["rot_dev_output"]):
        for k, v in d.items():
            if not v <= th["rotation"]:
                bad.append(f"rotation theta={theta:.3f} {k}: {v:.2e} > {th['rotation']:.0e}")
    for gamma, d in zip(results["gammas"], results["boost_dev_output"]):
        lim = th["boost_gamma_le_10"] if gamma <= 10 else th["boost_gamma_le_1000"] if gamma <= 1000 else th["boost_any"]
        for k, v in d.items():
            if not v <= lim:
                bad.append(f"boost gamma={gamma:.1f} {k}: {v:.2e} > {lim:.0e}")
    for k, v in results["perm_invariance_dev_output"].items():
        if not v <= th["perm_invariance"]:
            bad.append(f"permutation invariance {k}: {v:.2e} > {th['perm_invariance']:.0e}")
    return bad
