"""Drop-in boundary: every ``from lgn... import X`` the reference's CALLERS of the hot path write
(main.py, test.py, covariance_test.py, anomaly_detection.py, utils/**) must resolve inside this repo's ``lgn``
package, and the reference's own ``utils.initialize.initialize_autoencoder`` must construct this repo's classes.

The two tests that read the reference checkout run only where it exists (the build container) and are skipped elsewhere."""
import ast
import glob
import importlib
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lgn-autoencoder_amd")

needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def _caller_files():
    files = [os.path.join(REF, f) for f in ("main.py", "test.py", "covariance_test.py", "anomaly_detection.py")]
    files += glob.glob(os.path.join(REF, "utils", "**", "*.py"), recursive=True)
    return [f for f in files if os.path.isfile(f)]


def _lgn_imports():
    """{(module, name)} of every ``from lgn... import name`` / ``import lgn...`` in the caller files (parsed, not executed)."""
    found = set()
    for path in _caller_files():
        with open(path) as fh:
            tree = ast.parse(fh.read(), filename=path)
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.level == 0 and node.module.split(".")[0] == "lgn":
                for a in node.names:
                    found.add((node.module, a.name, os.path.relpath(path, REF)))
            elif isinstance(node, ast.Import):
                for a in node.names:
                    if a.name.split(".")[0] == "lgn":
                        found.add((a.name, None, os.path.relpath(path, REF)))
    return sorted(found, key=str)


@needs_ref
def test_every_lgn_import_of_the_reference_callers_resolves():
    imports = _lgn_imports()
    # the survey's list: utils/train.py:5-6, utils/initialize.py:3, distance_sq.py:3, main.py:19-20, test.py:15-16, covariance_test.py:1-2
    assert {(m, n) for m, n, _ in imports} >= {
        ("lgn.models.lgn_encoder", "LGNEncoder"), ("lgn.models.lgn_decoder", "LGNDecoder"), ("lgn.models", "LGNEncoder"),
        ("lgn.models", "LGNDecoder"), ("lgn.cg_lib.zonal_functions", "p_cplx_to_rep"), ("lgn.cg_lib.zonal_functions", "repdot"),
        ("lgn.models.autotest.lgn_tests", "lgn_tests"), ("lgn.models.autotest.utils", "plot_all_dev")}
    for module, name, where in imports:
        mod = importlib.import_module(module)
        assert os.path.abspath(mod.__file__).startswith(PKG), f"{module} resolved outside the repo: {mod.__file__}"
        if name is not None and name != "*":
            assert hasattr(mod, name), f"{where}: `from {module} import {name}` does not resolve in lgn-autoencoder_amd/lgn"


@needs_ref
def test_reference_initialize_autoencoder_builds_this_repos_modules():
    """The reference's utils/initialize.py (imported read-only, jetnet & co. absent -> empty stand-in module names, as in
    SURVEY Appendix C) with this package first on sys.path: ``initialize_autoencoder(args)`` returns this repo's
    LGNEncoder / LGNDecoder with the reference's parameter counts (34 146 / 29 342, SURVEY a-13 / a-14), and
    ``initialize_optimizers`` accepts them."""
    code = textwrap.dedent("""
        import sys, types, argparse, torch
        for m in ("jetnet",):
            sys.modules.setdefault(m, types.ModuleType(m))
        import lgn
        from utils.initialize import initialize_autoencoder, initialize_optimizers
        args = argparse.Namespace(num_jet_particles=30, tau_jet_scalars=1, tau_jet_vectors=1, map_to_latent="min&max",
                                  tau_latent_scalars=1, tau_latent_vectors=8, maxdim=[2], encoder_num_channels=[3, 3, 4, 4],
                                  decoder_num_channels=[4, 4, 3, 3], weight_init="randn", level_gain=[1.0], num_basis_fn=10,
                                  activation="leakyrelu", scale=1.0, jet_features=False, mlp=True, mlp_depth=6, mlp_width=6,
                                  device=torch.device("cpu"), dtype=torch.float64, optimizer="adam", lr=5e-4)
        enc, dec = initialize_autoencoder(args, print_models=False)
        oe, od = initialize_optimizers(args, enc, dec)
        from lgn.models.lgn_encoder import LGNEncoder
        from lgn.models.lgn_decoder import LGNDecoder
        assert type(enc) is LGNEncoder and type(dec) is LGNDecoder
        print(lgn.__file__, type(enc).__module__, type(dec).__module__, enc.num_learnable_parameters, dec.num_learnable_parameters,
              enc.cg_dict is dec.cg_dict)
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, REF]), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    path, m_enc, m_dec, n_enc, n_dec, shared = out.stdout.split()[-6:]
    assert path.startswith(PKG)
    assert m_enc == "lgn.models.encoder" and m_dec == "lgn.models.decoder"
    assert (int(n_enc), int(n_dec)) == (34146, 29342)
    assert shared == "True"


def test_zonal_function_helpers_match_reference_vectors():
    """p_cplx_to_rep / repdot / rep_to_p / p_to_rep / normsq4 of lgn.cg_lib.zonal_functions against the reference's
    vectors in g4_ops.npz (generated by tests/golden/gen_golden.py from the reference's functions)."""
    import torch
    import _util as U
    from lgn.cg_lib import zonal_functions as Z
    z = U.load("g4_ops.npz")
    p = torch.from_numpy(z["geo.p"])
    pc = torch.from_numpy(z["geo.pc"])
    U.assert_close(Z.p_to_rep(p)[(1, 1)], z["geo.p_to_rep"], 1e-15, "p_to_rep")
    U.assert_close(Z.p_cplx_to_rep(pc)[(1, 1)], z["geo.p_cplx_to_rep"], 1e-15, "p_cplx_to_rep")
    U.assert_close(Z.p_cplx_to_rep({(1, 1): pc})[(1, 1)], z["geo.p_cplx_to_rep"], 1e-15, "p_cplx_to_rep(dict)")
    U.assert_close(Z.rep_to_p(pc), z["geo.rep_to_p"], 1e-15, "rep_to_p")
    U.assert_close(Z.repdot({(1, 1): pc}, {(1, 1): pc})[(1, 1)], z["geo.repdot"], 1e-14, "repdot")
    U.assert_close(Z.normsq({(1, 1): pc})[(1, 1)], z["geo.repdot"], 1e-14, "normsq")
    U.assert_close(Z.normsq4(p), z["geo.normsq4"], 1e-15, "normsq4")
