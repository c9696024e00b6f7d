"""Helpers shared by the tests: golden-fixture loading, tolerance helpers."""
import ast
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rep_from(npz, prefix):
    """Rebuild an ordered {(k,n): tensor} from the '<prefix>.__keys__' + '<prefix>.(k, n)' entries."""
    keys = [tuple(k) for k in json.loads(str(npz[prefix + ".__keys__"]))]
    return {k: torch.from_numpy(npz[f"{prefix}.{k}"]) for k in keys}


def params_from(npz, prefix):
    out = {}
    for name in npz.files:
        if name.startswith(prefix + "."):
            out[name[len(prefix) + 1:]] = torch.from_numpy(npz[name])
    return out


def meta(npz):
    return json.loads(str(npz["meta"]))


def relerr(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


def assert_close(a, b, tol, what=""):
    a = torch.as_tensor(a); b = torch.as_tensor(b)
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = relerr(a.detach().cpu(), b.detach().cpu())
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


def assert_close_scaled(a, b, tol, scale, what=""):
    """|a - b|max <= tol * max(|b|max, scale): for gradients whose own magnitude is far below the network's gradient scale
    (rounding noise of the sums they come from is absolute, not relative to the survivor of a cancellation)."""
    a = torch.as_tensor(a, dtype=torch.float64).detach().cpu(); b = torch.as_tensor(b, dtype=torch.float64).detach().cpu()
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err, ref = (a - b).abs().max().item(), max(b.abs().max().item(), scale)
    assert err <= tol * ref, f"{what}: abs err {err:.3e} > {tol:.1e} * {ref:.3e}"
    return err / ref


def assert_rep_close(rep, ref, tol, what="", check_order=True):
    if check_order:
        assert list(rep.keys()) == list(ref.keys()), f"{what}: key order {list(rep.keys())} vs {list(ref.keys())}"
    else:
        assert set(rep.keys()) == set(ref.keys())
    for k in ref.keys():
        assert_close(rep[k], ref[k], tol, f"{what}{k}")


class OracleNet:
    """Gives the oracle (CPU restatement of the reference) the module call shape the equivariance harness expects."""

    def __init__(self, O, P, cfg, decoder, cg, maxdim=2):
        self.O, self.P, self.cfg, self.decoder = O, P, cfg, decoder
        self.maxdim, self.device, self.dtype, self.cg_dict = maxdim, torch.device("cpu"), torch.float64, cg

    def eval(self):
        return self

    def __call__(self, data, covariance_test=False, nodes_all=None):
        O = self.O
        if not self.decoder:
            return O.encoder_forward(self.P, self.cfg, data["p4"], data.get("labels"), covariance_test=True)
        gen, nodes = O.decoder_forward(self.P, self.cfg, data, covariance_test=True)
        return gen, list(nodes_all) + nodes
