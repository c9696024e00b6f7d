"""
lgn.models.autotest.utils -- the module path the reference's CLIs import ``plot_all_dev`` from (main.py:20,
test.py:16, covariance_test.py:2), plus the deviation helpers of lgn/models/autotest/utils.py:11-73.

``plot_all_dev(dev, save_path)`` persists the result dict of ``lgn_tests`` exactly where the reference does
(``<save_path>/pt_files/*.pt``, same file names and contents: autotest/utils.py:313-409) and writes the curves the
reference draws as plain-text tables next to them.  The matplotlib figures themselves are evaluation/plotting
code, which SURVEY section 2 (#11, #16) marks out of scope; the numbers behind every figure are kept.
"""
import os
import os.path as osp

import torch

from .lgn_tests import get_output as _get_output, node_dev


def get_output(encoder, decoder, data, covariance_test=True):
    """autotest/utils.py:11-19."""
    if covariance_test:
        return _get_output(encoder, decoder, data)
    return decoder(encoder(data)), None


def get_node_dev(transform_input, transform_output, eps=1e-16, mode="mean"):
    """autotest/utils.py:22-45: relative deviation of means (or max relative deviation) for (0,0) and (1,1)."""
    return node_dev(transform_input, transform_output, eps=eps, mode="max" if mode.lower() == "max" else "mean")


def get_dev(transform_input, transform_output, transform_input_nodes_all, transform_output_nodes_all, mode="mean"):
    """autotest/utils.py:48-72."""
    dev_output = [get_node_dev(a, b, mode=mode) for a, b in zip(transform_input, transform_output)]
    dev_internal = [[get_node_dev(a, b, mode=mode) for a, b in zip(la, lb)]
                    for la, lb in zip(transform_input_nodes_all, transform_output_nodes_all)]
    return dev_output, dev_internal


def get_internal_dev_stats(dev_internal):
    """Mean / max over layers for every alpha, and the per-layer curves (autotest/utils.py:166-210)."""
    keys = [(0, 0), (1, 1)]
    mean = {k: [sum(l[k] for l in row) / len(row) for row in dev_internal] for k in keys}
    mx = {k: [max(l[k] for l in row) for row in dev_internal] for k in keys}
    layers = {k: [[row[j][k] for row in dev_internal] for j in range(len(dev_internal[0]))] for k in keys}
    return mean, mx, layers


def make_dir(path):
    os.makedirs(path, exist_ok=True)
    return path


def _table(path, header, columns):
    with open(path, "w") as fh:
        fh.write(" ".join(f"{h:>16s}" for h in header) + "\n")
        for row in zip(*columns):
            fh.write(" ".join(f"{float(x):16.8e}" for x in row) + "\n")


def plot_all_dev(dev, save_path):
    """Store the equivariance-test results (autotest/utils.py:365-409; figures replaced by tables, see module docstring)."""
    make_dir(save_path)
    pt = make_dir(osp.join(save_path, "pt_files"))
    torch.save(dev["perm_invariance_dev_output"], osp.join(pt, "perm_invariance_dev_output.pt"))
    torch.save(dev["perm_equivariance_dev_output"], osp.join(pt, "perm_equivariance_dev_output.pt"))
    for kind, xs, xname in (("boost", dev["gammas"], "gamma"), ("rot", dev["thetas"], "theta")):
        out = dev[f"{kind}_dev_output"]
        for weight, name in (((1, 1), "p4"), ((0, 0), "scalars")):
            torch.save([d[weight] for d in out], osp.join(pt, f"{kind}_equivariance_{name}.pt"))
        long = "boost" if kind == "boost" else "rotation"
        _table(osp.join(save_path, f"{long}_equivariance_test_reconstructed.txt"), [xname, "dev(0,0)", "dev(1,1)"],
               [xs, [d[(0, 0)] for d in out], [d[(1, 1)] for d in out]])
        mean, mx, layers = get_internal_dev_stats(dev[f"{kind}_dev_internal"])
        for weight, name in (((1, 1), "4-vector"), ((0, 0), "scalar")):
            cols = [xs, mean[weight], mx[weight]] + layers[weight]
            head = [xname, "layers_mean", "layers_max"] + [f"layer_{i + 1}" for i in range(len(layers[weight]))]
            _table(osp.join(save_path, f"{long}_equivariance_test_internal_{name}.txt"), head, cols)


__all__ = ["get_output", "get_node_dev", "get_dev", "get_internal_dev_stats", "plot_all_dev", "make_dir"]
