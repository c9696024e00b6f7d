"""Worker of test_native_step_forked_reductions_are_bit_identical: one process per case, so that the library-owned events of the
forked branch and the graphs captured with them start from a fresh HIP runtime (in the long-lived pytest process a replay of a
forked graph once died inside the runtime after ~300 earlier tests had captured their own graphs).
    python _fork_worker.py MAXDIM USE_GRAPH        -> exit code 0 and "identical" on stdout."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lgn-autoencoder_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402
from lgn.step import NativeTrainStep  # noqa: E402
from oracle import lgn_oracle as O  # noqa: E402


def main():
    maxdim, use_graph = int(sys.argv[1]), sys.argv[2] == "1"
    che, chd = ((3, 3, 4, 4), (4, 4, 3, 3)) if maxdim == 2 else ((4, 4, 6, 6), (6, 6, 4, 4))
    dev = torch.device("cuda:0")
    p4, labels = O.synthetic_jets(6, 30, seed=8, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    outs = []
    for fork in ("0", "1"):
        os.environ["LGN_AMD_FORK"] = fork
        enc, dec = G._models(30, che, chd, dev, seed=3, maxdim=maxdim)
        st = NativeTrainStep(enc, dec, batch_size=6, lr=5e-4, l1_lambda=1e-8, use_graph=use_graph)
        assert (st._side is not None) == (fork == "1")
        losses = [float(st.step(batch)[0]) for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((losses, st.flat.flat.clone(), st.flat.grad_buf.clone()))
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    print("identical")


if __name__ == "__main__":
    main()
