"""GPU test of the training-step harness (lgn/step.py) on the native modules: one full step on the golden
configuration must reproduce the reference's total loss and gradients (Chamfer + 1e-8 L1)."""
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu


def test_train_step_matches_reference_golden():
    import __graft_entry__ as G
    from lgn.step import TrainStep
    dev = torch.device("cuda:0")
    z = U.load("g1_e2e_maxdim2.npz")
    m = U.meta(z)
    enc, dec = G._models(m["N"], m["ch_enc"], m["ch_dec"], dev, seed=m["seed"])
    step = TrainStep(enc, dec, lr=5e-4, l1_lambda=m["l1_lambda"], optimizer=False)
    batch = {"p4": torch.from_numpy(z["p4"]).to(dev), "labels": torch.from_numpy(z["labels"]).to(dev)}
    total, recon = step.forward_backward(batch)
    U.assert_close(total, z["loss_total"], 1e-11, "total loss")
    U.assert_close(recon, z["recon"], 1e-11, "recon")
    lam = m["l1_lambda"]
    for pre, mod in (("enc", enc), ("dec", dec)):
        sd = U.params_from(z, pre)
        for k, p in mod.named_parameters():
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"]) + lam * torch.sign(sd[k])
            U.assert_close(p.grad, ref, 1e-9, f"grad {pre}.{k}")
    # a second call starts from zeroed gradients (flat buffer), same result
    total2, _ = step.forward_backward(batch)
    assert float(total2) == float(total)
