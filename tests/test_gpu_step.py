"""GPU test of the training-step harness (lgn/step.py) on the native modules: one full step on the golden
configuration must reproduce the reference's total loss and gradients (Chamfer + 1e-8 L1)."""
import os

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
        for k, g in mod.named_grads():
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"]) + lam * torch.sign(sd[k])
            U.assert_close(g, ref, 1e-9, f"grad {pre}.{k}")
    # a second call starts from zeroed gradients (flat buffer), same result
    total2, _ = step.forward_backward(batch)
    assert float(total2) == float(total)


def _golden_setup(name="g1_e2e_maxdim2.npz"):
    import __graft_entry__ as G
    dev = torch.device("cuda:0")
    z = U.load(name)
    m = U.meta(z)
    enc, dec = G._models(m["N"], m["ch_enc"], m["ch_dec"], dev, seed=m["seed"], maxdim=m.get("maxdim", 2),
                         activation=m.get("activation", "leakyrelu"), map_to_latent=m.get("map_to_latent", "min&max"),
                         mlp_depth=m.get("mlp_depth", 6), num_basis_fn=m.get("num_basis_fn", 10), mlp_width=m.get("mlp_width", 6))
    batch = {"p4": torch.from_numpy(z["p4"]).to(dev), "labels": torch.from_numpy(z["labels"]).to(dev)}
    return z, m, enc, dec, batch


@pytest.mark.parametrize("name", ["g1_e2e_maxdim2.npz", "g3_e2e_n150.npz", "g2_e2e_maxdim3.npz", "g9_e2e_elu.npz", "g7_e2e_meanmax.npz", "g6_e2e_mix.npz",
                                  "g11_e2e_mlpdepth4.npz", "g11_e2e_mlpdepth3_maxdim3.npz", "g12_e2e_n150_maxdim3.npz", "g13_e2e_basis5.npz",
                                  "g13_e2e_basis5_maxdim3.npz", "g14_e2e_mlpwidth4.npz", "g14_e2e_mlpwidth5.npz", "g14_e2e_mlpwidth7.npz",
                                  "g14_e2e_mlpwidth5_maxdim3.npz"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_native_step_matches_reference_golden(name, use_graph):
    """lgn_step_fwd_bwd_f64 + lgn_step_finalize_f64 (one native call each, optionally replayed from a HIP graph)
    against the reference's loss / reconstruction / gradients."""
    from lgn.step import CapturedModuleStep, NativeTrainStep, native_train_step
    z, m, enc, dec, batch = _golden_setup(name)
    # (g12, 150 particles at maxdim 3, and g13, num_basis_fn = 5, included since round 6: the decoder's input stage of 150 x 6 fits a CU's
    # LDS now, the radial parameters are STORED 20 bells wide -- the chooser must take the whole-step call for every fixture)
    step = native_train_step(enc, dec, m["B"], l1_lambda=m["l1_lambda"], optimizer=False, use_graph=use_graph)
    assert isinstance(step, NativeTrainStep), "the chooser must take the whole-step call for this configuration"
    for _ in range(2):                      # second iteration = graph replay on the same buffers
        total, recon = step.step(batch)
    U.assert_close(total, z["loss_total"], 1e-11, "total loss")
    U.assert_close(step.loss_out[1], z["loss_chamfer"], 1e-11, "chamfer")
    U.assert_close(step.loss_out[2], z["l1_norm"], 1e-12, "l1")
    U.assert_close(recon, z["recon"], 1e-11, "recon")
    lam = m["l1_lambda"]
    for pre, mod in (("enc", enc), ("dec", dec)):
        sd = U.params_from(z, pre)
        for k, g in mod.named_grads():
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"]) + lam * torch.sign(sd[k])
            U.assert_close(g, ref, 1e-9, f"grad {pre}.{k}")
            if z[f"grad.{pre}.{k}"].max() == 0 and z[f"grad.{pre}.{k}"].min() == 0:
                assert torch.equal(g.cpu(), lam * torch.sign(sd[k])), f"{pre}.{k}: dead parameter must get exactly the L1 term"


@pytest.mark.parametrize("name", ["g15_e2e_basis12.npz", "g15_e2e_basis20_maxdim3.npz"])
def test_more_than_twenty_bells_train_through_the_captured_module_step(name):
    """num_basis_fn > 10 (lgn/nn/position_levels.py:60-97 takes any value): the whole-step call reads ONE group of 20 bells and must
    refuse; the chooser then captures the module-API step (per-operator calls, moments summed over the groups of bells) into one
    graph.  Loss, reconstruction and every gradient against the reference's vectors."""
    from lgn.step import CapturedModuleStep, NativeTrainStep, native_train_step
    z, m, enc, dec, batch = _golden_setup(name)
    with pytest.raises(NotImplementedError):
        NativeTrainStep(enc, dec, batch_size=m["B"], l1_lambda=m["l1_lambda"], optimizer=False)
    step = native_train_step(enc, dec, m["B"], l1_lambda=m["l1_lambda"], optimizer=False)
    assert isinstance(step, CapturedModuleStep)
    for _ in range(2):
        total, recon = step.step(batch)
    U.assert_close(total, z["loss_total"], 1e-11, "total loss")
    U.assert_close(recon, z["recon"], 1e-11, "recon")
    lam = m["l1_lambda"]
    for pre, mod in (("enc", enc), ("dec", dec)):
        sd = U.params_from(z, pre)
        for k, g in mod.named_grads():
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"]) + lam * torch.sign(sd[k])
            U.assert_close(g, ref, 1e-9, f"grad {pre}.{k}")


@pytest.mark.parametrize("flags", [("LGN_AMD_LEVEL_V2",), ("LGN_AMD_DEC_PAIRWISE",), ("LGN_AMD_BWD_ORDERED",), ("LGN_AMD_MLP_V1",), ("LGN_AMD_MLP_BWD1",),
                                   ("LGN_AMD_LEVEL_V2", "LGN_AMD_DEC_PAIRWISE")])
@pytest.mark.parametrize("use_graph", [False, True])
def test_native_step_maxdim2_alternative_kernels(flags, use_graph, monkeypatch):
    """The kernel-selecting switches of the maxdim = 2 step, frozen into the descriptor when the step is created (lgn/_native.py:
    net_flags): the three-kernel level backward, the decoder as pair sweeps, the ordered radial-gradient sweep, the 12-wave CGMLP
    kernels -- each against the reference's g1 gradients with the strict per-tensor tolerance.  (At g1's 4 jets the CGMLP runs its
    16-row workgroups either way; test_native_step_batch_regimes covers the chain kernels against the 12-wave ones at 512 jets.)"""
    from lgn.step import NativeTrainStep
    for f in flags:
        monkeypatch.setenv(f, "1")
    z, m, enc, dec, batch = _golden_setup("g1_e2e_maxdim2.npz")
    step = NativeTrainStep(enc, dec, batch_size=m["B"], l1_lambda=m["l1_lambda"], optimizer=False, use_graph=use_graph)
    for f in flags:
        monkeypatch.delenv(f)                # the switches live in the descriptor from here on
    for _ in range(2):
        total, recon = step.step(batch)
    U.assert_close(total, z["loss_total"], 1e-11, "total loss")
    U.assert_close(recon, z["recon"], 1e-11, "recon")
    lam = m["l1_lambda"]
    for pre, mod in (("enc", enc), ("dec", dec)):
        sd = U.params_from(z, pre)
        for k, g in mod.named_grads():
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"]) + lam * torch.sign(sd[k])
            U.assert_close(g, ref, 1e-9, f"grad {pre}.{k}")


@pytest.mark.parametrize("flag", ["LGN_AMD_NO_STATIC", "LGN_AMD_DEC_PAIRWISE", "LGN_AMD_MOMENTS_V1", "LGN_AMD_DEC_UNFUSED", "LGN_AMD_MOMENTS_SPLIT"])
def test_native_step_maxdim3_alternative_kernels(flag, monkeypatch):
    """The table-driven (maxdim 3) native step through its cross-check kernels: run-time-table local kernels instead of the
    compile-time-table ones (LGN_AMD_NO_STATIC), decoder moments as pair sweeps instead of the separable jet sums
    (LGN_AMD_DEC_PAIRWISE), component-chunked moments kernels (LGN_AMD_MOMENTS_V1), the decoder's moments as a tensor between two
    kernels per level instead of the fused separable form of round 6 (LGN_AMD_DEC_UNFUSED), the encoder's two backward pair sweeps as
    two kernels instead of the merged one (LGN_AMD_MOMENTS_SPLIT).  Same golden vectors, same tolerances;
    the switches are read per call, so the step object is built after setting them (the workspace is sized per form)."""
    from lgn.step import NativeTrainStep
    monkeypatch.setenv(flag, "1")
    if flag == "LGN_AMD_MOMENTS_V1":        # those kernels know the node-major layout only (the library says so otherwise)
        monkeypatch.setenv("LGN_AMD_NO_STATIC", "1")
    z, m, enc, dec, batch = _golden_setup("g2_e2e_maxdim3.npz")
    step = NativeTrainStep(enc, dec, batch_size=m["B"], l1_lambda=m["l1_lambda"], optimizer=False, use_graph=False)
    total, recon = step.step(batch)
    U.assert_close(total, z["loss_total"], 1e-11, "total loss")
    U.assert_close(recon, z["recon"], 1e-11, "recon")
    lam = m["l1_lambda"]
    for pre, mod in (("enc", enc), ("dec", dec)):
        sd = U.params_from(z, pre)
        for k, g in mod.named_grads():
            ref = torch.from_numpy(z[f"grad.{pre}.{k}"]) + lam * torch.sign(sd[k])
            U.assert_close(g, ref, 1e-9, f"grad {pre}.{k}")


@pytest.mark.parametrize("B,N,che,chd", [(5, 30, (4, 4, 6, 6), (6, 6, 4, 4)), (64, 30, (4, 4, 6, 6), (6, 6, 4, 4)), (33, 21, (3, 4, 4), (4, 4, 3)),
                                         (6, 32, (2, 3, 8), (8, 3, 2)), (9, 13, (3, 4, 4), (4, 4, 3)), (3, 7, (2, 5), (5, 2)), (1, 30, (3, 4), (4, 3))])
def test_native_step_maxdim3_fused_decoder_matches_the_two_kernel_form(B, N, che, chd, monkeypatch):
    """Decoder levels of the table-driven step with the separable moments kept on chip (csrc/generic_local_sep.hip: jet table instead
    of the moments tensor, one backward kernel per level on pairs of jets) against the round-5 sequence (LGN_AMD_DEC_UNFUSED=1:
    dec_sep_fwd/bwd_tb + local_fwd/bwd_static) on the same weights: odd jet counts (the last pair half empty), one jet, jets of
    7 .. 32 particles (idle lanes in every half wave), 2 .. 8 channels (every compile-time channel bound)."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    p4, labels = O.synthetic_jets(B, N, seed=B + N, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    runs = []
    for unfused in (False, True):
        enc, dec = G._models(N, che, chd, dev, seed=3, maxdim=3)
        if unfused:
            monkeypatch.setenv("LGN_AMD_DEC_UNFUSED", "1")
        st = NativeTrainStep(enc, dec, batch_size=B, optimizer=False, use_graph=True)
        if unfused:
            monkeypatch.delenv("LGN_AMD_DEC_UNFUSED")
        st.step(batch)
        loss, recon = st.step(batch)          # (the replay)
        runs.append((loss.clone(), recon.clone(), st.flat.grad.clone()))
    U.assert_close(runs[0][0], runs[1][0], 1e-13, "loss")
    U.assert_close(runs[0][1], runs[1][1], 1e-12, "recon")
    # gradients: every named tensor against its own scale (the two forms add the same terms in a different order)
    U.assert_close(runs[0][2], runs[1][2], 1e-11, "flat gradient")
    assert torch.isfinite(runs[0][2]).all()


def test_native_adam_matches_torch_adam_and_modular_path():
    """Three optimiser steps: native graph-replayed step vs the autograd/module path with torch.optim.Adam."""
    from lgn.step import NativeTrainStep, TrainStep
    z, m, enc, dec, batch = _golden_setup()
    _, _, enc2, dec2, _ = _golden_setup()
    a = NativeTrainStep(enc, dec, batch_size=m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True)
    b = TrainStep(enc2, dec2, lr=5e-4, l1_lambda=1e-8)
    for it in range(3):
        la, _ = a.step(batch)
        lb, _ = b.step(batch)
        U.assert_close(la, lb, 1e-10, f"loss at step {it}")
    U.assert_close(a.flat.flat, b.flat.flat, 1e-9, "parameters after 3 Adam steps")
    assert int(a.step_dev.item()) == 3


@pytest.mark.parametrize("name,B,N,maxdim,che,chd", [("cfg2", 512, 30, 2, (3, 3, 4, 4), (4, 4, 3, 3)),
                                                     ("cfg4", 256, 150, 2, (3, 3, 4, 4), (4, 4, 3, 3)),
                                                     ("cfg5", 512, 30, 3, (4, 4, 6, 6), (6, 6, 4, 4))])
def test_native_step_full_size_properties(name, B, N, maxdim, che, chd):
    """BASELINE sizes (cfg2, cfg4, cfg5): the graph-replayed native step agrees with the module / autograd path on the same
    weights (which the tests of test_gpu_parity.py tie to the oracle and to the reference's golden vectors)."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    enc, dec = G._models(N, che, chd, dev, seed=11, maxdim=maxdim)
    enc2, dec2 = G._models(N, che, chd, dev, seed=11, maxdim=maxdim)
    p4, labels = O.synthetic_jets(B, N, seed=4, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    a = NativeTrainStep(enc, dec, batch_size=B, optimizer=False, use_graph=True)
    b = TrainStep(enc2, dec2, optimizer=False)
    la, ra = a.step(batch)
    lb, rb = b.forward_backward(batch)
    U.assert_close(la, lb, 1e-12, "loss")
    U.assert_close(ra, rb, 1e-12, "recon")
    U.assert_close(a.flat.grad, b.flat.grad, 1e-9, "flat gradient")
    assert torch.isfinite(a.flat.grad).all()


@pytest.mark.parametrize("B", [64, 100, 128, 256, 300, 512])
def test_native_step_batch_regimes(B, monkeypatch):
    """cfg2 shapes at the batch sizes that change the launch geometry: the level kernels split a jet over 8 / 4 / 2 / 1 workgroups
    (level.hpp: level_jet_split: 64 / 65 .. 128 / 129 .. 256 / more jets), the CGMLP runs 16-row workgroups with kept activations
    (<= 8 128 rows: chain kernels with every layer split over three waves), or 64-row workgroups (>= 8 129 rows: 300 and 512 jets).  The graph-replayed
    native step against the module / autograd path on the same weights, and at the chain-kernel sizes also against the same step
    on the 12-wave CGMLP kernels (LGN_AMD_MLP_V1=1) and on the one-role chain backward (LGN_AMD_MLP_BWD1=1)."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    N, che, chd = 30, (3, 3, 4, 4), (4, 4, 3, 3)
    enc, dec = G._models(N, che, chd, dev, seed=5)
    enc2, dec2 = G._models(N, che, chd, dev, seed=5)
    p4, labels = O.synthetic_jets(B, N, seed=B, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    a = NativeTrainStep(enc, dec, batch_size=B, optimizer=False, use_graph=True)
    b = TrainStep(enc2, dec2, optimizer=False)
    la, ra = a.step(batch)
    la, ra = a.step(batch)                  # (the replay)
    lb, rb = b.forward_backward(batch)
    U.assert_close(la, lb, 1e-12, "loss")
    U.assert_close(ra, rb, 1e-12, "recon")
    U.assert_close(a.flat.grad, b.flat.grad, 1e-9, "flat gradient")
    # every size runs chain CGMLP kernels (16-row workgroups below 8 129 rows, since round 6): against the 12-wave kernels
    enc3, dec3 = G._models(N, che, chd, dev, seed=5)
    monkeypatch.setenv("LGN_AMD_MLP_V1", "1")
    c = NativeTrainStep(enc3, dec3, batch_size=B, optimizer=False, use_graph=True)
    monkeypatch.delenv("LGN_AMD_MLP_V1")
    lc, rc = c.step(batch)
    U.assert_close(la, lc, 1e-13, "loss, chain vs 12-wave CGMLP kernels")
    U.assert_close(a.flat.grad, c.flat.grad, 1e-10, "flat gradient, chain vs 12-wave CGMLP kernels")
    # LGN_AMD_MLP_BWD1: below 8 129 rows the 16-row kernels with ONE chain wave (the default splits every layer over three),
    # from there on the 64-row kernels with one role per wave
    if B * N >= 8129:
        # the chain backward as one role per wave (round 5's kernel; the default splits chain and weight gradients over two sets of
        # waves): the same products; one 16 x 16 tile of every hidden layer's weight gradient is summed in four quarters
        enc4, dec4 = G._models(N, che, chd, dev, seed=5)
        monkeypatch.setenv("LGN_AMD_MLP_BWD1", "1")
        d = NativeTrainStep(enc4, dec4, batch_size=B, optimizer=False, use_graph=True)
        monkeypatch.delenv("LGN_AMD_MLP_BWD1")
        ld, _ = d.step(batch)
        assert torch.equal(la, ld), "two-role vs one-role chain backward: loss"
        U.assert_close(a.flat.grad, d.flat.grad, 1e-13, "flat gradient, two-role vs one-role chain backward")
    else:
        enc4, dec4 = G._models(N, che, chd, dev, seed=5)
        monkeypatch.setenv("LGN_AMD_MLP_BWD1", "1")
        d = NativeTrainStep(enc4, dec4, batch_size=B, optimizer=False, use_graph=True)
        monkeypatch.delenv("LGN_AMD_MLP_BWD1")
        ld, _ = d.step(batch)
        U.assert_close(la, ld, 1e-13, "loss, three chain waves vs one per 16 rows")
        U.assert_close(a.flat.grad, d.flat.grad, 1e-11, "flat gradient, three chain waves vs one per 16 rows")


@pytest.mark.parametrize("width,maxdim,B", [(4, 2, 280), (5, 2, 280), (7, 2, 280), (5, 2, 64), (4, 3, 280), (5, 3, 140)])
def test_native_step_mlp_widths(width, maxdim, B):
    """mlp_width other than 6 through the whole-step call at batch sizes that take the 64-row CGMLP workgroups (>= 8 192 rows; the
    g14 fixtures have 36 rows): the graph-replayed native step against the module / autograd path on the same weights, whose
    operators are held to the oracle one by one (test_cgmlp_widths, the level tests)."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    N = 30
    che, chd = ((3, 3, 4, 4), (4, 4, 3, 3)) if maxdim == 2 else ((3, 4, 4), (4, 4, 3))
    enc, dec = G._models(N, che, chd, dev, seed=5, maxdim=maxdim, mlp_width=width)
    enc2, dec2 = G._models(N, che, chd, dev, seed=5, maxdim=maxdim, mlp_width=width)
    for m in (enc2, dec2):
        m.use_fused = False                   # one native call per operator
    p4, labels = O.synthetic_jets(B, N, seed=B + width, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    a = NativeTrainStep(enc, dec, batch_size=B, optimizer=False, use_graph=True)
    b = TrainStep(enc2, dec2, optimizer=False)
    la, ra = a.step(batch)
    la, ra = a.step(batch)                  # (the replay)
    lb, rb = b.forward_backward(batch)
    U.assert_close(la, lb, 1e-12, "loss")
    U.assert_close(ra, rb, 1e-12, "recon")
    U.assert_close(a.flat.grad, b.flat.grad, 1e-9, "flat gradient")
    assert torch.isfinite(a.flat.grad).all()


@pytest.mark.parametrize("B,N,use_graph", [(512, 30, True), (64, 30, True), (7, 30, False), (5, 70, True), (3, 150, False)])
def test_native_step_fused_tail_is_bit_identical_to_the_three_launches(B, N, use_graph, monkeypatch):
    """lgn_step_train_f64 (csrc/step_tail.hip: deferred reductions + radial finalisation + L1 + Adam + loss assembly in ONE launch)
    against the same step with LGN_AMD_SPLIT_TAIL=1 (reduce_segments, rad_finalize_batch, l1_adam as three launches): four Adam
    steps from the same weights -- gradients, moments, weights and the device-side step counter agree BIT FOR BIT (the loss VALUE
    to rounding: its scalar sums run over different partitions), at the launch geometries of 512 / 64 jets, at a handful of jets, and at the jet sizes that take the other level-backward kernels."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    che, chd = (3, 3, 4, 4), (4, 4, 3, 3)
    p4, labels = O.synthetic_jets(B, N, seed=B + N, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    runs = []
    for split in (False, True):
        enc, dec = G._models(N, che, chd, dev, seed=21)
        if split:
            monkeypatch.setenv("LGN_AMD_SPLIT_TAIL", "1")
        st = NativeTrainStep(enc, dec, batch_size=B, lr=1e-3, l1_lambda=1e-6, use_graph=use_graph)
        losses = [st.step(batch)[0].clone() for _ in range(4)]       # (the graph is captured while the switch is set)
        if split:
            monkeypatch.delenv("LGN_AMD_SPLIT_TAIL")
        torch.cuda.synchronize()
        runs.append((torch.stack(losses), st.loss_out.clone(), st.flat.grad.clone(), st.adam_m.clone(), st.adam_v.clone(),
                     st.flat.flat.clone(), st.step_dev.clone(), st._loss_buf[-12:].clone(), st._loss_buf[3:-12].clone()))
    names = ("losses of the four steps", "loss terms", "gradients (with the L1 term)", "Adam m", "Adam v", "weights", "step counter",
             "counters and cached powers at the end of the scratch block")
    for what, x, y in zip(names, *runs):
        if what.startswith("loss"):     # the scalar sums (chamfer terms, |w|) are added up over different partitions: equal to rounding
            U.assert_close(x, y, 1e-13, what)
            continue
        assert torch.equal(x, y), f"{what}: fused tail != three launches (max diff {(x.double() - y.double()).abs().max().item():.3e})"
    assert int(runs[0][6].item()) == 4 and torch.isfinite(runs[0][5]).all()
    assert float(runs[0][7][:4].abs().sum()) == 0.0 and float(runs[0][7][-1]) == 0.0, "the launch left its counters non-zero"
    assert float(runs[0][8].abs().sum()) == 0.0, "the fused launch left |w| slots behind (they mark 'not yet written' for the next one)"


@pytest.mark.parametrize("which", ["native", "captured"])
def test_native_step_two_ranks_match_single_process(tmp_path, which):
    dev = torch.device("cuda:0")
    """2 ranks x 8 jets (one all-reduce of gradients | loss terms between the two captured graphs) must reproduce the
    single-process 16-jet step: the loss is a SUM over jets, gradients are summed, the L1 term is added once."""
    import socket
    import subprocess
    import sys as _sys
    import bench
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep
    per_rank, world, steps = 8, 2, 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_native_worker.py")
    procs = [subprocess.Popen([_sys.executable, worker, str(r), str(world), str(port), str(tmp_path), str(per_rank), str(steps), which],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
    enc, dec = G._models(bench.N_PART, bench.CH_ENC, bench.CH_DEC, dev, seed=0)
    ref = NativeTrainStep(enc, dec, batch_size=per_rank * world, lr=5e-4, l1_lambda=1e-8, use_graph=True)
    p4, labels = bench.synthetic_jets(per_rank * world, bench.N_PART, seed=5)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    ref_losses = [float(ref.step(batch)[0]) for _ in range(steps)]
    for r in range(world):
        z = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"))
        U.assert_close(z["params"].to(dev), ref.flat.flat.detach(), 1e-9, f"rank {r} parameters after {steps} steps")
        for a, b in zip(z["losses"], ref_losses):
            assert abs(a - b) <= 1e-10 * max(1.0, abs(b)), (z["losses"], ref_losses)


def test_native_step_scaled_input_uses_unscaled_target():
    """--scale != 1 (lgn_encoder.py:376): only the encoder input is scaled, the reconstruction is compared with the
    UNscaled batch (utils/train.py:285-292).  Native step vs the module/autograd harness on the same weights."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    z, m, _, _, batch = _golden_setup()
    dev = torch.device("cuda:0")
    nets = []
    for _ in range(2):
        enc, dec = G._models(m["N"], m["ch_enc"], m["ch_dec"], dev, seed=m["seed"])
        enc.scale = 0.5
        nets.append((enc, dec))
    a = NativeTrainStep(*nets[0], batch_size=m["B"], optimizer=False, use_graph=True)
    b = TrainStep(*nets[1], optimizer=False)
    for _ in range(2):
        la, ra = a.step(batch)
    lb, rb = b.forward_backward(batch)
    U.assert_close(la, lb, 1e-11, "loss (scale = 0.5)")
    U.assert_close(ra, rb, 1e-11, "recon (scale = 0.5)")
    U.assert_close(a.flat.grad, b.flat.grad, 1e-9, "gradients (scale = 0.5)")
    assert abs(float(la) - float(z["loss_total"])) > 1e-9 * abs(float(z["loss_total"])), "the scale must reach the encoder input"
    with pytest.raises(ValueError, match="static buffers"):
        a.load_batch({"p4": batch["p4"][:1]})


def test_native_step_rccl_collective_branch_single_rank():
    """RCCL on the one GPU of the box: an `nccl` process group of world size 1 with ``force_collective=True`` runs the
    branches data-parallel runs take -- (a) the all-reduce(grad_buf) captured INSIDE the step graph (one launch per step),
    (b) graph | all-reduce | graph -- and both must equal the plain one-graph step bit for bit (a 1-rank SUM is the
    identity), over several optimiser steps."""
    import socket
    import torch.distributed as dist
    from lgn.step import NativeTrainStep
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        z, m, enc, dec, batch = _golden_setup()
        _, _, enc2, dec2, _ = _golden_setup()
        _, _, enc3, dec3, _ = _golden_setup()
        a = NativeTrainStep(enc, dec, batch_size=m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True, force_collective=True)
        b = NativeTrainStep(enc2, dec2, batch_size=m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True)
        c = NativeTrainStep(enc3, dec3, batch_size=m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True, force_collective=True,
                            graph_collective=False)
        assert a.collective and c.collective and not b.collective
        for it in range(3):
            la, _ = a.step(batch)
            lb, _ = b.step(batch)
            lc, _ = c.step(batch)
            # (a and c assemble the loss in l1_adam, b in the fused tail of lgn_step_train_f64: the same terms over another partition)
            assert float(la) == float(lc), f"step {it}: {float(la)!r} vs {float(lc)!r}"
            assert abs(float(la) - float(lb)) <= 1e-13 * abs(float(lb)), f"step {it}: {float(la)!r} vs {float(lb)!r}"
        assert b._g2 is None and b.launches_per_step == 1
        assert c._g2 is not None and c.launches_per_step == 3
        print(f"all-reduce inside the step graph: {a._in_graph} (launches per step: {a.launches_per_step})")
        assert a.launches_per_step in (1, 3)
        for t in (a, c):
            assert torch.equal(t.flat.flat, b.flat.flat) and torch.equal(t.flat.grad_buf, b.flat.grad_buf)
    finally:
        dist.destroy_process_group()


def test_module_api_training_loop_matches_native_step():
    """The reference's loop shape (utils/train.py:285-343): enc(batch) -> dec(latent) -> Chamfer + l1 -> backward -> two
    torch Adams, on the fused module path (ONE flat parameter per network), against the native graph-replayed step."""
    from lgn.step import NativeTrainStep, chamfer_loss, get_real
    z, m, enc, dec, batch = _golden_setup()
    _, _, enc2, dec2, _ = _golden_setup()
    ref = NativeTrainStep(enc2, dec2, batch_size=m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True)
    assert enc._fused_ok() and dec._fused_ok()
    oe, od = torch.optim.Adam(enc.parameters(), 5e-4), torch.optim.Adam(dec.parameters(), 5e-4)
    target = batch["p4"]
    for it in range(3):
        rec = dec(enc(batch))
        loss = chamfer_loss(get_real(rec, "sum"), target) + 1e-8 * (enc.l1_norm() + dec.l1_norm())
        oe.zero_grad(); od.zero_grad()
        loss.backward()
        oe.step(); od.step()
        lr, _ = ref.step(batch)
        U.assert_close(loss, lr, 1e-10, f"loss at step {it}")
    U.assert_close(torch.cat([enc.flat_params.detach(), dec.flat_params.detach()]), ref.flat.flat, 1e-9, "parameters after 3 steps")


@pytest.mark.parametrize("native_loss", [True, False])
def test_reference_loop_step_matches_native_step(native_loss):
    """lgn.step.ReferenceLoopStep (the loop body of utils/train.py:283-343 on the module API, with lgn.losses.ChamferLoss or the
    torch restatement of the reference's loss) against the native graph-replayed step: same losses, same parameters after 3 steps."""
    from lgn.step import NativeTrainStep, ReferenceLoopStep
    z, m, enc, dec, batch = _golden_setup()
    _, _, enc2, dec2, _ = _golden_setup()
    ref = NativeTrainStep(enc2, dec2, batch_size=m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True)
    loop = ReferenceLoopStep(enc, dec, lr=5e-4, l1_lambda=1e-8, native_loss=native_loss)
    for it in range(3):
        loss, _ = loop.step(batch)
        lr, _ = ref.step(batch)
        U.assert_close(loss, lr, 1e-10, f"loss at step {it}")
    U.assert_close(torch.cat([enc.flat_params.detach(), dec.flat_params.detach()]), ref.flat.flat, 1e-9, "parameters after 3 steps")


@pytest.mark.parametrize("route", ["native", "captured"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_jet_features_step_matches_reference_golden(use_graph, route):
    """g10 -- jet_features (one more encoder node than the decoder reconstructs) + an extra input scalar per node: the reference's
    loss, reconstruction and gradients through the whole-step native call (round 6: lgn_net_desc.dec_N / in_scalars; what the chooser
    takes) and through CapturedModuleStep (enc(batch) -> dec -> ChamferLoss -> backward() -> native L1 + Adam in one graph: the route
    of round 5, still what --chamfer-jet-features and maxdim 3 with jet features get), eager and replayed."""
    import __graft_entry__ as G
    from lgn.step import CapturedModuleStep, NativeTrainStep, native_train_step
    dev = torch.device("cuda:0")
    z = U.load("g10_e2e_jetfeat.npz")
    m = U.meta(z)
    enc, dec = G._models(m["N"], m["ch_enc"], m["ch_dec"], dev, seed=m["seed"], maxdim=m.get("maxdim", 2), jet_features=True,
                         tau_input_scalars=1 + m.get("extra_scalars", 0))
    enc.load_state_dict(U.params_from(z, "enc")); dec.load_state_dict(U.params_from(z, "dec"))
    batch = {"p4": torch.from_numpy(z["p4"]).to(dev), "labels": torch.from_numpy(z["labels"]).to(dev)}
    if "scalars" in z.files:
        batch["scalars"] = torch.from_numpy(z["scalars"]).to(dev)
    if route == "native":
        step = native_train_step(enc, dec, m["B"], l1_lambda=m["l1_lambda"], optimizer=False, use_graph=use_graph,
                                 extra_scalars=m.get("extra_scalars", 0))
        assert isinstance(step, NativeTrainStep) and step.split
    else:
        step = CapturedModuleStep(enc, dec, m["B"], l1_lambda=m["l1_lambda"], optimizer=False, use_graph=use_graph,
                                  extra_scalars=m.get("extra_scalars", 0))
    for _ in range(3):
        total, recon = step.step(batch)
    U.assert_close(total, z["loss_total"], 1e-11, "total loss")
    U.assert_close(step.loss_out[1], z["loss_chamfer"], 1e-11, "chamfer")
    U.assert_close(recon, z["recon"], 1e-11, "recon")
    lam = m["l1_lambda"]
    for pre, mod in (("enc", enc), ("dec", dec)):
        sd = U.params_from(z, pre)
        for k, g in mod.named_grads():
            U.assert_close(g, torch.from_numpy(z[f"grad.{pre}.{k}"]) + lam * torch.sign(sd[k]), 1e-9, f"grad {pre}.{k}")
    assert step.launches_per_step == (1 if use_graph else None)


@pytest.mark.parametrize("B,N,K,jet", [(64, 30, 2, True), (300, 30, 1, True), (7, 30, 3, False), (5, 70, 2, True)])
def test_native_step_with_input_scalars_matches_module_path(B, N, K, jet):
    """The split form of the whole-step call (jet_features and / or K - 1 extra input scalars) at the batch sizes that change the
    launch geometry and at a jet size that takes the large-jet level kernels, against the module / autograd path on the same weights
    (per-network native calls; the operators themselves are held to the oracle elsewhere)."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    che, chd = (3, 3, 4, 4), (4, 4, 3, 3)
    # tau_input_scalars = mass + (K - 1) extra scalars; jet_features adds the jet-mass term (and the jet node) by itself
    enc, dec = G._models(N, che, chd, dev, seed=5, jet_features=jet, tau_input_scalars=K)
    enc2, dec2 = G._models(N, che, chd, dev, seed=5, jet_features=jet, tau_input_scalars=K)
    p4, labels = O.synthetic_jets(B, N, seed=B, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    if K > 1:
        g = torch.Generator().manual_seed(B + K)
        batch["scalars"] = torch.randn(B, N + int(jet), K - 1, dtype=torch.float64, generator=g).to(dev)
    a = NativeTrainStep(enc, dec, batch_size=B, optimizer=False, use_graph=True)
    assert a.split
    b = TrainStep(enc2, dec2, optimizer=False)
    la, ra = a.step(batch)
    la, ra = a.step(batch)                  # (the replay)
    lb, rb = b.forward_backward(batch)
    U.assert_close(la, lb, 1e-12, "loss")
    U.assert_close(ra, rb, 1e-12, "recon")
    U.assert_close(a.flat.grad, b.flat.grad, 1e-9, "flat gradient")
    assert torch.isfinite(a.flat.grad).all()


@pytest.mark.parametrize("case", ["sum", "mixed_maxdim"])
def test_native_train_step_chooser_covers_module_only_configurations(case):
    """native_train_step falls back to CapturedModuleStep for what the whole-step call refuses -- the 'sum' latent map (an extra axis
    in the reference), an encoder with maxdim 3 feeding a decoder with maxdim 2 -- and the replayed graph reproduces the eager
    module-API step (TrainStep: per-network calls under autograd) on the same weights."""
    import __graft_entry__ as G
    from lgn.step import CapturedModuleStep, TrainStep, native_train_step
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    N, B, chans = 12, 4, ((2, 3, 3, 4), (4, 3, 3, 2))

    def build():
        if case == "sum":
            return G._models(N, chans[0], chans[1], dev, seed=7, map_to_latent="sum")
        enc, _ = G._models(N, chans[0], chans[1], dev, seed=7, maxdim=3)
        _, dec = G._models(N, chans[0], chans[1], dev, seed=7, maxdim=2)
        return enc, dec

    p4, labels = O.synthetic_jets(B, N, seed=11, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    a = native_train_step(*build(), B, optimizer=False, use_graph=True)
    assert isinstance(a, CapturedModuleStep)
    b = TrainStep(*build(), optimizer=False)
    for _ in range(2):
        la, ra = a.step(batch)
    lb, rb = b.forward_backward(batch)
    U.assert_close(la, lb, 1e-11, "loss")
    U.assert_close(ra, rb, 1e-11, "recon")
    U.assert_close(a.flat.grad, b.flat.grad, 1e-9, "flat gradient")


@pytest.mark.parametrize("which", ["captured", "native_maxdim3"])
def test_replayed_graphs_clear_gradients_to_exact_zero(which):
    """The gradient buffer and the zero blocks are cleared by a kernel of the library, not by hipMemsetAsync: on this stack the memset
    node of a REPLAYED graph fills its range with a stale pattern (denormals ~7e-310, tools/graph_memset_check.py) -- numerically
    invisible, but dead parameters (the decoder's radial bells and weights, the last levels' CGMLPs) must keep an EXACT zero
    gradient on every replay, with the L1 term switched off."""
    from lgn.step import CapturedModuleStep, NativeTrainStep
    if which == "captured":
        z, m, enc, dec, batch = _golden_setup()
        st = CapturedModuleStep(enc, dec, batch_size=m["B"], lr=5e-4, l1_lambda=0.0, optimizer=False, use_graph=True)
    else:
        z, m, enc, dec, batch = _golden_setup("g2_e2e_maxdim3.npz")
        st = NativeTrainStep(enc, dec, batch_size=m["B"], lr=5e-4, l1_lambda=0.0, optimizer=False, use_graph=True)
    for _ in range(4):
        st.step(batch)
    torch.cuda.synchronize()
    g = st.flat.grad
    tiny = (g != 0) & (g.abs() < 1e-200)
    assert int(tiny.sum()) == 0, f"{int(tiny.sum())} gradient entries hold denormal garbage, e.g. {g[tiny][:3].tolist()}"
    assert int((g == 0).sum()) > 100, "expected dead parameters with an exact zero gradient"


@pytest.mark.parametrize("jet_loss", [False, True])
def test_captured_module_step_trains_like_the_native_step(jet_loss):
    """Three Adam steps of CapturedModuleStep (graph replay) against NativeTrainStep on a configuration both cover; with
    --chamfer-jet-features (utils/train.py:432: the MSE of the summed momenta inside ChamferLoss) against the eager module loop."""
    from lgn.step import CapturedModuleStep, NativeTrainStep, ReferenceLoopStep, native_train_step
    z, m, enc, dec, batch = _golden_setup()
    _, _, enc2, dec2, _ = _golden_setup()
    a = CapturedModuleStep(enc, dec, batch_size=m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True, chamfer_jet_features=jet_loss)
    if jet_loss:
        b = ReferenceLoopStep(enc2, dec2, lr=5e-4, l1_lambda=1e-8)
        loss_fn = b.loss_fn
        b.loss_fn = lambda x, y: loss_fn(x, y, jet_features=True)
        flat_b = lambda: torch.cat([enc2.flat_params.detach(), dec2.flat_params.detach()])      # noqa: E731
    else:
        b = native_train_step(enc2, dec2, m["B"], lr=5e-4, l1_lambda=1e-8, use_graph=True)
        assert isinstance(b, NativeTrainStep)
        flat_b = lambda: b.flat.flat                                                               # noqa: E731
    for it in range(3):
        la, _ = a.step(batch)
        lb, _ = b.step(batch)
        U.assert_close(la, lb, 1e-10, f"loss at step {it}")
    U.assert_close(a.flat.flat, flat_b(), 1e-9, "parameters after 3 Adam steps")
    assert int(a.step_dev.item()) == 3


WIDE = ((2, 4, 7, 8), (8, 6, 5, 3))
@pytest.mark.parametrize("maxdim", [2, 3])
@pytest.mark.parametrize("latent", ["mean", "max", "min", "mean&max", "max&min", "min+max", "mean&min&max", "mean+min+max",
                                    "Mean&Max&Min&Mean", "mix"])
def test_latent_poolings_native_calls_match_per_op_path(latent, maxdim):
    _latent_case(latent, maxdim, 12, 5)


@pytest.mark.parametrize("N,B", [(30, 3), (150, 2), (7, 4)])
@pytest.mark.parametrize("latent", ["mix", "mean&min&max", "min+max"])
def test_latent_poolings_other_jet_sizes(latent, N, B):
    """The same at jet sizes where the particle loops of the pooling kernels run partial rounds (7, 30) and where the junction's
    two stages share their LDS (150: with 'mix' the latent weights alone are 86 KB)."""
    _latent_case(latent, 2, N, B)


def test_latent_stage_beyond_lds_is_refused_at_plan_time():
    """Four pooled blocks of 8 latent vectors for 150 particles: the decoder's input stage would need 165 KB of LDS (three blocks fit
    since round 6: 150 KB).  Decided at PLAN time (lgn_*_lds_bytes, lgn/_native.py: end_stages_fit): the whole-step class refuses with
    NotImplementedError, the chooser falls back to the captured module step, the decoder takes the per-operator path -- nothing fails
    at a launch."""
    import __graft_entry__ as G
    from lgn.step import CapturedModuleStep, NativeTrainStep, TrainStep, native_train_step
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    N, B, latent = 150, 2, "mean&max&min&mean"
    enc, dec = G._models(N, (2, 3, 3, 4), (4, 3, 3, 2), dev, seed=7, map_to_latent=latent)
    enc2, dec2 = G._models(N, (2, 3, 3, 4), (4, 3, 3, 2), dev, seed=7, map_to_latent=latent)
    p4, labels = O.synthetic_jets(B, N, seed=11, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    assert not dec._fused_ok()          # (the decoder's input stage does not fit)
    with pytest.raises(NotImplementedError, match="LDS"):
        NativeTrainStep(enc, dec, batch_size=B, optimizer=False, use_graph=False)
    a = native_train_step(enc, dec, B, optimizer=False, use_graph=False)
    assert isinstance(a, CapturedModuleStep)
    la, ra = a.step(batch)
    enc2.use_fused = dec2.use_fused = False                 # every operator on its own native call
    lb, rb = TrainStep(enc2, dec2, optimizer=False).forward_backward(batch)
    U.assert_close(la, lb, 1e-11, "loss")
    U.assert_close(ra, rb, 1e-11, "recon")


def _latent_case(latent, maxdim, N, B):
    """--map-to-latent variants (aggregate(), lgn/models/lgn_encoder.py:419-496; 'mean+max' is pinned by the reference fixture g7,
    the pooling operators one by one by test_gpu_parity.py) through the three native routes -- whole step, one call per network
    under autograd, junction kernels with the decoder taking P x tau_v latent vectors -- against the per-operator module path
    on the same weights."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    chans = ((2, 3, 3, 4), (4, 3, 3, 2))
    nets = [G._models(N, chans[0], chans[1], dev, seed=7, maxdim=maxdim, map_to_latent=latent) for _ in range(3)]
    for enc, dec in nets[:2]:
        assert enc._fused_ok() and dec._fused_ok()
    nets[2][0].use_fused = nets[2][1].use_fused = False
    p4, labels = O.synthetic_jets(B, N, seed=11, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    a = NativeTrainStep(*nets[0], batch_size=B, optimizer=False, use_graph=False)
    b = TrainStep(*nets[1], optimizer=False)
    c = TrainStep(*nets[2], optimizer=False)
    la, ra = a.step(batch)
    lb, rb = b.forward_backward(batch)
    lc, rc = c.forward_backward(batch)
    lat_b, lat_c = nets[1][0](batch), nets[2][0](batch)
    for k in lat_c.keys():
        assert lat_b[k].shape == lat_c[k].shape
        U.assert_close(lat_b[k], lat_c[k], 1e-12, f"latent {k}")
    for tag, (l, r, st) in {"whole step": (la, ra, a), "per-network calls": (lb, rb, b)}.items():
        U.assert_close(l, lc, 1e-11, f"{tag}: loss")
        U.assert_close(r, rc, 1e-11, f"{tag}: recon")
        for (k, g), (_, ref) in zip(list(st.encoder.named_grads()) + list(st.decoder.named_grads()),
                                    list(c.encoder.named_grads()) + list(c.decoder.named_grads())):
            if ref.abs().max() == 0:
                assert g.abs().max() == 0, f"{tag}: {k} must have exactly zero gradient"
            else:
                U.assert_close(g, ref, 1e-9, f"{tag}: grad {k}")


CFG5 = ((4, 4, 6, 6), (6, 6, 4, 4))
NARROW = ((3, 3, 4, 4), (4, 4, 3, 3))


# 40, 70: beyond the tile-blocked kernels (N <= 32) -- node-major layouts, all-channels-in-flight moments kernels
# 70 particles with cfg5's channels: the jet's packed features (134 KB + gradients) no longer fit a CU's LDS -- refused until round 5,
# now read from global memory by the pair-sweep kernels (generic_moments.hip: XL = false); the reference fixture g12 has 150
@pytest.mark.parametrize("N,chans", [(13, WIDE), (32, WIDE), (40, WIDE), (70, NARROW), (70, ((4, 4, 6, 6), (6, 6, 4, 4)))])
def test_native_step_maxdim3_other_shapes_match_per_op_path(N, chans):
    """Table-driven native step at shapes the golden fixture does not have -- channel counts 2..8 (every padded output width of
    the compile-time-table kernels, both level kinds), jets of 13 and of 32 particles (tiles of 64 nodes cut jets at other
    places, the last tile partly empty), zero-padded jets -- against the per-operator module path (each operator oracle-tested
    on its own in test_gpu_parity.py) on the same weights."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    enc, dec = G._models(N, chans[0], chans[1], dev, seed=5, maxdim=3)
    enc2, dec2 = G._models(N, chans[0], chans[1], dev, seed=5, maxdim=3)
    enc2.use_fused = dec2.use_fused = False
    B = 5 if N <= 40 else 2
    p4, labels = O.synthetic_jets(B, N, seed=4, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    a = NativeTrainStep(enc, dec, batch_size=B, optimizer=False, use_graph=False)
    b = TrainStep(enc2, dec2, optimizer=False)
    la, ra = a.step(batch)
    lb, rb = b.forward_backward(batch)
    U.assert_close(la, lb, 1e-11, "loss")
    U.assert_close(ra, rb, 1e-11, "recon")
    U.assert_close(a.flat.grad, b.flat.grad, 1e-9, "flat gradient")


@pytest.mark.parametrize("maxdim,ch_enc,ch_dec,N,B", [(3, (2, 4, 7, 8), (8, 6, 5, 3), 13, 5), (3, (4, 4, 6, 6), (6, 6, 4, 4), 30, 3),
                                                     (2, (3, 3, 4, 4), (4, 4, 3, 3), 30, 3), (2, (3, 3, 4, 4), (4, 4, 3, 3), 50, 2)])
def test_poisoned_buffers_padding_lanes_never_reach_a_result(maxdim, ch_enc, ch_dec, N, B, monkeypatch):
    """Every workspace / activation / scratch buffer pre-filled with NaN (B * N is not a multiple of the 64-node tile, so the
    last tile of the tile-blocked maxdim-3 layouts has padding lanes nobody writes): the native step and the whole-network
    module calls must still give finite results equal to the per-operator path -- a kernel that read a scalar it did not
    write, or summed padding lanes into a weight gradient (0 * NaN), would show up as NaN."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep, TrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    assert (B * N) % 64 != 0
    nets = [G._models(N, ch_enc, ch_dec, dev, seed=5, maxdim=maxdim) for _ in range(3)]
    nets[2][0].use_fused = nets[2][1].use_fused = False
    p4, labels = O.synthetic_jets(B, N, seed=4, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    ref = TrainStep(*nets[2], optimizer=False)
    lb, rb = ref.forward_backward(batch)
    a = NativeTrainStep(*nets[0], batch_size=B, optimizer=False, use_graph=False)
    a.workspace.fill_(float("nan"))
    a.recon.fill_(float("nan"))
    a.flat.grad_buf.fill_(float("nan"))
    la, ra = a.step(batch)
    assert torch.isfinite(a.flat.grad).all() and torch.isfinite(ra).all() and torch.isfinite(la)
    U.assert_close(la, lb, 1e-11, "native step: loss")
    U.assert_close(a.flat.grad, ref.flat.grad, 1e-9, "native step: flat gradient")
    monkeypatch.setenv("LGN_AMD_POISON", "1")
    m = TrainStep(*nets[1], optimizer=False)          # whole-network native calls under autograd
    lm, rm = m.forward_backward(batch)
    assert torch.isfinite(m.flat.grad).all() and torch.isfinite(rm).all()
    U.assert_close(lm, lb, 1e-11, "module path: loss")
    U.assert_close(m.flat.grad, ref.flat.grad, 1e-9, "module path: flat gradient")


def test_deepcopy_of_a_network_that_has_run():
    """copy.deepcopy(encoder) after a GPU forward (EMA copy, best-model snapshot): the native cache (ctypes descriptors,
    device tables) is rebuilt for the copy, which then computes the same result from its own parameter block."""
    import copy
    import __graft_entry__ as G
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    for maxdim, che, chd in ((2, (3, 3, 4, 4), (4, 4, 3, 3)), (3, (4, 4, 6, 6), (6, 6, 4, 4))):
        enc, dec = G._models(30, che, chd, dev, seed=2, maxdim=maxdim)
        p4, labels = O.synthetic_jets(3, 30, seed=1, pad=True)
        batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
        r0 = dec(enc(batch)).detach().clone()
        assert "_native_cache" in enc.__dict__
        enc2, dec2 = copy.deepcopy(enc), copy.deepcopy(dec)
        assert enc2.flat_params.data_ptr() != enc.flat_params.data_ptr()
        with torch.no_grad():
            enc.flat_params.zero_(); dec.flat_params.zero_()          # the copies must not look at the originals
        r1 = dec2(enc2(batch)).detach()
        assert torch.equal(r0, r1)


@pytest.mark.parametrize("B,N,che,chd", [(40, 30, (4, 4, 6, 6), (6, 6, 4, 4)), (5, 13, (2, 3, 4), (4, 3, 2))])
def test_native_step_maxdim3_fused_tail_is_bit_identical_to_the_separate_launches(B, N, che, chd, monkeypatch):
    """Round 6: the table-driven step leaves its CatMix partial rows in parameter layout, so its tail is the fused launch of the
    maxdim-2 step as well (csrc/step_tail.hip) -- against LGN_AMD_SPLIT_TAIL=1 (reduce_segments, rad_finalize_batch, l1_adam): three
    Adam steps, weights / moments / gradients / step counter bit for bit, the loss value to rounding."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    p4, labels = O.synthetic_jets(B, N, seed=B + N, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    runs = []
    for split in (False, True):
        enc, dec = G._models(N, che, chd, dev, seed=21, maxdim=3)
        if split:
            monkeypatch.setenv("LGN_AMD_SPLIT_TAIL", "1")
        st = NativeTrainStep(enc, dec, batch_size=B, lr=1e-3, l1_lambda=1e-6, use_graph=True)
        losses = [st.step(batch)[0].clone() for _ in range(3)]
        if split:
            monkeypatch.delenv("LGN_AMD_SPLIT_TAIL")
        torch.cuda.synchronize()
        runs.append((torch.stack(losses), st.flat.grad.clone(), st.adam_m.clone(), st.adam_v.clone(), st.flat.flat.clone(), st.step_dev.clone(),
                     st._loss_buf[3:].clone()))
    U.assert_close(runs[0][0], runs[1][0], 1e-13, "losses of the three steps")
    for what, x, y in zip(("gradients (with the L1 term)", "Adam m", "Adam v", "weights", "step counter"), runs[0][1:6], runs[1][1:6]):
        assert torch.equal(x, y), f"{what}: fused tail != separate launches (max diff {(x.double() - y.double()).abs().max().item():.3e})"
    assert int(runs[0][5].item()) == 3
    assert float(runs[0][6][:-12].abs().sum()) == 0.0 and float(runs[0][6][-1]) == 0.0, "the fused launch left slots / counters behind"


@pytest.mark.parametrize("B", [512, 9])
def test_native_step_split_and_fused_tail_share_one_scratch_block(B):
    """lgn_step_finalize_f64 (l1_adam: positive |w| partials in the scratch slots) followed by lgn_step_train_f64 (step_tail: reads
    "zero slot = not yet written in this launch") on ONE loss_out block -- the mix include/lgn_amd.h allows: every kernel leaves the
    slots at zero, the fused launch reports the same loss terms as a step object that never saw the other kernel."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    p4, labels = O.synthetic_jets(B, 30, seed=B, pad=True)
    batch = {"p4": p4.to(dev), "labels": labels.to(dev)}
    outs = []
    for mix in (False, True):
        enc, dec = G._models(30, (3, 3, 4, 4), (4, 4, 3, 3), dev, seed=5)
        st = NativeTrainStep(enc, dec, batch_size=B, lr=1e-3, l1_lambda=1e-3, use_graph=False, optimizer=False)
        st.load_batch(batch)
        if mix:
            st._fwd_bwd()
            st._finalize(False)
            torch.cuda.synchronize()
            assert float(st._loss_buf[3:].abs().sum()) == 0.0, "l1_adam left |w| partials / counters behind"
        st._train(False)
        torch.cuda.synchronize()
        assert float(st._loss_buf[3:].abs().sum()) == 0.0, "the fused tail left slots / counters behind"
        outs.append((st.loss_out.clone(), st.flat.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]), (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0][2]) > 0.0 and torch.isfinite(outs[0][0]).all()


@pytest.mark.parametrize("maxdim,ch_enc,ch_dec", [(2, (3, 3, 4, 4), (4, 4, 3, 3)), (3, (4, 4, 6, 6), (6, 6, 4, 4))])
def test_native_step_trains(maxdim, ch_enc, ch_dec):
    """Sixty graph-replayed Adam steps on one synthetic batch: the loss must fall well below its starting value and stay finite
    (the whole native step -- forward, Chamfer, backward, L1, Adam, device-side step counter -- as a training loop would run it)."""
    import __graft_entry__ as G
    from lgn.step import NativeTrainStep
    from oracle import lgn_oracle as O
    dev = torch.device("cuda:0")
    enc, dec = G._models(30, ch_enc, ch_dec, dev, seed=0, maxdim=maxdim)
    p4, labels = O.synthetic_jets(32, 30, seed=0, pad=True)
    step = NativeTrainStep(enc, dec, batch_size=32, lr=5e-4, l1_lambda=1e-8, use_graph=True)
    step.load_batch({"p4": p4.to(dev), "labels": labels.to(dev)})
    losses = [float(step.step()[0]) for _ in range(60)]
    assert all(l == l and l < float("inf") for l in losses), "non-finite loss"
    assert losses[-1] < 0.75 * losses[0], f"loss did not fall: {losses[0]:.4f} -> {losses[-1]:.4f}"
