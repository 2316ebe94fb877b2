"""Training kernels (forward + backward + clip + AdamW) against torch autograd of the oracle
and against one optimiser step captured from the reference's own FlowModel."""
import os

import numpy as np
import pytest
import torch

from helpers import DEV, rel_err
from nessai_amd.flowmodel import FlowModel
from nessai_amd.flows import NeuralSplineFlow, RealNVP
from nessai_amd.train import Trainer
from oracle import flows as O

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_goldens.npz"))

CASES = {
    "nsf_d4": (NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2), {}),
    "nsf_d7_k4": (NeuralSplineFlow, O.NeuralSplineFlow, (7, 20, 3, 1), dict(num_bins=4)),
    "nsf_c3": (NeuralSplineFlow, O.NeuralSplineFlow, (32, 64, 6, 2), {}),
    # bin counts without a fused training kernel (any num_bins from 2 to 16: the per-item spline kernels)
    "nsf_k5": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 2, 2), dict(num_bins=5)),
    "nsf_k13_lu": (NeuralSplineFlow, O.NeuralSplineFlow, (5, 24, 2, 1), dict(num_bins=13, linear_transform="lu")),
    "nsf_d6_tanh": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 2, 2), dict(activation=torch.tanh)),
    "rnvp_perm": (RealNVP, O.RealNVP, (6, 16, 3, 2),
                  dict(linear_transform="permutation", batch_norm_between_layers=False)),
    "rnvp_additive": (RealNVP, O.RealNVP, (5, 8, 2, 1),
                      dict(linear_transform="permutation", batch_norm_between_layers=False,
                           use_volume_preserving=True)),
    "rnvp_custom_scale": (RealNVP, O.RealNVP, (6, 16, 3, 2),
                          dict(linear_transform="permutation", batch_norm_between_layers=False,
                               scale_activation=lambda x: torch.sigmoid(x - 0.5) + 0.05)),
    # wider than the second-generation kernels take (hidden width > 64): the first-generation fused kernels
    "nsf_wide": (NeuralSplineFlow, O.NeuralSplineFlow, (10, 96, 2, 2), {}),
    # config 4's shape (64 inputs, 128-wide conditioners, 736 spline parameters per row): the third-generation
    # kernels (16-row workgroups); the same width on a flow with 92 parameters per row (a last group of the final
    # layer's gradient that is not whole, one residual block), and on an affine flow
    "nsf_c4": (NeuralSplineFlow, O.NeuralSplineFlow, (64, 128, 2, 2), {}),
    "nsf_d9_h128": (NeuralSplineFlow, O.NeuralSplineFlow, (9, 128, 2, 1), {}),
    # config 4's flow itself: six coupling layers (six ranges of the gradient copies), and a narrower one of 32 inputs.
    # (tanh: with relu and the residual branches scaled up by pair(), one of the ~250 000 pre-activations of a batch
    # lies within float32 rounding of zero - the device and torch then differentiate different branches of the kink,
    # one row's worth of gradient apart: scripts/debug_grad_rows.py finds the row, and with a smooth activation the
    # same shapes agree to 3e-7)
    "nsf_c4_six": (NeuralSplineFlow, O.NeuralSplineFlow, (64, 128, 6, 2), dict(activation=torch.tanh)),
    "nsf_d32_h96_six": (NeuralSplineFlow, O.NeuralSplineFlow, (32, 96, 6, 2), {}),
    "rnvp_d40_h128": (RealNVP, O.RealNVP, (40, 128, 2, 2), dict(linear_transform="permutation",
                                                               batch_norm_between_layers=False)),
    # the other widths of the 16-row kernels (80, 96, 112: the width at run time, MFMA steps and tiles beyond it
    # skipped) - the reference's default n_neurons = 2 n_inputs at 40, 48 and 56 inputs
    "nsf_d48_h96": (NeuralSplineFlow, O.NeuralSplineFlow, (48, 96, 2, 2), {}),
    "nsf_d16_h112": (NeuralSplineFlow, O.NeuralSplineFlow, (16, 112, 2, 1), {}),
    "rnvp_d40_h80": (RealNVP, O.RealNVP, (40, 80, 2, 2), dict(linear_transform="permutation",
                                                             batch_norm_between_layers=False)),
    # widths that are not a multiple of 16: a partial last tile (100 = 6 tiles + 4 columns, 68 = 4 + 4, 104 = 6 + 8);
    # and the reference's default at 50 inputs, whose 575 parameters per row are not a multiple of 4 - the 16-row
    # forward kernel with the first generation's backward kernel
    "nsf_d48_h100": (NeuralSplineFlow, O.NeuralSplineFlow, (48, 100, 2, 2), {}),
    "nsf_d16_h68": (NeuralSplineFlow, O.NeuralSplineFlow, (16, 68, 2, 1), {}),
    "nsf_d50_h100": (NeuralSplineFlow, O.NeuralSplineFlow, (50, 100, 2, 2), {}),
    "rnvp_d52_h104": (RealNVP, O.RealNVP, (52, 104, 2, 2), dict(linear_transform="permutation",
                                                               batch_norm_between_layers=False)),
    # parameter rows that are not a multiple of 16 bytes (np = 3 x 23): the scalar paths of the backward kernel
    "nsf_d5": (NeuralSplineFlow, O.NeuralSplineFlow, (5, 24, 2, 2), {}),
    # four residual blocks per conditioner: nine activation tiles, beyond what the backward kernel copies through
    # registers (its async-copy path)
    "nsf_deep": (NeuralSplineFlow, O.NeuralSplineFlow, (8, 16, 2, 4), {}),
    # config 5's shape: 15 dimensions, 30 neurons (a hidden width that is not a multiple of 4: row-by-row tile copies)
    "nsf_c5": (NeuralSplineFlow, O.NeuralSplineFlow, (15, 30, 4, 2), {}),
    "rnvp_c2_default": (RealNVP, O.RealNVP, (16, 32, 4, 2), {}),  # LU-linear + batch norm
    "rnvp_lu_only": (RealNVP, O.RealNVP, (5, 8, 2, 2), dict(batch_norm_between_layers=False)),
    "nsf_bn_lu": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 3, 2),
                  dict(batch_norm_between_layers=True, linear_transform="lu")),
    # plain MLP conditioner (src/nessai/flows/nets.py:12-126), the "MLP" variant of config 2
    "rnvp_c2_mlp": (RealNVP, O.RealNVP, (16, 32, 4, 2), dict(net="mlp")),
    # per-layer masks (flows/realnvp.py:100-110) whose fourth coupling has more identity features than the run's
    # tile is wide: the forward run of the second-generation kernels stops after THREE couplings, a length that is
    # not instantiated (6 / 4 / 2 / 1) and must be cut back to two, not handed to the six-coupling kernel
    "rnvp_ragged_masks": (RealNVP, O.RealNVP, (40, 8, 5, 1),
                          dict(mask=[[-1.0] * 4 + [1.0] * 36] * 3 + [[-1.0] * 30 + [1.0] * 10] + [[-1.0] * 4 + [1.0] * 36],
                               linear_transform=None, batch_norm_between_layers=False)),
    # the reference's defaults at 100 inputs: 200-wide conditioners (the layer-wise coupling kernels) and an
    # LU-linear layer over 100 features (beyond the one-launch LU kernels' 64)
    "rnvp_d100_default": (RealNVP, O.RealNVP, (100, 200, 2, 1), {}),
    "rnvp_mlp_deep_tanh": (RealNVP, O.RealNVP, (5, 12, 2, 3),
                           dict(net="mlp", linear_transform="permutation", batch_norm_between_layers=False,
                                activation=torch.tanh)),
}


def pair(name, seed=3):
    P, Oc, args, kw = CASES[name]
    torch.manual_seed(seed)
    ref = Oc(*args, **kw)
    # make the residual branches non-trivial (their last layers start at ~1e-3)
    with torch.no_grad():
        for n_, p in ref.named_parameters():
            if "linear_layers.1" in n_:
                p.mul_(200.0)
        g = torch.Generator().manual_seed(seed)
        for n_, p in ref.named_parameters():  # LU / batch-norm parameters off their trivial values
            if "entries" in n_ or "unconstrained" in n_:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    flow = P(*args, **kw)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    return ref, flow, args[0]


def flat_of(ref):
    return torch.cat([p.detach().reshape(-1) for p in ref.parameters()])


@pytest.mark.parametrize("name", sorted(CASES))
def test_loss_and_gradients_match_autograd(name):
    ref, flow, D = pair(name)
    x = torch.randn(333, D) * 1.5  # not a multiple of 32; some rows beyond the tail bound
    ref.train()
    loss_ref = -ref.log_prob(x).mean()
    loss_ref.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    tr = Trainer(flow)
    before = tr.params.clone()
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0)
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    g = tr.grads.cpu()
    scale = max(1e-3, float(g_ref.abs().max()))
    assert float((g - g_ref).abs().max()) <= 2e-4 * scale
    assert rel_err(torch.linalg.norm(g), torch.linalg.norm(g_ref), floor=1e-6) <= 1e-4
    # lr = 0 and decoupled weight decay lr*wd = 0: weights untouched
    assert torch.equal(tr.params, before)


@pytest.mark.parametrize("name", ["nsf_c3", "nsf_d7_k4", "nsf_d6_tanh", "rnvp_perm", "rnvp_additive", "nsf_bn_lu",
                                  "nsf_c4", "nsf_d9_h128"])
def test_first_generation_fused_kernels_match_autograd(name, monkeypatch):
    """The first-generation per-coupling kernels (nsa_train_fused.h) remain the cross-check of the second and third
    generations (and the backward pass of widths between 65 and 127): same bar, on the shapes the later generations
    now take."""
    monkeypatch.setenv("NESSAI_AMD_TRAIN_V2", "0")
    monkeypatch.setenv("NESSAI_AMD_TRAIN_V3", "0")
    test_loss_and_gradients_match_autograd(name)


def test_couplings_per_forward_launch_do_not_change_the_step(monkeypatch):
    """One, two, four or six coupling layers per forward launch (nsa_train_fused2.h, NESSAI_AMD_TRAIN_RUN) are the
    same arithmetic up to the compiler's choice of fused multiply-adds per instantiation and the order log|det J| is
    summed in.  On a smooth (tanh) conditioner that is rounding noise; with relu a pre-activation within an ulp of
    zero can fall on the other side of the kink in one row (scripts/fuzz_train.py: "kink_cases"), which is why this
    is not asserted on config 3."""
    results = {}
    for run in ("1", "2", "4", "6"):
        monkeypatch.setenv("NESSAI_AMD_TRAIN_RUN", run)
        torch.manual_seed(5)
        ref = O.NeuralSplineFlow(8, 16, 6, 2, activation=torch.tanh)
        with torch.no_grad():
            for n_, p_ in ref.named_parameters():
                if "linear_layers.1" in n_:
                    p_.mul_(200.0)
        flow = NeuralSplineFlow(8, 16, 6, 2, activation=torch.tanh)
        flow.load_state_dict(ref.state_dict())
        flow.to(DEV)
        x = (torch.randn(1000, 8, generator=torch.Generator().manual_seed(11)) * 1.2).to(DEV)
        tr = Trainer(flow)
        tr.step(x, clip_norm=5.0, lr=0.0)
        torch.cuda.synchronize()
        results[run] = (tr.loss_acc.item(), tr.grads.cpu().clone())
    l1, g1 = results["1"]
    scale = max(1e-3, float(g1.abs().max()))
    for run in ("2", "4", "6"):
        l, g = results[run]
        assert abs(l - l1) <= 2e-6 * max(1.0, abs(l1)), run
        assert float((g - g1).abs().max()) <= 1e-5 * scale, run


def test_eval_matches_inference_kernel():
    ref, flow, D = pair("nsf_c3")
    x = torch.randn(500, D, device=DEV)
    tr = Trainer(flow)
    lp = torch.empty(500, dtype=torch.float32, device=DEV)
    tr.evaluate(x, log_prob_out=lp)
    flow.eval()
    flow._image_key = None
    with torch.no_grad():
        lp2 = flow.log_prob(x)
    assert rel_err(lp.cpu(), lp2.cpu()) <= 1e-5
    assert abs(tr.loss_acc.item() + lp2.mean().item()) <= 1e-4 * abs(lp2.mean().item())


def state_of(prefix):
    pre = prefix + "/state/"
    return {k[len(pre):]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith(pre)}


def test_one_step_matches_reference_flowmodel():
    """Loss, gradients, clipped norm and post-AdamW weights of ONE optimiser step captured
    from the reference's FlowModel (tests/golden/make_golden.py, G4)."""
    flow = NeuralSplineFlow(4, 8, 2, 2)
    flow.load_state_dict(state_of("flowmodel/before"))
    flow.to(DEV)
    tr = Trainer(flow)  # AdamW defaults: lr 1e-3, betas (0.9, 0.999), eps 1e-8, wd 0.01
    x = torch.from_numpy(GOLD["flowmodel/step_x"]).to(DEV)
    tr.step(x, clip_norm=5.0)
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - float(GOLD["flowmodel/step_loss"])) <= 2e-5
    names = [k for k, _ in flow.named_parameters()]
    g_ref = torch.cat([torch.from_numpy(GOLD[f"flowmodel/grad/{k}"]).reshape(-1) for k in names])
    g = tr.grads.cpu()
    assert float((g - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    assert rel_err(torch.linalg.norm(g), float(GOLD["flowmodel/step_grad_norm"]), floor=1e-6) <= 1e-4
    after = state_of("flowmodel/after")
    w_ref = torch.cat([after[k].reshape(-1) for k in names])
    w = tr.params.cpu()
    # Adam's first step moves every weight by ~lr*sign(g): compare the update itself
    before = torch.cat([state_of("flowmodel/before")[k].reshape(-1) for k in names])
    upd, upd_ref = w - before, w_ref - before
    assert float((upd - upd_ref).abs().max()) <= 2e-5  # lr = 1e-3
    # parameters are views of the flat buffer: state_dict follows the update
    sd = flow.state_dict()
    for k in names:
        assert torch.allclose(sd[k].cpu(), w_ref[: 0].new_tensor(after[k].numpy()), atol=2e-5)


def test_flowmodel_train_end_to_end(tmp_path):
    """FlowModel.train: loss goes down, history/early-stopping protocol of the reference,
    weights file written, inference image follows the trained weights."""
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    data = rng.normal(size=(1000, 4)) * np.array([1.0, 0.3, 2.0, 0.5]) + np.array([0.5, -1.0, 0.0, 2.0])
    data[:, 1] += 0.8 * data[:, 0] ** 2
    fm = FlowModel(
        flow_config=dict(n_inputs=4, n_neurons=16, n_blocks=4, n_layers=2, ftype="nsf"),
        training_config=dict(batch_size=250, max_epochs=60, patience=10),
        output=str(tmp_path), rng=rng,
    )
    fm.initialise()
    lp_before = fm.log_prob(data).mean()
    history = fm.train(data)
    assert set(history) == {"loss", "val_loss"}
    assert 1 <= len(history["loss"]) <= 60 and len(history["loss"]) == len(history["val_loss"])
    assert history["loss"][-1] < history["loss"][0] - 0.5
    assert os.path.exists(os.path.join(str(tmp_path), "model.pt"))
    lp_after = fm.log_prob(data).mean()
    assert lp_after > lp_before + 0.5
    # the best validation loss is what the restored weights give
    x_val = fm.numpy_array_to_tensor(data)
    assert np.isfinite(lp_after)
    # a second train() call continues from the Adam state (optimiser persists, base.py:174)
    steps = fm._trainer.step_count
    fm.train(data, max_epochs=2)
    assert fm._trainer.step_count > steps
    fm.reset_model()
    assert fm._trainer is None
    del x_val


def test_weighted_loss_and_gradients_match_autograd():
    """Weighted KL of the importance sampler: -sum(w log_prob) / sum(w) (flowmodel/base.py:404-407)."""
    ref, flow, D = pair("nsf_d4")
    x = torch.randn(200, D)
    w = torch.rand(200) + 0.1
    ref.train()
    loss_ref = -torch.sum(ref.log_prob(x) * w) / torch.sum(w)
    loss_ref.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    tr = Trainer(flow)
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0, weights=w.to(DEV))
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    g = tr.grads.cpu()
    assert float((g - g_ref).abs().max()) <= 2e-4 * max(1e-3, float(g_ref.abs().max()))


def test_importance_flow_model(tmp_path):
    """ImportanceFlowModel: add flows, log_prob_all == per-flow log_prob, sample_ith, weighted train."""
    from nessai_amd.flowmodel_importance import ImportanceFlowModel

    rng = np.random.default_rng(1)
    torch.manual_seed(1)
    ifm = ImportanceFlowModel(
        flow_config=dict(n_inputs=3, n_neurons=8, n_blocks=2, n_layers=1, ftype="nsf"),
        training_config=dict(batch_size=100, max_epochs=5, patience=5), output=str(tmp_path), rng=rng,
    )
    assert ifm.model is None and ifm.n_models == 0
    ifm.add_new_flow()
    data = rng.normal(size=(300, 3))
    hist = ifm.train(data, weights=rng.uniform(0.5, 1.5, size=300))
    assert len(hist["loss"]) >= 1 and np.isfinite(hist["loss"]).all()
    ifm.add_new_flow()  # copy of the previous level
    ifm.add_new_flow(reset=True)
    assert ifm.n_models == 3
    x = rng.normal(size=(50, 3))
    all_lp = ifm.log_prob_all(x)
    assert all_lp.shape == (50, 3)
    for i in range(3):
        np.testing.assert_allclose(all_lp[:, i], ifm.log_prob_ith(x, i), rtol=0, atol=1e-6)
    np.testing.assert_allclose(all_lp[:, 0], all_lp[:, 1], rtol=0, atol=1e-6)  # level 1 copied level 0
    assert ifm.sample_ith(2, 7).shape == (7, 3)
    assert len(ifm.weights_files) == 1


def test_importance_flow_proposal_levels(tmp_path):
    """Two levels of the importance sampler's proposal: train on weighted samples, draw, meta-proposal."""
    from nessai_amd.model import GaussianModel
    from nessai_amd.proposal_importance import ImportanceFlowProposal

    rng = np.random.default_rng(2)
    torch.manual_seed(2)
    model = GaussianModel(3)
    prop = ImportanceFlowProposal(
        model, output=str(tmp_path), rng=rng,
        flow_config=dict(n_neurons=8, n_blocks=2, n_layers=1, ftype="nsf"),
        training_config=dict(batch_size=200, max_epochs=5, patience=5),
    )
    prop.initialise()
    samples, log_q0 = prop.draw_from_prior(400)
    assert log_q0.shape == (400, 1) and np.allclose(samples["logQ"], 0.0)
    prop.train(samples)  # weighted KL with exp(logW)
    prop.update_proposal_weights({-1: 0.5, 0: 0.5})
    new, log_q = prop.draw(100)
    assert new.size == 100 and log_q.shape == (100, 2)
    assert model.in_unit_hypercube(new).all()
    assert np.isfinite(new["logQ"]).all() and np.isfinite(new["logW"]).all()
    old_q = prop.update_log_q(samples, log_q0)
    assert old_q.shape == (400, 2)
    np.testing.assert_allclose(prop.compute_meta_proposal_from_log_q(log_q), new["logQ"], rtol=1e-12)
    prop.train(new)
    assert prop.flow.n_models == 2 and prop.level_count == 1


def test_batch_norm_running_statistics_follow_the_reference():
    """Training-mode batch norm updates running_mean / running_var with momentum 0.1 and the
    unbiased batch variance, exactly as the oracle's transform does."""
    ref, flow, D = pair("rnvp_c2_default")
    x = torch.randn(400, D) * 1.3 + 0.2
    ref.train()
    ref.log_prob(x)
    tr = Trainer(flow)
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0)
    torch.cuda.synchronize()
    ref_stats = {k: v for k, v in ref.state_dict().items() if "running" in k}
    got = {k: v.cpu() for k, v in flow.state_dict().items() if "running" in k}
    assert ref_stats and set(ref_stats) == set(got)
    for k in ref_stats:
        assert torch.allclose(got[k], ref_stats[k], rtol=1e-4, atol=1e-5), k


def test_act_norm_trains_with_data_dependent_initialisation():
    """nflows ActNorm: the first training batch sets log_scale = -log(std), shift = -mean(x / std);
    loss, gradients and the initialised parameters against the oracle (two steps)."""
    P, Oc, args = RealNVP, O.RealNVP, (6, 16, 2, 2)
    kw = dict(actnorm=True, batch_norm_between_layers=False, linear_transform="permutation")
    torch.manual_seed(3)
    ref = Oc(*args, **kw)
    flow = P(*args, **kw)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    ref.train()
    tr = Trainer(flow)
    g = torch.Generator().manual_seed(8)
    for step in range(2):
        x = torch.randn(257, 6, generator=g) * torch.tensor([1.0, 0.2, 3.0, 0.5, 2.0, 1.5]) + 0.7
        ref.zero_grad()
        loss_ref = -ref.log_prob(x).mean()
        loss_ref.backward()
        tr.loss_acc.zero_()
        tr.step(x.to(DEV), clip_norm=5.0, lr=0.0)
        torch.cuda.synchronize()
        assert abs(tr.loss_acc.item() - loss_ref.item()) <= 2e-5 * max(1.0, abs(loss_ref.item()))
        g_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        scale = max(1e-3, float(g_ref.abs().max()))
        assert float((tr.grads.cpu() - g_ref).abs().max()) <= 3e-4 * scale
        # lr = 0: the only parameter change is the data-dependent initialisation of step 0
        assert float((tr.params.cpu() - flat_of(ref)).abs().max()) <= 2e-6
    sd = flow.state_dict()
    assert all(bool(v) for k, v in sd.items() if k.endswith("initialized"))
    # eval mode goes through the inference kernel with the initialised parameters
    flow.eval(); ref.eval()
    flow._image_key = None
    xe = torch.randn(100, 6)
    with torch.no_grad():
        assert rel_err(flow.log_prob(xe.to(DEV)).cpu(), ref.log_prob(xe)) <= 1e-5


@pytest.mark.parametrize("name", ["nsf_c4", "nsf_d9_h128", "rnvp_d40_h128", "nsf_d48_h96", "nsf_d16_h112", "rnvp_d40_h80",
                                  "nsf_c4_six", "nsf_d32_h96_six", "nsf_d48_h100", "nsf_d16_h68", "nsf_d50_h100",
                                  "rnvp_d52_h104"])
def test_third_generation_fused_kernels_match_the_first(name, monkeypatch):
    """128-wide conditioners (nsa_train_fused3.h; config 4's importance-sampling flows): the third-generation kernels
    - 16-row workgroups on 16x16x4 matrix tiles - against the first generation (NESSAI_AMD_TRAIN_V3=0) on the same
    batch: loss identical, gradients equal up to the order of the float32 sums (whole-k tiles instead of chains of 32
    k, per-workgroup gradient copies instead of float atomics), ``log_prob`` of the validation pass likewise."""
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NESSAI_AMD_TRAIN_V3", mode)
        ref, flow, D = pair(name)
        x = (torch.randn(333, D, generator=torch.Generator().manual_seed(9)) * 1.5).to(DEV)
        tr = Trainer(flow)
        tr.step(x, clip_norm=5.0, lr=0.0)
        torch.cuda.synchronize()
        lp = torch.empty(333, dtype=torch.float32, device=DEV)
        loss = tr.loss_acc.item()
        tr.evaluate(x, log_prob_out=lp)
        results[mode] = (loss, tr.grads.cpu().clone(), lp.cpu().clone())
    (l0, g0, lp0), (l1, g1, lp1) = results["0"], results["1"]
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0))
    scale = max(1e-3, float(g0.abs().max()))
    assert float((g0 - g1).abs().max()) <= 5e-5 * scale
    assert rel_err(lp0, lp1) <= 2e-6


def test_third_generation_kernels_with_many_gradient_copies():
    """1100 rows of config 4's shape: 69 per-workgroup gradient copies, summed in a fixed order - more than one wave of
    workgroups' worth, and past the 256 MB that used to cap them."""
    ref, flow, D = pair("nsf_c4")
    x = torch.randn(1100, D, generator=torch.Generator().manual_seed(5)) * 1.5
    ref.train()
    loss_ref = -ref.log_prob(x).mean()
    loss_ref.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    tr = Trainer(flow)
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0)
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    g = tr.grads.cpu()
    scale = max(1e-3, float(g_ref.abs().max()))
    assert float((g - g_ref).abs().max()) <= 2e-4 * scale
    assert rel_err(torch.linalg.norm(g), torch.linalg.norm(g_ref), floor=1e-6) <= 1e-4


def test_beyond_the_gradient_copy_cap_the_first_generation_backward_takes_over():
    """NESSAI_AMD_TRAIN_SLAB_MB (read once per process: a child process here) caps the gradient copies; beyond it the
    16-row forward kernel runs with the first generation's backward kernel (float atomics) - the same gradients."""
    import subprocess
    import sys

    env = dict(os.environ, NESSAI_AMD_TRAIN_SLAB_MB="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                          "-k", "test_loss_and_gradients_match_autograd and (nsf_c4 or nsf_d9_h128 or nsf_c3)"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, \
        out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("shape,kw", [
    ((5, 68, 2, 1), {}), ((6, 72, 3, 2), {}), ((7, 100, 2, 3), {}), ((9, 128, 3, 1), {}), ((11, 84, 2, 2), {}),
    ((13, 124, 2, 1), {}), ((17, 128, 2, 2), dict(num_bins=4)), ((12, 96, 2, 1), dict(num_bins=10)),
    ((10, 128, 2, 2), dict(num_bins=16)), ((33, 76, 2, 2), {}), ((64, 128, 2, 3), {}), ((30, 104, 4, 1), dict(num_bins=4)),
])
def test_third_generation_kernels_over_widths_and_parameter_rows(shape, kw, monkeypatch):
    """Neural spline flows whose conditioner width is any multiple of 4 from 68 to 128 and whose parameter rows have
    any length - 16-byte multiples or not, aligned or pushed off 16-byte boundaries by the coupling before them, one
    to three residual blocks, 4 / 8 / 10 / 16 bins: the 16-row kernels (16-byte or element-wise parameter tile,
    transposed or natural final-layer weights, partial last tiles) against the first generation on 150 rows (nine
    full workgroups and one of six rows)."""
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NESSAI_AMD_TRAIN_V3", mode)
        torch.manual_seed(7)
        flow = NeuralSplineFlow(*shape, activation=torch.tanh, **kw)
        with torch.no_grad():
            for n_, p in flow.named_parameters():
                if "linear_layers.1" in n_:  # (the residual branches start at ~1e-3: make them count)
                    p.mul_(100.0)
        flow.to(DEV)
        x = (torch.randn(150, shape[0], generator=torch.Generator().manual_seed(9)) * 1.5).to(DEV)
        tr = Trainer(flow)
        tr.step(x, clip_norm=5.0, lr=0.0)
        torch.cuda.synchronize()
        results[mode] = (tr.loss_acc.item(), tr.grads.cpu().clone())
    (l0, g0), (l1, g1) = results["0"], results["1"]
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0))
    scale = max(1e-3, float(g0.abs().max()))
    assert float((g0 - g1).abs().max()) <= 5e-5 * scale, (float((g0 - g1).abs().max()), scale)


@pytest.mark.parametrize("name", ["nsf_c3", "nsf_d7_k4", "nsf_d6_tanh", "rnvp_perm", "rnvp_additive", "nsf_bn_lu"])
def test_fused_coupling_kernels_match_the_layer_wise_kernels(name, monkeypatch):
    """The fused per-coupling kernels (nsa_train_fused.h) and the layer-wise kernels are two
    implementations of the same step: loss identical, gradients equal up to summation order."""
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NESSAI_AMD_TRAIN_FUSED", mode)
        ref, flow, D = pair(name)
        x = (torch.randn(333, D, generator=torch.Generator().manual_seed(9)) * 1.5).to(DEV)
        tr = Trainer(flow)
        tr.step(x, clip_norm=5.0, lr=0.0)
        torch.cuda.synchronize()
        lp = torch.empty(333, dtype=torch.float32, device=DEV)
        loss = tr.loss_acc.item()
        tr.evaluate(x, log_prob_out=lp)
        results[mode] = (loss, tr.grads.cpu().clone(), lp.cpu().clone())
    (l0, g0, lp0), (l1, g1, lp1) = results["0"], results["1"]
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0))
    scale = max(1e-3, float(g0.abs().max()))
    assert float((g0 - g1).abs().max()) <= 5e-5 * scale
    if name != "nsf_bn_lu":  # eval mode only differs for batch norm (running statistics after one step)
        assert rel_err(lp0, lp1) <= 2e-6


def test_act_norm_is_initialised_by_the_first_minibatch_only():
    """Several minibatches through the graph-capturable step (``Trainer.step_dev``, what ``train_flow`` runs
    from the first epoch on): act norm takes its data-dependent initialisation from the FIRST batch and is
    then trained by the optimiser like every other parameter, as nflows' ActNorm does - three AdamW steps
    against torch on the oracle."""
    P, Oc, args = RealNVP, O.RealNVP, (6, 16, 2, 2)
    kw = dict(actnorm=True, batch_norm_between_layers=False, linear_transform="permutation")
    torch.manual_seed(3)
    ref = Oc(*args, **kw)
    flow = P(*args, **kw)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    ref.train()
    tr = Trainer(flow, lr=1e-2, weight_decay=0.01, optimiser="adamw")
    tr.opt[0] = 1e-2
    tr.opt[1] = 0.0
    g = torch.Generator().manual_seed(8)
    opt = None
    for step in range(3):
        x = torch.randn(300, 6, generator=g) * torch.tensor([1.0, 0.2, 3.0, 0.5, 2.0, 1.5]) + 0.7 * (step + 1)
        loss = -ref.log_prob(x).mean()  # the first pass initialises the oracle's act norm in place
        if opt is None:
            opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
        opt.step()
        tr.step_dev(x.to(DEV), clip_norm=5.0)
    torch.cuda.synchronize()
    assert float((tr.params.cpu() - flat_of(ref)).abs().max()) <= 5e-5


@pytest.mark.parametrize("optimiser,torch_cls,kwargs", [
    ("adam", torch.optim.Adam, dict(weight_decay=1e-6)),
    ("adam", torch.optim.Adam, dict(weight_decay=0.05)),
    ("sgd", torch.optim.SGD, dict()),
    ("adamw", torch.optim.AdamW, dict()),
])
def test_optimiser_choices_match_torch(optimiser, torch_cls, kwargs):
    """The three optimisers of FlowModel.get_optimiser (flowmodel/base.py:105-135): three steps on
    the device against the same torch optimiser on the oracle."""
    ref, flow, D = pair("nsf_d4")
    ref.train()
    opt = torch_cls(ref.parameters(), lr=1e-2, **kwargs)
    tr = Trainer(flow, lr=1e-2, weight_decay=kwargs.get("weight_decay", 0.01 if optimiser == "adamw" else 0.0),
                 optimiser=optimiser)
    g = torch.Generator().manual_seed(4)
    for _ in range(3):
        x = torch.randn(200, D, generator=g)
        opt.zero_grad()
        (-ref.log_prob(x).mean()).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
        opt.step()
        tr.step(x.to(DEV), clip_norm=5.0)
    torch.cuda.synchronize()
    assert float((tr.params.cpu() - flat_of(ref)).abs().max()) <= 3e-5


def test_annealing_and_noise_options(tmp_path):
    """Cosine annealing of the learning rate per epoch and Gaussian training noise
    (flowmodel/base.py:424-425, 449-450, 596-605, 628-631)."""
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    data = rng.normal(size=(600, 3)) * np.array([1.0, 0.3, 2.0])
    fm = FlowModel(
        flow_config=dict(n_inputs=3, n_neurons=8, n_blocks=2, n_layers=1, ftype="realnvp",
                         batch_norm_between_layers=False, linear_transform="permutation"),
        training_config=dict(batch_size=300, max_epochs=8, patience=100, annealing=True, noise_type="adaptive",
                             noise_scale=0.1, optimiser="adam"),
        output=str(tmp_path), rng=rng,
    )
    fm.initialise()
    history = fm.train(data)
    assert len(history["loss"]) == 8 and np.isfinite(history["loss"]).all()
    tr = fm._trainer
    assert tr.optimiser == Trainer.OPTIMISERS["adam"] and tr.weight_decay == pytest.approx(1e-6)
    # lr of the last epoch: 7 scheduler steps of CosineAnnealingLR(T_max=8) from 1e-3
    sched_ref = torch.optim.lr_scheduler.CosineAnnealingLR(
        torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3), 8)
    for _ in range(7):
        sched_ref.optimizer.step()
        sched_ref.step()
    assert tr.lr == pytest.approx(sched_ref.get_last_lr()[0], rel=1e-9)
    with pytest.raises(NotImplementedError):
        bad = FlowModel(flow_config=dict(n_inputs=3, n_neurons=8, n_blocks=2, n_layers=1),
                        training_config=dict(optimiser="rmsprop"), output=str(tmp_path), rng=rng)
        bad.initialise()
        bad.train(data, max_epochs=1)


def test_deterministic_training_is_bit_reproducible(tmp_path):
    """``training_config["deterministic"]``: the layer-wise kernels with fixed-order reductions - two
    fits from the same seed end with identical weights (the fused kernels' float atomics do not
    promise that) and agree with the default fit to float32 round-off."""
    from nessai_amd.flowmodel import FlowModel

    data = np.random.default_rng(3).normal(size=(600, 6))
    cfg = dict(n_inputs=6, n_neurons=16, n_blocks=3, n_layers=2, ftype="nsf")

    def fit(deterministic, tag):
        torch.manual_seed(7)
        fm = FlowModel(flow_config=cfg, training_config=dict(batch_size=250, max_epochs=8, patience=20,
                                                            deterministic=deterministic),
                       output=str(tmp_path / tag), rng=np.random.default_rng(7))
        fm.initialise()
        hist = fm.train(data)
        return fm.model.trainer().flat.detach().cpu().clone(), hist

    a, ha = fit(True, "a")
    b, hb = fit(True, "b")
    assert torch.equal(a, b) and ha["val_loss"] == hb["val_loss"]
    c, hc = fit(False, "c")
    assert float((a - c).abs().max()) < 5e-3 and abs(ha["loss"][-1] - hc["loss"][-1]) < 1e-2
    assert "NESSAI_AMD_TRAIN_FUSED" not in os.environ
