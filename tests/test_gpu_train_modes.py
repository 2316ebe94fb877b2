"""Training-mode options of the conditioners on the device (runs on a real MI355X): ``dropout_probability``
(nflows ResidualBlock: activation -> dropout -> second linear layer; ``src/nessai/flows/realnvp.py:136-146``,
``nsf.py:85-95``) against torch autograd of the oracle FED THE SAME MASKS, and ``batch_norm_within_layers`` (two
``BatchNorm1d(eps=1e-3)`` per block: batch statistics, running-statistics updates) against torch's own BatchNorm1d.

The device draws its masks from a counter-based stream (Philox4x32-10 keyed by {seed, step}, indexed by row, column
quad, coupling and block: ``csrc/nsa_train.hip`` ``tr_dropout_mask``) - restated here in numpy (``oracle/philox.py``)
and put in place of the oracle blocks' ``nn.Dropout``.
"""
import numpy as np
import pytest
import torch

from helpers import DEV, rel_err
from nessai_amd.flows import NeuralSplineFlow, RealNVP
from nessai_amd.train import Trainer
from oracle import flows as O
from oracle import philox

pytestmark = pytest.mark.gpu


def dropout_mask(seed, step, n, H, coupling, block, p):
    """``M[n][H]`` of ``tr_dropout_mask``: keep <=> Philox word >= round(p 2^32); kept values scale by 1 / (1 - p)."""
    rows = np.arange(n, dtype=np.uint64)
    quads = (H + 3) // 4
    lo = np.repeat((rows & np.uint64(0xFFFFFFFF)).astype(np.uint32), quads)
    hi = np.repeat((rows >> np.uint64(32)).astype(np.uint32), quads)
    q = np.tile(np.arange(quads, dtype=np.uint32), n)
    tag = np.uint32(0x44000000 | (coupling << 8) | block)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32(((seed >> 32) & 0xFFFFFFFF) ^ (step & 0xFFFFFFFF))
    words = np.stack(philox.philox4x32_10(lo, hi, q, np.full_like(q, tag), k0, k1), axis=1)  # (n * quads, 4)
    words = words.reshape(n, quads * 4)[:, :H]
    thr = min(int(np.float32(p).astype(np.float64) * 4294967296.0 + 0.5), 0xFFFFFFFF)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(np.where(words >= np.uint32(thr), scale, np.float32(0.0)).astype(np.float32))


class _FixedMask(torch.nn.Module):
    """Stands in for ``nn.Dropout`` in an oracle block: multiplies by a given mask while training."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask if self.training else x


def feed_masks(ref, seed, step, n, p):
    coupling = 0
    for t in ref._transform._transforms:
        net = getattr(t, "transform_net", None)
        if net is None:
            continue
        for l, block in enumerate(net.blocks):
            H = block.linear_layers[1].in_features
            block.dropout = _FixedMask(dropout_mask(seed, step, n, H, coupling, l, p))
        coupling += 1


CASES = {
    "nsf_d6": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 2, 2), dict(dropout_probability=0.25)),
    "nsf_d5_k5_h30": (NeuralSplineFlow, O.NeuralSplineFlow, (5, 30, 2, 1), dict(dropout_probability=0.5, num_bins=5)),
    "rnvp_default": (RealNVP, O.RealNVP, (8, 16, 3, 2), dict(dropout_probability=0.1)),  # LU + batch norm between
    "nsf_conditional": (NeuralSplineFlow, O.NeuralSplineFlow, (4, 16, 2, 2),
                        dict(dropout_probability=0.3, context_features=3)),
}


def pair(name, seed=3):
    P, Oc, args, kw = CASES[name]
    torch.manual_seed(seed)
    ref = Oc(*args, **kw)
    with torch.no_grad():  # make the residual branches non-trivial (their last layers start at ~1e-3)
        for n_, q in ref.named_parameters():
            if "linear_layers.1" in n_:
                q.mul_(200.0)
    flow = P(*args, **kw)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    return ref, flow, args[0], kw


@pytest.mark.parametrize("name", sorted(CASES))
def test_dropout_step_matches_autograd_with_the_same_masks(name):
    ref, flow, D, kw = pair(name)
    p, C = kw["dropout_probability"], kw.get("context_features", 0)
    n = 301
    x = torch.randn(n, D) * 1.3
    ctx = torch.randn(n, C) if C else None
    tr = Trainer(flow)
    assert tr.dropout_probability == pytest.approx(p) and tr.drop_state is not None
    seed, step = (int(v) for v in tr.drop_state.cpu())
    assert step == 0
    ref.train()
    feed_masks(ref, seed, step, n, p)
    loss_ref = -ref.log_prob(x, context=ctx).mean() if C else -ref.log_prob(x).mean()
    loss_ref.backward()
    g_ref = torch.cat([q.grad.reshape(-1) for q in ref.parameters()])
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0, context=ctx.to(DEV) if C else None)
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    g = tr.grads.cpu()
    scale = max(1e-3, float(g_ref.abs().max()))
    assert float((g - g_ref).abs().max()) <= 2e-4 * scale
    assert rel_err(torch.linalg.norm(g), torch.linalg.norm(g_ref), floor=1e-6) <= 1e-4
    # the step ticked on the device: the next pass draws other masks, and those are the stream's next ones too
    assert [int(v) for v in tr.drop_state.cpu()] == [seed, 1]
    for q in ref.parameters():
        q.grad = None
    feed_masks(ref, seed, 1, n, p)
    loss2 = -ref.log_prob(x, context=ctx).mean() if C else -ref.log_prob(x).mean()
    tr.loss_acc.zero_()
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0, context=ctx.to(DEV) if C else None)
    assert abs(tr.loss_acc.item() - loss2.item()) <= 1e-5 * max(1.0, abs(loss2.item()))
    assert abs(loss2.item() - loss_ref.item()) > 1e-6  # (other masks, another loss)
    # evaluation (validation loss, eval mode) has no dropout
    ref.eval()
    with torch.no_grad():
        loss_eval = -ref.log_prob(x, context=ctx).mean() if C else -ref.log_prob(x).mean()
    tr.loss_acc.zero_()
    tr.evaluate(x.to(DEV), context=ctx.to(DEV) if C else None)
    assert abs(tr.loss_acc.item() - loss_eval.item()) <= 1e-5 * max(1.0, abs(loss_eval.item()))


def test_dropout_flow_trains_through_flowmodel_and_through_autograd(tmp_path):
    """``FlowModel.train`` (the epoch loop, HIP graph included: the mask step lives on the device) and the reference's
    own loop through the autograd ops both run a flow with dropout; in eval mode the flow is deterministic and
    inverse / sample work, in training mode they refuse."""
    from nessai_amd.flowmodel import FlowModel

    fm = FlowModel(flow_config=dict(n_inputs=4, n_neurons=16, n_blocks=2, n_layers=2, ftype="nsf", dropout_probability=0.2),
                   training_config=dict(max_epochs=30, patience=30, batch_size=100, device_tag="cuda"),
                   output=str(tmp_path), rng=np.random.default_rng(1))
    fm.initialise()
    data = np.random.default_rng(2).normal(size=(600, 4)) * np.array([0.5, 1.0, 2.0, 0.3])
    history = fm.train(data)
    assert np.isfinite(history["loss"]).all() and history["loss"][-1] < history["loss"][0]
    steps = int(fm.model.trainer().drop_state.cpu()[1])
    assert steps >= 30 * 5  # one tick per training-mode pass, graph replays included
    lp1, lp2 = fm.log_prob(data[:50]), fm.log_prob(data[:50])
    assert np.array_equal(lp1, lp2)  # eval mode: no dropout
    flow = fm.model
    flow.train()
    x = torch.from_numpy(data[:64]).float().to(DEV)
    with torch.enable_grad():
        a = flow.log_prob(x)
        b = flow.log_prob(x)
        assert not torch.equal(a, b)  # training mode: a new mask per pass
        (-a.mean()).backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in flow.parameters())
    with pytest.raises(RuntimeError, match="dropout_probability > 0"):
        flow.inverse(x)
    flow.eval()
    z, _ = flow.forward(x)
    xb, _ = flow.inverse(z)
    assert float((xb - x).abs().max()) < 1e-4


BN_CASES = {
    "nsf_d6": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 2, 2), dict(batch_norm_within_layers=True)),
    "rnvp_default_tanh": (RealNVP, O.RealNVP, (8, 16, 3, 1), dict(batch_norm_within_layers=True, activation=torch.tanh)),
    "nsf_conditional": (NeuralSplineFlow, O.NeuralSplineFlow, (4, 16, 2, 2),
                        dict(batch_norm_within_layers=True, context_features=3)),
    "nsf_with_dropout": (NeuralSplineFlow, O.NeuralSplineFlow, (5, 24, 2, 2),
                         dict(batch_norm_within_layers=True, dropout_probability=0.2)),
}


def bn_pair(name, seed=3):
    P, Oc, args, kw = BN_CASES[name]
    torch.manual_seed(seed)
    ref = Oc(*args, **kw)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n_, q in ref.named_parameters():
            if "linear_layers.1" in n_:
                q.mul_(200.0)
            if "batch_norm_layers" in n_:  # gamma / beta off their trivial values
                q.add_(0.3 * torch.randn(q.shape, generator=g))
        for n_, b in ref.named_buffers():
            if "batch_norm_layers" in n_ and n_.endswith("running_mean"):
                b.add_(0.2 * torch.randn(b.shape, generator=g))
            if "batch_norm_layers" in n_ and n_.endswith("running_var"):
                b.mul_(0.5 + torch.rand(b.shape, generator=g))
    flow = P(*args, **kw)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    return ref, flow, args[0], kw


@pytest.mark.parametrize("name", sorted(BN_CASES))
def test_batch_norm_within_layers_step_matches_autograd(name):
    ref, flow, D, kw = bn_pair(name)
    p, C = kw.get("dropout_probability", 0.0), kw.get("context_features", 0)
    n = 257
    x = torch.randn(n, D) * 1.3
    ctx = torch.randn(n, C) if C else None
    tr = Trainer(flow)
    assert tr.inner_batch_norm
    ref.train()
    if p:
        seed, step = (int(v) for v in tr.drop_state.cpu())
        feed_masks(ref, seed, step, n, p)
    loss_ref = -ref.log_prob(x, context=ctx).mean() if C else -ref.log_prob(x).mean()
    loss_ref.backward()
    g_ref = torch.cat([q.grad.reshape(-1) for q in ref.parameters()])
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0, context=ctx.to(DEV) if C else None)
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    g = tr.grads.cpu()
    scale = max(1e-3, float(g_ref.abs().max()))
    assert float((g - g_ref).abs().max()) <= 2e-4 * scale
    assert rel_err(torch.linalg.norm(g), torch.linalg.norm(g_ref), floor=1e-6) <= 1e-4
    # the running statistics moved as torch's BatchNorm1d moves them (momentum 0.1, unbiased variance), the pass counter ticked
    want, got = dict(ref.named_buffers()), dict(flow.named_buffers())
    for k, b in want.items():
        if "batch_norm_layers" in k:
            if k.endswith("num_batches_tracked"):
                assert int(got[k]) == int(b) == 1
            else:
                np.testing.assert_allclose(got[k].cpu().numpy(), b.numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    # validation pass: running statistics, nothing updated
    ref.eval()
    with torch.no_grad():
        loss_eval = -ref.log_prob(x, context=ctx).mean() if C else -ref.log_prob(x).mean()
    before = {k: v.clone() for k, v in flow.named_buffers() if "batch_norm_layers" in k}
    tr.loss_acc.zero_()
    tr.evaluate(x.to(DEV), context=ctx.to(DEV) if C else None)
    assert abs(tr.loss_acc.item() - loss_eval.item()) <= 1e-5 * max(1.0, abs(loss_eval.item()))
    for k, v in before.items():
        assert torch.equal(dict(flow.named_buffers())[k], v), k
    # ... and the inference kernels (running statistics folded into the image) agree with it
    flow.eval()
    lp = flow.log_prob(x.to(DEV), context=ctx.to(DEV)) if C else flow.log_prob(x.to(DEV))
    assert abs(float(-lp.mean()) - loss_eval.item()) <= 2e-5 * max(1.0, abs(loss_eval.item()))


def test_batch_norm_within_layers_trains_through_flowmodel(tmp_path):
    from nessai_amd.flowmodel import FlowModel

    fm = FlowModel(flow_config=dict(n_inputs=4, n_neurons=16, n_blocks=2, n_layers=2, ftype="realnvp",
                                    batch_norm_within_layers=True),
                   training_config=dict(max_epochs=40, patience=40, batch_size=100, device_tag="cuda"),
                   output=str(tmp_path), rng=np.random.default_rng(1))
    fm.initialise()
    data = np.random.default_rng(2).normal(size=(600, 4)) * np.array([0.5, 1.0, 2.0, 0.3]) + 1.0
    history = fm.train(data)
    assert np.isfinite(history["loss"]).all() and history["loss"][-1] < history["loss"][0]
    x = fm.sample(2000)
    assert np.isfinite(x).all() and abs(float(x[:, 2].std()) - 2.0) < 0.6
    z, lp = fm.forward_and_log_prob(data[:100])
    xb, _ = fm.inverse(z)
    assert float(np.abs(xb - data[:100]).max()) < 1e-3


SVD_CASES = {
    "rnvp_svd": (RealNVP, O.RealNVP, (6, 16, 3, 2), dict(linear_transform="svd")),  # + batch norm between
    "nsf_svd_d9": (NeuralSplineFlow, O.NeuralSplineFlow, (9, 16, 2, 1), dict(linear_transform="svd")),
    "nsf_svd_d70": (NeuralSplineFlow, O.NeuralSplineFlow, (70, 32, 2, 1), dict(linear_transform="svd")),  # two elements per lane
}


@pytest.mark.parametrize("name", sorted(SVD_CASES))
def test_svd_linear_step_matches_autograd(name):
    """``linear_transform="svd"`` (``src/nessai/flows/utils.py:316-323``: two sequences of 10 Householder reflections
    around a positive diagonal) in the training kernels: loss and every gradient - the reflection vectors', the
    diagonal's (its log-determinant term included), the bias' - against torch autograd of the oracle."""
    P, Oc, args, kw = SVD_CASES[name]
    torch.manual_seed(3)
    ref = Oc(*args, **kw)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n_, q in ref.named_parameters():
            if "linear_layers.1" in n_:
                q.mul_(200.0)
            if "q_vectors" in n_:  # off their identity pairs
                q.add_(0.3 * torch.randn(q.shape, generator=g))
            if "unconstrained_diagonal" in n_ or n_.endswith(".bias") and "orthogonal" not in n_:
                q.add_(0.2 * torch.randn(q.shape, generator=g))
    flow = P(*args, **kw)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    D = args[0]
    x = torch.randn(203, D) * 1.2
    ref.train()
    loss_ref = -ref.log_prob(x).mean()
    loss_ref.backward()
    names = [n_ for n_, _ in ref.named_parameters()]
    g_ref = torch.cat([q.grad.reshape(-1) for q in ref.parameters()])
    tr = Trainer(flow)
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0)
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    got = tr.grads.cpu()
    scale = max(1e-3, float(g_ref.abs().max()))
    worst = float((got - g_ref).abs().max())
    if worst > 2e-4 * scale:  # say which parameter
        pos = 0
        for n_, q in zip(names, ref.parameters()):
            e = float((got[pos:pos + q.numel()] - q.grad.reshape(-1)).abs().max())
            if e > 2e-4 * scale:
                print(f"{n_}: {e:.3e} (scale {scale:.3e})")
            pos += q.numel()
    assert worst <= 2e-4 * scale
    assert rel_err(torch.linalg.norm(got), torch.linalg.norm(g_ref), floor=1e-6) <= 1e-4


def test_svd_flow_trains_through_flowmodel(tmp_path):
    from nessai_amd.flowmodel import FlowModel

    fm = FlowModel(flow_config=dict(n_inputs=5, n_neurons=16, n_blocks=2, n_layers=1, ftype="realnvp",
                                    linear_transform="svd"),
                   training_config=dict(max_epochs=40, patience=40, batch_size=100, device_tag="cuda"),
                   output=str(tmp_path), rng=np.random.default_rng(1))
    fm.initialise()
    rng = np.random.default_rng(2)
    base = rng.normal(size=(600, 5))
    data = base @ np.array([[1.0, 0.8, 0, 0, 0], [0, 0.6, 0, 0, 0], [0, 0, 1, 0.5, 0], [0, 0, 0, 0.9, 0], [0, 0, 0, 0, 2.0]])
    history = fm.train(data)
    assert np.isfinite(history["loss"]).all() and history["loss"][-1] < history["loss"][0] - 0.05
    z, lp = fm.forward_and_log_prob(data[:100])
    xb, _ = fm.inverse(z)
    assert float(np.abs(xb - data[:100]).max()) < 1e-3
