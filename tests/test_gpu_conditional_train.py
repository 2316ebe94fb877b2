"""Training of conditional flows (``FlowModel.train(conditional=...)``, ``loss = -model.log_prob(x, c).mean()``:
``src/nessai/flowmodel/base.py:286-302, 404-436``) on the gfx950 training kernels (runs on a real MI355X; this module
runs with autograd on: ``tests/conftest.py``)."""
import numpy as np
import pytest
import torch

from helpers import DEV
from nessai_amd.flows import NeuralSplineFlow, RealNVP
from oracle import flows as O

pytestmark = pytest.mark.gpu


# ---- training (the layer-wise kernels: nsa_train.hip, conditional branch) ---------------------------------------------

TRAIN_CASES = {
    "nsf_c2": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 2, 2), dict(context_features=2)),
    "rnvp_lu_bn_c1": (RealNVP, O.RealNVP, (5, 24, 3, 1), dict(context_features=1)),
    "nsf_k5_c7_tanh": (NeuralSplineFlow, O.NeuralSplineFlow, (7, 40, 2, 3),
                       dict(context_features=7, num_bins=5, activation=torch.tanh)),
}


def _train_pair(name, seed=3):
    P, Oc, args, kw = TRAIN_CASES[name]
    torch.manual_seed(seed)
    ref = Oc(*args, **kw)
    with torch.no_grad():
        for n_, p in ref.named_parameters():
            if "linear_layers.1" in n_:
                p.mul_(200.0)
        g = torch.Generator().manual_seed(seed)
        for n_, p in ref.named_parameters():
            if "entries" in n_ or "unconstrained" in n_:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    flow = P(*args, **kw)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    return ref, flow, args[0], kw["context_features"]


@pytest.mark.parametrize("name", sorted(TRAIN_CASES))
def test_conditional_loss_and_gradients_match_autograd(name):
    """One minibatch of ``-mean(log_prob(x, context))`` (flowmodel/base.py:411-412): loss and every gradient - the
    context layers' included - against torch autograd of the oracle."""
    from nessai_amd.train import Trainer

    ref, flow, D, C = _train_pair(name)
    x = torch.randn(333, D) * 1.5
    c = torch.randn(333, C)
    ref.train()
    loss_ref = -ref.log_prob(x, context=c).mean()
    loss_ref.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    assert float(torch.cat([p.grad.reshape(-1) for n_, p in ref.named_parameters() if "context_layer" in n_]).abs().max()) > 1e-4
    tr = Trainer(flow)
    before = tr.params.clone()
    tr.step(x.to(DEV), clip_norm=5.0, lr=0.0, context=c.to(DEV))
    torch.cuda.synchronize()
    assert abs(tr.loss_acc.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    g = tr.grads.cpu()
    scale = max(1e-3, float(g_ref.abs().max()))
    assert float((g - g_ref).abs().max()) <= 2e-4 * scale
    assert torch.equal(tr.params, before)


def test_reference_training_loop_on_a_conditional_flow():
    """The reference's own loop - ``loss = -model.log_prob(x, c).mean(); loss.backward(); optimiser.step()``
    (flowmodel/base.py:400-444) - drives the HIP flow through autograd with a context; three AdamW steps against
    torch on the oracle."""
    ref, flow, D, C = _train_pair("nsf_c2")
    ref.train()
    flow.train()
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    opt = torch.optim.AdamW(flow.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        x, c = torch.randn(200, D, generator=g), torch.randn(200, C, generator=g)
        opt_ref.zero_grad()
        loss_ref = -ref.log_prob(x, context=c).mean()
        loss_ref.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
        opt_ref.step()
        opt.zero_grad()
        loss = -flow.log_prob(x.to(DEV), context=c.to(DEV)).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(flow.parameters(), 5.0)
        opt.step()
        assert abs(loss.item() - loss_ref.item()) <= 2e-5 * max(1.0, abs(loss_ref.item()))
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), flow.named_parameters()):
        assert n1 == n2 and float((p1 - p2.cpu()).abs().max()) <= 2e-4 * max(1.0, float(p1.abs().max())), n1


def test_flowmodel_train_conditional_learns_the_dependence(tmp_path):
    """``FlowModel.train(samples, conditional=...)`` (flowmodel/base.py:286-302, 610-659): x ~ N(3 c, 1.5^2) per
    feature (a spread above 1: the affine coupling's forward scale, sigmoid(.) + 1e-3, can only shrink); after the
    fit, samples drawn under a context follow it and the loss has dropped."""
    from nessai_amd.flowmodel import FlowModel

    g = np.random.default_rng(2)
    n = 2000
    c = g.uniform(-1.0, 1.0, size=(n, 1))
    x = 3.0 * c + 1.5 * g.normal(size=(n, 2))
    fm = FlowModel(flow_config=dict(n_inputs=2, n_neurons=16, n_blocks=2, n_layers=2, ftype="realnvp",
                                    context_features=1, batch_norm_between_layers=False, linear_transform=None),
                   training_config=dict(max_epochs=400, patience=400, lr=1e-2, batch_size=500, val_size=0.1),
                   output=str(tmp_path), rng=np.random.default_rng(1))
    fm.initialise()
    history = fm.train(x, conditional=c)
    assert history["loss"][-1] < history["loss"][0] - 0.3
    with pytest.raises(RuntimeError, match="weights and conditional"):
        fm.train(x, conditional=c, weights=np.ones(n))
    for value in (-0.8, 0.7):
        cc = np.full((4000, 1), value)
        s = fm.sample(4000, conditional=cc)
        assert np.all(np.abs(s.mean(axis=0) - 3.0 * value) < 0.4), (value, s.mean(axis=0))
        assert np.all(np.abs(s.std(axis=0) - 1.5) < 0.35)  # (the marginal over the contexts has 2.3)
    # log_prob under the right context beats the wrong one
    lp_right = fm.log_prob(x[:500], conditional=c[:500]).mean()
    lp_wrong = fm.log_prob(x[:500], conditional=-c[:500]).mean()
    assert lp_right > lp_wrong + 0.5
