"""B1 under autograd: the reference's own training loop drives the HIP flow classes.

``FlowModel._train`` (``src/nessai/flowmodel/base.py:400-444``) does, per minibatch::

    optimiser.zero_grad()
    loss = -model.log_prob(x).mean()          # or the weighted -sum(w lp) / sum(w), base.py:404-407
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm)
    optimiser.step()

The tests below run exactly that on ``nessai_amd.flows.*`` (``torch.ops.nessai_amd.flow_forward``
/ ``flow_backward`` behind ``torch.autograd``) and on the CPU oracle with identical weights and
batches, and compare losses, gradients and the updated weights.
"""
import numpy as np
import pytest
import torch

from helpers import DEV, rel_err
from test_gpu_train import CASES, flat_of, pair

pytestmark = pytest.mark.gpu


def reference_training_step(model, optimiser, x, weights=None, clip=5.0):
    """One minibatch of the reference's ``_train`` loop, verbatim in structure."""
    optimiser.zero_grad()
    if weights is not None:
        loss = -torch.sum(model.log_prob(x) * weights) / torch.sum(weights)
    else:
        loss = -model.log_prob(x).mean()
    loss.backward()
    if clip:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    optimiser.step()
    return loss.item()


def grads_of(model):
    return torch.cat([p.grad.detach().reshape(-1).cpu() for p in model.parameters()])


@pytest.mark.parametrize("name", sorted(CASES))
def test_loss_backward_matches_the_oracle(name):
    """`loss = -flow.log_prob(x).mean(); loss.backward()` on the HIP flow: loss and every
    parameter gradient against torch autograd of the oracle (tolerances of
    test_gpu_train.py::test_loss_and_gradients_match_autograd)."""
    ref, flow, D = pair(name)
    x = torch.randn(333, D) * 1.5
    ref.train()
    flow.train()
    loss_ref = -ref.log_prob(x).mean()
    loss_ref.backward()
    lp = flow.log_prob(x.to(DEV))
    assert lp.requires_grad and lp.shape == (333,)
    loss = -lp.mean()
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    g, g_ref = grads_of(flow), grads_of(ref)
    scale = max(1e-3, float(g_ref.abs().max()))
    assert float((g - g_ref).abs().max()) <= 2e-4 * scale
    assert rel_err(torch.linalg.norm(g), torch.linalg.norm(g_ref), floor=1e-6) <= 1e-4
    for p in flow.parameters():
        assert p.grad is not None and p.grad.shape == p.shape


@pytest.mark.parametrize("name,opt_cls,opt_kw", [
    ("nsf_c3", torch.optim.AdamW, dict(lr=1e-3)),
    ("nsf_d4", torch.optim.AdamW, dict(lr=1e-2)),
    ("rnvp_c2_default", torch.optim.AdamW, dict(lr=1e-3)),  # LU-linear + training-mode batch norm
    ("rnvp_c2_mlp", torch.optim.Adam, dict(lr=1e-3, weight_decay=1e-6)),
    ("rnvp_perm", torch.optim.SGD, dict(lr=1e-2)),
])
def test_reference_training_loop_drives_the_hip_flow(name, opt_cls, opt_kw):
    """Three optimiser steps of the reference loop with a stock torch optimiser on
    ``flow.parameters()``: same losses and same weights as the oracle afterwards."""
    ref, flow, D = pair(name)
    ref.train()
    flow.train()
    opt_ref = opt_cls(ref.parameters(), **opt_kw)
    opt = opt_cls(flow.parameters(), **opt_kw)
    g = torch.Generator().manual_seed(4)
    for _ in range(3):
        x = torch.randn(200, D, generator=g)
        l_ref = reference_training_step(ref, opt_ref, x)
        l = reference_training_step(flow, opt, x.to(DEV))
        assert abs(l - l_ref) <= 2e-5 * max(1.0, abs(l_ref))
    got = torch.cat([p.detach().reshape(-1).cpu() for p in flow.parameters()])
    diff = (got - flat_of(ref)).abs()
    if opt_cls is torch.optim.SGD:
        assert float(diff.max()) <= 3e-5
    else:
        # Adam's step is lr * m / (sqrt(v) + eps): a weight whose gradient is rounding noise (a
        # shift that the following batch norm removes again has an analytically ZERO gradient)
        # moves by +-lr per step in either implementation, so the worst weight is only bounded by
        # Adam's own step size; the tight bound is on the weights with a real gradient.
        assert float(diff.max()) <= 2.0 * 3 * opt_kw["lr"]
        g_ref = grads_of(ref).abs()
        real = g_ref > 1e-4 * g_ref.max()
        assert float(real.float().mean()) > 0.5
        assert float((diff[real] > 3e-5).float().mean()) <= 5e-3
        assert float(diff[real].mean()) <= 3e-6
    # running statistics of batch norm followed the same three batches
    ref_stats = {k: v for k, v in ref.state_dict().items() if "running" in k}
    for k, v in flow.state_dict().items():
        if "running" in k:  # (the weights in front of them differ by Adam's noise, see above)
            assert torch.allclose(v.cpu(), ref_stats[k], rtol=1e-3, atol=5e-4 if opt_cls is not torch.optim.SGD else 1e-5), k
    # ... and the inference kernel sees the updated weights (eval mode, no grad)
    ref.load_state_dict({k: v.cpu() for k, v in flow.state_dict().items()})
    flow.eval()
    ref.eval()
    xe = torch.randn(64, D, generator=g)
    with torch.no_grad():
        assert rel_err(flow.log_prob(xe.to(DEV)).cpu(), ref.log_prob(xe)) <= 1e-5


def test_weighted_loss_of_the_importance_sampler():
    ref, flow, D = pair("nsf_d4")
    ref.train()
    flow.train()
    x = torch.randn(200, D)
    w = torch.rand(200) + 0.1
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    opt = torch.optim.AdamW(flow.parameters(), lr=1e-3)
    l_ref = reference_training_step(ref, opt_ref, x, weights=w)
    l = reference_training_step(flow, opt, x.to(DEV), weights=w.to(DEV))
    assert abs(l - l_ref) <= 1e-5 * max(1.0, abs(l_ref))
    assert float((grads_of(flow) - grads_of(ref)).abs().max()) <= 2e-4 * max(1e-3, float(grads_of(ref).abs().max()))


@pytest.mark.parametrize("name", ["nsf_d4", "nsf_c3", "rnvp_lu_only", "rnvp_mlp_deep_tanh"])
def test_forward_is_differentiable_in_inputs_and_parameters(name):
    """`z, logabsdet = flow.forward(x)` with upstream gradients on BOTH outputs, and the gradient
    with respect to the input rows."""
    ref, flow, D = pair(name)
    ref.train()
    flow.train()
    x = torch.randn(150, D) * 1.2
    a = torch.randn(150, D)
    b = torch.randn(150)
    x_ref = x.clone().requires_grad_(True)
    z_ref, ld_ref = ref.forward(x_ref)
    (torch.sum(z_ref * a) + torch.sum(ld_ref * b)).backward()
    x_dev = x.to(DEV).requires_grad_(True)
    z, ld = flow.forward(x_dev)
    assert float((z.detach().cpu() - z_ref.detach()).abs().max()) <= 2e-4
    assert float((ld.detach().cpu() - ld_ref.detach()).abs().max()) <= 2e-4
    (torch.sum(z * a.to(DEV)) + torch.sum(ld * b.to(DEV))).backward()
    g, g_ref = grads_of(flow), grads_of(ref)
    assert float((g - g_ref).abs().max()) <= 2e-4 * max(1e-3, float(g_ref.abs().max()))
    gx, gx_ref = x_dev.grad.cpu(), x_ref.grad
    assert float((gx - gx_ref).abs().max()) <= 2e-4 * max(1e-3, float(gx_ref.abs().max()))


def test_forward_and_log_prob_and_input_gradient():
    ref, flow, D = pair("nsf_d7_k4")
    ref.train()
    flow.train()
    x = torch.randn(90, D)
    x_ref = x.clone().requires_grad_(True)
    z_ref, lp_ref = ref.forward_and_log_prob(x_ref)
    (lp_ref.sum() + 0.5 * (z_ref ** 2).sum()).backward()
    x_dev = x.to(DEV).requires_grad_(True)
    z, lp = flow.forward_and_log_prob(x_dev)
    (lp.sum() + 0.5 * (z ** 2).sum()).backward()
    assert rel_err(lp.detach().cpu(), lp_ref.detach()) <= 1e-5
    assert float((x_dev.grad.cpu() - x_ref.grad).abs().max()) <= 2e-4 * max(1e-3, float(x_ref.grad.abs().max()))
    assert float((grads_of(flow) - grads_of(ref)).abs().max()) <= 2e-4 * max(1e-3, float(grads_of(ref).abs().max()))


def test_two_graphs_alive_at_once_and_gradient_accumulation():
    """Each differentiable call owns its workspace: two forwards, then the two backwards in the
    other order, accumulate into ``.grad`` exactly like the oracle."""
    ref, flow, D = pair("nsf_d6_tanh")
    ref.train()
    flow.train()
    x1, x2 = torch.randn(70, D), torch.randn(45, D) * 2.0
    l1r, l2r = -ref.log_prob(x1).mean(), -ref.log_prob(x2).sum()
    l2r.backward()
    l1r.backward()
    l1, l2 = -flow.log_prob(x1.to(DEV)).mean(), -flow.log_prob(x2.to(DEV)).sum()
    l2.backward()
    l1.backward()
    g, g_ref = grads_of(flow), grads_of(ref)
    assert float((g - g_ref).abs().max()) <= 2e-4 * max(1e-3, float(g_ref.abs().max()))
    with pytest.raises(RuntimeError):
        l1.backward()  # the graph (and its workspace) was consumed


def test_backward_refuses_stale_parameters():
    """The backward kernels re-read the live parameter buffer: a parameter changed in place between a
    forward pass and its backward (an optimiser step; a second training-mode forward of a batch-norm flow)
    raises instead of giving gradients against the new values (torch's own version check for saved tensors)."""
    ref, flow, D = pair("nsf_d4")
    flow.train()
    x = torch.randn(40, D, device=DEV)
    loss = -flow.log_prob(x).mean()
    with torch.no_grad():
        next(flow.parameters()).mul_(1.001)
    with pytest.raises(RuntimeError, match="modified in place"):
        loss.backward()
    ref, flow, D = pair("rnvp_c2_default")  # batch norm: a training-mode forward rewrites the running statistics
    flow.train()
    x = torch.randn(64, D, device=DEV)
    loss = -flow.log_prob(x).mean()
    flow.log_prob(x)
    with pytest.raises(RuntimeError, match="modified in place"):
        loss.backward()


def test_frozen_transform_and_no_grad_use_the_inference_kernel():
    ref, flow, D = pair("nsf_d4")
    x = torch.randn(50, D, device=DEV)
    flow.eval()
    with torch.no_grad():
        assert not flow.log_prob(x).requires_grad
    with torch.inference_mode():
        assert not flow.log_prob(x).requires_grad
    flow.freeze_transform()
    assert not flow.log_prob(x).requires_grad  # nothing requires a gradient
    flow.unfreeze_transform()
    assert flow.log_prob(x).requires_grad
    # inverse / sample are evaluated without a graph
    xs, ld = flow.inverse(torch.randn(20, D, device=DEV))
    assert not xs.requires_grad and not ld.requires_grad


def test_training_mode_batch_norm_uses_batch_statistics():
    """Training mode without autograd (as nflows' BatchNorm does): batch statistics, running
    statistics updated; eval mode differentiates through the running statistics."""
    ref, flow, D = pair("rnvp_c2_default")
    x = torch.randn(300, D) * 1.3 + 0.2
    ref.train()
    flow.train()
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
        lp = flow.log_prob(x.to(DEV))
    assert rel_err(lp.cpu(), lp_ref) <= 1e-5
    ref_stats = {k: v for k, v in ref.state_dict().items() if "running" in k}
    for k, v in flow.state_dict().items():
        if "running" in k:
            assert torch.allclose(v.cpu(), ref_stats[k], rtol=1e-4, atol=1e-5), k
    with pytest.raises(RuntimeError):
        flow.inverse(x.to(DEV))  # inverse needs running statistics: eval mode only
    # eval mode: running statistics, with and without a graph
    flow.eval()
    ref.eval()
    with torch.no_grad():
        assert rel_err(flow.log_prob(x.to(DEV)).cpu(), ref.log_prob(x)) <= 1e-5
    ref.train()
    ref.eval()  # drop the LU cache filled under no_grad (flowmodel/base.py:681-682 does the same)
    lp_ref = ref.log_prob(x)
    (-lp_ref.mean()).backward()
    lp = flow.log_prob(x.to(DEV))
    assert lp.requires_grad and rel_err(lp.detach().cpu(), lp_ref.detach()) <= 1e-5
    (-lp.mean()).backward()
    g, g_ref = grads_of(flow), grads_of(ref)
    assert float((g - g_ref).abs().max()) <= 2e-4 * max(1e-3, float(g_ref.abs().max()))


def test_operators_are_registered_with_the_dispatcher():
    """`torch.ops.nessai_amd.*`: schemas, the ROCm kernels and the Meta shape functions."""
    ref, flow, D = pair("nsf_c3")
    flow.eval()
    for name in ("flow_run", "flow_forward", "flow_backward"):
        op = getattr(torch.ops.nessai_amd, name)
        assert op.default._schema.name == f"nessai_amd::{name}"
    x = torch.randn(257, D, device=DEV)
    header, image = flow.device_image()
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    z, lp = torch.ops.nessai_amd.flow_run(x, image, flow._header_t, 0, 1, True, status, 0, 0)
    with torch.no_grad():
        assert torch.equal(lp, flow.log_prob(x))
        assert rel_err(lp.cpu(), ref.eval().log_prob(x.cpu())) <= 1e-5
    # shape inference without a kernel launch
    xm = torch.empty(257, D, device="meta")
    zm, lpm = torch.ops.nessai_amd.flow_run(xm, torch.empty(image.shape, dtype=image.dtype, device="meta"),
                                            flow._header_t, 0, 1, True,
                                            torch.empty(4, dtype=torch.int32, device="meta"), 0, 0)
    assert zm.shape == (257, D) and lpm.shape == (257,)
    tr = flow.trainer()
    zf, ldf, lpf, ws = torch.ops.nessai_amd.flow_forward(x, tr.flat, tr.handle, False, True, True, True)
    assert rel_err(lpf.cpu(), lp.cpu()) <= 1e-5 and zf.shape == (257, D) and ldf.shape == (257,)
    grads, gx = torch.ops.nessai_amd.flow_backward(ws, tr.flat, tr.handle, 257, False, None, None,
                                                   torch.full((257,), -1.0 / 257, device=DEV), True)
    assert grads.shape == (tr.n_params,) and gx.shape == (257, D)
    assert np.isfinite(grads.cpu().numpy()).all()


def test_flowmodel_and_autograd_share_one_parameter_buffer(tmp_path):
    """The fused trainer of nessai_amd.FlowModel and the autograd path work on the same flat
    buffer: a fused fit followed by reference-loop steps keeps `state_dict` consistent."""
    from nessai_amd.flowmodel import FlowModel

    rng = np.random.default_rng(0)
    data = rng.normal(size=(400, 4))
    fm = FlowModel(flow_config=dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=1, ftype="nsf"),
                   training_config=dict(batch_size=200, max_epochs=3, patience=5), output=str(tmp_path), rng=rng)
    fm.initialise()
    fm.train(data)
    flow = fm.model
    assert flow.trainer() is fm._trainer
    before = flow.state_dict()["_transform._transforms.1.transform_net.initial_layer.weight"].clone()
    flow.train()
    opt = torch.optim.AdamW(flow.parameters(), lr=1e-2)
    reference_training_step(flow, opt, fm.numpy_array_to_tensor(data[:100]))
    after = flow.state_dict()["_transform._transforms.1.transform_net.initial_layer.weight"]
    assert not torch.equal(before, after)
    assert flow.trainer() is fm._trainer  # still the same buffer
    lp = fm.log_prob(data[:50])  # eval + inference kernel on the stepped weights
    tr = flow.trainer()
    out = torch.empty(50, dtype=torch.float32, device=DEV)
    tr.evaluate(fm.numpy_array_to_tensor(data[:50]), log_prob_out=out)
    assert rel_err(lp, out.cpu().numpy()) <= 1e-5
