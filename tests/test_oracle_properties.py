"""The oracle against the reference's own property tests and analytic known answers.

The reference holds no numeric golden vectors for the coupling / spline
arithmetic (it lives in the absent ``glasflow`` package); what its
test-suite pins are properties, restated here on the oracle, plus analytic
known-answer tests derived from the spline definition.
"""
import numpy as np
import pytest
import scipy.stats
import torch

from oracle import flows as O
from oracle.nflows_restated import splines
from oracle.nflows_restated import transforms as T
from oracle.nflows_restated.distributions import StandardNormal
from oracle.proposal import check_batch_size, compute_radius

FLOWS = [O.RealNVP, O.NeuralSplineFlow]


def make(flow_cls, n_inputs):
    kwargs = dict(batch_norm_between_layers=False) if flow_cls is O.RealNVP else {}
    return flow_cls(n_inputs, 8, 2, 2, **kwargs).eval()  # test_included_flows.py:39-43


@pytest.mark.parametrize("flow_cls", FLOWS)
@pytest.mark.parametrize("n_inputs", [2, 4])
def test_invertibility(flow_cls, n_inputs):
    """tests/test_flows/test_included_flows.py:144-154."""
    flow = make(flow_cls, n_inputs)
    x = torch.randn(1000, n_inputs)
    with torch.no_grad():
        z, log_J = flow.forward(x)
        x_out, log_J_out = flow.inverse(z)
    np.testing.assert_array_almost_equal(x.numpy(), x_out.numpy(), decimal=5)
    np.testing.assert_array_almost_equal(log_J.numpy(), -log_J_out.numpy(), decimal=5)


@pytest.mark.parametrize("flow_cls", FLOWS)
def test_forward_and_log_prob(flow_cls):
    """tests/test_flows/test_included_flows.py:114-126 (exact equality)."""
    flow = make(flow_cls, 4)
    x = torch.randn(1000, 4)
    with torch.no_grad():
        z, log_prob = flow.forward_and_log_prob(x)
        z_target, _ = flow.forward(x)
        log_prob_target = flow.log_prob(x)
    np.testing.assert_equal(z.numpy(), z_target.numpy())
    np.testing.assert_equal(log_prob.numpy(), log_prob_target.numpy())


@pytest.mark.parametrize("flow_cls", FLOWS)
def test_sample_and_log_prob(flow_cls):
    """tests/test_flows/test_included_flows.py:129-141."""
    flow = make(flow_cls, 4)
    with torch.no_grad():
        x, log_prob = flow.sample_and_log_prob(1000)
        log_prob_target = flow.log_prob(x)
    np.testing.assert_array_almost_equal(log_prob.numpy(), log_prob_target.numpy(), decimal=5)


def test_standard_normal_vs_scipy():
    """Gaussian base log-prob against scipy (style of
    tests/test_flows/test_distributions/test_multivariate_normal.py:34-38)."""
    dist = StandardNormal([3])
    x = torch.randn(100, 3)
    expected = scipy.stats.multivariate_normal(mean=np.zeros(3), cov=np.eye(3)).logpdf(x.numpy())
    np.testing.assert_array_almost_equal(dist.log_prob(x).numpy(), expected, decimal=5)


def test_lu_cache_reset():
    """Forward identical after a train()/eval() toggle
    (tests/test_flowmodel/test_flowmodel_base.py:751-788)."""
    flow = O.RealNVP(4, 8, 2, 2).eval()
    x = torch.randn(50, 4)
    with torch.no_grad():
        z1, _ = flow.forward(x)
        lu = [m for m in flow.modules() if isinstance(m, T.LULinear)][0]
        lu.lower_entries.add_(0.2)
        z_stale, _ = flow.forward(x)  # cache still holds the old weight
        flow.train()
        flow.eval()
        z2, _ = flow.forward(x)
    assert torch.equal(z1, z_stale)
    assert not torch.equal(z1, z2)


def test_batch_norm_constants():
    """BatchNorm reset constants (tests/test_flows/test_flow_utils.py:158-173)."""
    bn = T.BatchNorm(3)
    const = np.log(np.exp(1 - bn.eps) - 1)
    np.testing.assert_allclose(bn.unconstrained_weight.detach().numpy(), const, rtol=1e-6)
    assert torch.equal(bn.running_mean, torch.zeros(3)) and torch.equal(bn.running_var, torch.ones(3))
    np.testing.assert_allclose(bn.weight.detach().numpy(), 1.0, rtol=1e-6)


def test_default_scale_activation():
    """scale = sigmoid(x + 2) + 1e-3 (tests/test_flows/test_specific_flows.py:32)."""
    x = torch.tensor([-3.0, 0.0, 2.0])
    s = T.AffineCouplingTransform.DEFAULT_SCALE_ACTIVATION(x)
    np.testing.assert_allclose(s.numpy(), 1 / (1 + np.exp(-(x.numpy() + 2))) + 1e-3, rtol=1e-6)


def test_constant_volume_radius():
    """chi.ppf(q, D): 2-D q=0.8647 -> 2.000 (test_flowproposal_integration.py:149-181)."""
    assert round(float(compute_radius(2, 0.8647)), 3) == 2.000
    for d, r in {16: 5.1280, 32: 6.7966, 64: 9.1474, 15: 4.9996, 2: 2.4477}.items():
        assert abs(compute_radius(d, 0.95) - r) < 1e-3


def test_check_batch_size_table():
    """flowmodel/base.py:195-236 (reference tests/test_flowmodel/test_flowmodel_base.py:145-183)."""
    assert check_batch_size(np.zeros(1800), 1000) == 1000  # 800 left >= 100
    assert check_batch_size(np.zeros(1005), 1000) == 905  # 1005 % 905 = 100 >= 100
    for data_size, batch_size in [(4010, 1000), (106, 21), (1000, 1000), (2000, 1000)]:
        out = check_batch_size(np.arange(data_size), batch_size)
        assert out >= int(0.1 * batch_size)
        if not data_size % batch_size:
            assert out == batch_size
    # minimum valid size 8 would leave a final batch of 1 -> 7
    assert check_batch_size(np.arange(25), 10, min_fraction=0.8) == 7
    with pytest.raises(RuntimeError, match="Could not find a valid batch size"):
        check_batch_size(np.arange(3), 2, min_fraction=1.0)
    with pytest.raises(ValueError, match="Cannot use a batch size of 1"):
        check_batch_size(np.arange(2), 1, min_fraction=1.0)


# -- analytic known answers for the rational-quadratic spline ---------------------


def test_spline_identity_tails():
    x = torch.tensor([-7.0, -5.0001, 5.0001, 9.0])
    p = torch.randn(4, 8), torch.randn(4, 8), torch.randn(4, 7)
    for inverse in (False, True):
        y, ld = splines.unconstrained_rational_quadratic_spline(
            x, *p, inverse=inverse, tails="linear", tail_bound=5.0)
        assert torch.equal(y, x) and torch.equal(ld, torch.zeros(4))


def test_spline_zero_parameters():
    """All-zero parameters: equal bins; knots map to knots; interior knot
    derivative 1e-3 + ln 2; boundary derivative exactly (to f32) 1."""
    K = 8
    knots = torch.linspace(-5, 5, K + 1)[1:-1]
    zeros = torch.zeros(len(knots), K)
    y, _ = splines.unconstrained_rational_quadratic_spline(
        knots, zeros, zeros, torch.zeros(len(knots), K - 1), tails="linear", tail_bound=5.0)
    np.testing.assert_allclose(y.numpy(), knots.numpy(), atol=2e-6)
    # derivative at an interior knot and just inside the boundary
    eps = 1e-3
    x = torch.tensor([knots[2].item() + eps, -5.0 + eps])
    z = torch.zeros(2, K)
    _, ld = splines.unconstrained_rational_quadratic_spline(
        x, z, z, torch.zeros(2, K - 1), tails="linear", tail_bound=5.0)
    assert abs(np.exp(ld[0].item()) - (1e-3 + np.log(2))) < 2e-3
    assert abs(np.exp(ld[1].item()) - 1.0) < 2e-3


def test_spline_monotone_and_inverse():
    torch.manual_seed(0)
    n = 400
    x = torch.linspace(-4.999, 4.999, n)
    uw, uh, ud = (torch.randn(1, 8).repeat(n, 1), torch.randn(1, 8).repeat(n, 1),
                  torch.randn(1, 7).repeat(n, 1))
    y, ld = splines.unconstrained_rational_quadratic_spline(x, uw, uh, ud, tails="linear", tail_bound=5.0)
    assert (y[1:] > y[:-1]).all()
    xr, ldi = splines.unconstrained_rational_quadratic_spline(
        y, uw, uh, ud, inverse=True, tails="linear", tail_bound=5.0)
    np.testing.assert_allclose(xr.numpy(), x.numpy(), atol=2e-5)
    np.testing.assert_allclose(ldi.numpy(), -ld.numpy(), atol=2e-5)
    # log-det equals the numerical derivative (float64, central difference)
    xd, h = x.double(), 1e-6
    p64 = (uw.double(), uh.double(), ud.double())
    yp, _ = splines.unconstrained_rational_quadratic_spline(xd + h, *p64, tails="linear", tail_bound=5.0)
    ym, _ = splines.unconstrained_rational_quadratic_spline(xd - h, *p64, tails="linear", tail_bound=5.0)
    _, ld64 = splines.unconstrained_rational_quadratic_spline(xd, *p64, tails="linear", tail_bound=5.0)
    num = torch.log((yp - ym) / (2 * h))
    np.testing.assert_allclose(ld64.numpy(), num.numpy(), atol=1e-4)
    np.testing.assert_allclose(ld.numpy(), ld64.numpy(), atol=1e-4)


def test_residual_net_init_scale():
    """Last linear layer of every residual block starts in U(-1e-3, 1e-3)."""
    flow = O.NeuralSplineFlow(4, 8, 2, 2)
    for name, p in flow.named_parameters():
        if "linear_layers.1" in name:
            assert p.abs().max().item() <= 1e-3


@pytest.mark.parametrize("K", [4, 8, 10, 16])
def test_spline_matches_the_independent_huggingface_port_of_nflows(K):
    """The restated rational-quadratic spline against golden vectors produced by an independent
    third-party port of the nflows spline functions (HuggingFace transformers' VITS model,
    ``tests/golden/make_spline_goldens.py``): forward and inverse, inputs inside, on and beyond the
    tail bound - bit for bit."""
    import os

    from oracle.nflows_restated import splines

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spline_hf_goldens.npz"))
    x, uw, uh, ud = (torch.from_numpy(g[f"k{K}/{n}"]) for n in ("x", "uw", "uh", "ud"))
    for tag, inverse in (("fwd", False), ("inv", True)):
        y, lad = splines.unconstrained_rational_quadratic_spline(
            x, uw, uh, ud, inverse=inverse, tails="linear", tail_bound=5.0, min_bin_width=1e-3,
            min_bin_height=1e-3, min_derivative=1e-3)
        assert np.array_equal(y.numpy(), g[f"k{K}/{tag}/y"])
        assert np.array_equal(lad.numpy(), g[f"k{K}/{tag}/logabsdet"])


def test_spline_matches_huggingface_port_live():
    """Same comparison against the live function when transformers is importable."""
    V = pytest.importorskip("transformers.models.vits.modeling_vits")
    from oracle.nflows_restated import splines

    gen = torch.Generator().manual_seed(9)
    x = torch.randn(3000, generator=gen) * 2.5
    uw, uh, ud = torch.randn(3000, 8, generator=gen), torch.randn(3000, 8, generator=gen), torch.randn(3000, 7, generator=gen)
    for inverse in (False, True):
        a, la = V._unconstrained_rational_quadratic_spline(x, uw, uh, ud, reverse=inverse, tail_bound=5.0,
                                                           min_bin_width=1e-3, min_bin_height=1e-3,
                                                           min_derivative=1e-3)
        b, lb = splines.unconstrained_rational_quadratic_spline(x, uw, uh, ud, inverse=inverse, tails="linear",
                                                                tail_bound=5.0)
        assert torch.equal(a, b) and torch.equal(la, lb)
