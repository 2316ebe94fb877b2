"""Parity of the gfx950 flow kernels against the CPU oracle (runs on a real MI355X).

Bar (BASELINE.json north_star): ``flow.log_prob`` within 1e-5 relative of the
reference CPU path on identical weights and inputs; the reference's own
property tests for the flows (``tests/test_flows/test_included_flows.py``)
are restated on the device path.
"""
import copy
import os

import numpy as np
import pytest
import torch

from helpers import DEV, make_pair, rel_err
from nessai_amd.flows import NeuralSplineFlow, RealNVP, configure_model
from oracle import flows as O

pytestmark = pytest.mark.gpu

LOG_PROB_RTOL = 1e-5  # north_star tolerance, relative
X_ATOL = 2e-4  # float32 coupling stack, |x| <= ~15
LOGDET_ATOL = 5e-4  # the inverse amplifies float32 round-off of x into its log-det

CONFIGS = {
    "nsf_d2": (NeuralSplineFlow, O.NeuralSplineFlow, (2, 8, 2, 2), {}),
    "nsf_d4": (NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2), {}),
    "nsf_d5_k10_bn": (NeuralSplineFlow, O.NeuralSplineFlow, (5, 40, 2, 1),
                      dict(num_bins=10, batch_norm_between_layers=True)),
    "nsf_c3": (NeuralSplineFlow, O.NeuralSplineFlow, (32, 64, 6, 2), {}),
    "nsf_c5_d15": (NeuralSplineFlow, O.NeuralSplineFlow, (15, 30, 4, 2), {}),
    "nsf_c4_d64": (NeuralSplineFlow, O.NeuralSplineFlow, (64, 128, 3, 2), {}),
    "nsf_k4_lu_tanh": (NeuralSplineFlow, O.NeuralSplineFlow, (8, 16, 2, 2),
                       dict(num_bins=4, linear_transform="lu", activation=torch.tanh)),
    "nsf_k16": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 12, 2, 2), dict(num_bins=16)),
    # bin counts without an exact kernel instance (the reference passes num_bins through, flows/nsf.py:66,103): the
    # generic spline instance on parameter sets padded to 16 bins
    "nsf_k5": (NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 2, 2), dict(num_bins=5)),
    "nsf_k2": (NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2), dict(num_bins=2)),
    "nsf_k13_d32": (NeuralSplineFlow, O.NeuralSplineFlow, (32, 64, 2, 2), dict(num_bins=13)),
    # linear_transform="svd" (flows/utils.py:316-323): two Householder sequences and a diagonal, one dense op in eval mode
    "rnvp_svd": (RealNVP, O.RealNVP, (6, 16, 3, 2), dict(linear_transform="svd")),
    "nsf_svd": (NeuralSplineFlow, O.NeuralSplineFlow, (8, 16, 2, 2), dict(linear_transform="svd")),
    "rnvp_c2": (RealNVP, O.RealNVP, (16, 32, 4, 2), {}),
    "rnvp_c2_mlp": (RealNVP, O.RealNVP, (16, 32, 4, 2), dict(net="mlp")),
    "rnvp_d2": (RealNVP, O.RealNVP, (2, 8, 2, 2), dict(batch_norm_between_layers=False)),
    "rnvp_additive_actnorm": (RealNVP, O.RealNVP, (6, 16, 2, 2),
                              dict(use_volume_preserving=True, actnorm=True,
                                   batch_norm_between_layers=False, linear_transform=None)),
    "rnvp_silu_perm": (RealNVP, O.RealNVP, (7, 33, 3, 1),
                       dict(linear_transform="permutation", activation=torch.nn.functional.silu)),
    "rnvp_pre_bn": (RealNVP, O.RealNVP, (4, 8, 2, 2), dict(pre_transform="batch_norm")),
    # a user scale_activation of the sigmoid(x + a) + b family (tests/test_flows/test_specific_flows.py:32
    # of the reference passes the default as a lambda); a, b are recovered by probing the callable
    "rnvp_custom_scale": (RealNVP, O.RealNVP, (6, 16, 3, 2),
                          dict(linear_transform="permutation", batch_norm_between_layers=False,
                               scale_activation=lambda x: torch.sigmoid(x - 0.5) + 0.05)),
    # the reference's defaults beyond 64 inputs: n_neurons = 2 * n_inputs (src/nessai/flows/utils.py:148-149) and,
    # for RealNVP, linear_transform="lu" (flows/realnvp.py:90) - 192- / 200-wide conditioners on the
    # eight-hidden-tiles instances, an LU-linear layer over 100 features as four output tiles
    "nsf_d96_h192": (NeuralSplineFlow, O.NeuralSplineFlow, (96, 192, 2, 1), {}),
    "rnvp_d100_lu": (RealNVP, O.RealNVP, (100, 200, 2, 1), {}),
}


@pytest.fixture(params=[(name, prec) for name in sorted(CONFIGS) for prec in ("bf16x3", "f16x3", "f32")],
                ids=lambda p: f"{p[0]}-{p[1]}")
def pair(request):
    name, precision = request.param
    P, Oc, args, kw = CONFIGS[name]
    ref, flow = make_pair(P, Oc, args, kw)
    flow.precision = precision
    return ref, flow, args[0]


# split-bf16: validated within the parity bar for the benchmark-size spline flows only (config 5's 30-wide
# conditioner sits AT the bar in this mode - 1.18e-5 at input scale 3 since the spline's softmax lost its shift in
# round 5, 9e-6 before - and is held to the mode's documented envelope; `auto` takes split-f16 for it)
BF16X3_WITHIN_BAR = {"nsf_c3", "nsf_c4_d64"}


@pytest.mark.parametrize("scale", [0.5, 1.0, 3.0])
def test_log_prob_matches_oracle(pair, scale, request):
    ref, flow, D = pair
    x = torch.randn(4000, D) * scale  # scale 3 puts rows beyond the spline tail bound
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
    lp = flow.log_prob(x.to(DEV)).cpu()
    name = request.node.callspec.params["pair"][0]
    if flow.precision in ("f32", "f16x3") or name in BF16X3_WITHIN_BAR:  # split-f16: float32-class, every flow
        # TRULY relative (|a - b| / |b|; the floor only guards a log_prob within 1e-3 of zero):
        # measured <= 7.7e-6 over every configuration incl. the 2- to 5-D flows whose |log_prob|
        # goes down to 0.8 (scripts/rel_error_stats.py)
        assert rel_err(lp, lp_ref, floor=1e-3) <= LOG_PROB_RTOL
    else:
        assert rel_err(lp, lp_ref) <= 2e-4  # documented split-bf16 envelope


def _held_against_float64(name, dev, o32, o64, bar, factor=3.0):
    """The rule of tests/test_gpu_reference_bars.py for a quantity with a float64 twin: the device is inside the bar,
    or - where float32 arithmetic itself is not - no further from float64 than the float32 oracle is (x `factor`).
    `bar` is absolute on max(|value|, 1) - i.e. relative for large values.  The measured pair is printed.
    (The figures are MAXIMA over 3000 rows x D values of two float32 evaluations that round differently - torch's
    libm against the bare transcendental units, another summation order in the matrix products: x 2 between two such
    maxima is rounding noise - measured: 2.3 at worst (nsf_d5_k10_bn's inverse log|J| in split-f16: 6.6e-5 against 2.9e-5,
    the old tolerance of this test was 5e-4), 1.6 for nsf_d4's - where x 1.5 on the single row of the property tests
    is not.)"""
    o64 = o64.double()
    scale = o64.abs().clamp_min(1.0)
    e_dev = float(((dev.cpu().double() - o64).abs() / scale).max())
    e_o32 = float(((o32.double() - o64).abs() / scale).max())
    ok = e_dev <= bar or e_dev <= factor * e_o32
    tag = "" if e_dev <= bar else f"  [bar {bar:.0e} missed by float32 itself: oracle {e_o32:.2e}]"
    print(f"{name}: device {e_dev:.2e} from float64, float32 oracle {e_o32:.2e}{tag}")
    return ok


def _twin64(ref):
    """float64 copy of an oracle flow (eval mode); the LU-linear layers' cached matrices - plain attributes, which
    ``.double()`` does not convert - are dropped the way the reference drops them: ``train(); eval()``
    (``flowmodel/base.py:681-682``)."""
    r64 = copy.deepcopy(ref).double()
    r64.train()
    r64.eval()
    return r64


GATE = 1.0e-5  # |inverse(forward(x)) - x| <= 1e-5, and the bar of log_prob (relative)
BAR5 = 1.5e-5  # numpy's decimal=5, the reference's own property bar (tests/test_flows/test_included_flows.py:129-154)


def test_forward_inverse_match_oracle(pair, request):
    """z, log|J| of both directions against the oracle.  Exact-f32 and split-f16 kernels: within 1e-5 (on
    max(|value|, 1)) of the FLOAT64 twin of the oracle, or no further from it than the float32 oracle itself x 3
    (deep / wide flows: float32 round-off of x is amplified by the flow; both figures are printed).  Split-bf16: its
    documented envelope against the float32 oracle."""
    ref, flow, D = pair
    name = request.node.callspec.params["pair"][0]
    x = torch.randn(3000, D) * 1.5
    with torch.no_grad():
        z_ref, ld_ref = ref.forward(x)
        xi_ref, ldi_ref = ref.inverse(x)
    z, ld = flow.forward(x.to(DEV))
    xi, ldi = flow.inverse(x.to(DEV))
    if flow.precision == "bf16x3":
        k = 10.0
        assert rel_err(z.cpu(), z_ref) <= X_ATOL * k
        assert rel_err(ld.cpu(), ld_ref) <= LOGDET_ATOL * k
        assert rel_err(xi.cpu(), xi_ref) <= X_ATOL * k
        assert rel_err(ldi.cpu(), ldi_ref) <= LOGDET_ATOL * k
        return
    r64 = _twin64(ref)
    with torch.no_grad():
        z64, ld64 = r64.forward(x.double())
        xi64, ldi64 = r64.inverse(x.double())
    for what, dev, o32, o64 in (("forward z", z, z_ref, z64), ("forward log|J|", ld, ld_ref, ld64),
                                ("inverse x", xi, xi_ref, xi64), ("inverse log|J|", ldi, ldi_ref, ldi64)):
        assert _held_against_float64(f"{name} {flow.precision} {what}", dev, o32, o64, GATE), what


def test_forward_and_log_prob_consistent(pair):
    """forward_and_log_prob == (forward, log_prob) exactly
    (reference tests/test_flows/test_included_flows.py:114-126)."""
    _, flow, D = pair
    x = torch.randn(1000, D, device=DEV)
    z, lp = flow.forward_and_log_prob(x)
    z2, _ = flow.forward(x)
    lp2 = flow.log_prob(x)
    assert torch.equal(z, z2)
    assert torch.equal(lp, lp2)


def test_invertibility(pair, request):
    """x -> z -> x and log_J = -log_J_inv to 5 decimals (reference tests/test_flows/test_included_flows.py:144-154,
    ``decimal=5``; SURVEY.md section 8d: ``|inverse(forward(x)) - x| <= 1e-5``) for EVERY configuration - with the rule
    of tests/test_gpu_reference_bars.py where float32 arithmetic itself misses a bar (deep / wide flows; the reference
    marks these tests ``flaky(reruns=5)`` at D in {2, 4} already): the float32 oracle's own figure for the same property
    on the same weights and inputs is measured, and the device must not be further out than it x 1.5.  Both figures are
    printed (config 3's and config 4's flows included: the numbers DESIGN.md section 2 quotes)."""
    ref, flow, D = pair
    name = request.node.callspec.params["pair"][0]
    x = torch.randn(1000, D, device=DEV)
    z, log_J = flow.forward(x)
    x_out, log_J_out = flow.inverse(z)
    if flow.precision == "bf16x3":
        np.testing.assert_array_almost_equal(x.cpu().numpy(), x_out.cpu().numpy(), decimal=3)
        np.testing.assert_array_almost_equal(log_J.cpu().numpy(), -log_J_out.cpu().numpy(), decimal=3)
        return
    with torch.no_grad():
        z_o, lj_o = ref.forward(x.cpu())
        xo_o, lji_o = ref.inverse(z_o)
    e_x, o_x = float((x_out - x).abs().max()), float((xo_o - x.cpu()).abs().max())
    e_j, o_j = float((log_J + log_J_out).abs().max()), float((lj_o + lji_o).abs().max())
    print(f"{name} {flow.precision}: |inverse(forward(x)) - x| device {e_x:.2e}, float32 oracle {o_x:.2e} (gate {GATE:.0e}); "
          f"|log_J + log_J_inv| device {e_j:.2e}, float32 oracle {o_j:.2e} (bar {BAR5:.1e})")
    assert e_x <= GATE or e_x <= 1.5 * o_x, (name, e_x, o_x)
    if not (e_j <= BAR5 or e_j <= 1.5 * o_j):
        # the sum of the two log-Jacobians is a difference of rounding errors, and the float32 oracle can come out ahead
        # by cancellation (tests/test_gpu_reference_bars.py): taken apart against the float64 twin, term by term
        r64 = _twin64(ref)
        with torch.no_grad():
            z64, lj64 = r64.forward(x.cpu().double())
            _, lji64_dev = r64.inverse(z.cpu().double())
            _, lji64_o = r64.inverse(z_o.double())
        for term, dev, o32 in (("forward log|J|", (log_J.cpu() - lj64).abs().max(), (lj_o - lj64).abs().max()),
                               ("inverse log|J| at its own z", (log_J_out.cpu() - lji64_dev).abs().max(),
                                (lji_o - lji64_o).abs().max()),
                               ("z", (z.cpu() - z64).abs().max(), (z_o - z64).abs().max())):
            print(f"    {term}: device {float(dev):.2e} from float64, float32 oracle {float(o32):.2e}")
            # (x 2: maxima of two float32 evaluations that round differently; measured worst 1.8, nsf_d2 in split-f16)
            assert float(dev) <= max(BAR5, 2.0 * float(o32)), (name, term, float(dev), float(o32))


def test_sample_and_log_prob(pair):
    """sample_and_log_prob agrees with log_prob of the samples
    (reference tests/test_flows/test_included_flows.py:129-141)."""
    _, flow, D = pair
    x, lp = flow.sample_and_log_prob(1000)
    assert x.shape == (1000, D) and lp.shape == (1000,)
    lp2 = flow.log_prob(x)
    assert rel_err(lp.cpu(), lp2.cpu()) <= 1e-4


def test_shapes_and_api(pair):
    _, flow, D = pair
    assert flow.sample(10).shape == (10, D)
    z = flow.sample_latent_distribution(7)
    assert z.shape == (7, D) and z.device.type == "cuda"
    assert flow.base_distribution_log_prob(z).shape == (7,)
    flow.freeze_transform()
    assert not any(p.requires_grad for p in flow._transform.parameters())
    flow.unfreeze_transform()
    assert all(p.requires_grad for p in flow._transform.parameters())


@pytest.mark.parametrize("n", [1, 31, 32, 33, 257])
def test_ragged_batches(n):
    """Batches that do not fill a 32-row tile / a workgroup."""
    ref, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 3, 2))
    x = torch.randn(n, 6) * 2
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
        xi_ref, _ = ref.inverse(x)
    assert rel_err(flow.log_prob(x.to(DEV)).cpu(), lp_ref) <= LOG_PROB_RTOL
    assert rel_err(flow.inverse(x.to(DEV))[0].cpu(), xi_ref) <= X_ATOL


def test_empty_batch():
    _, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2))
    z, ld = flow.forward(torch.empty(0, 4, device=DEV))
    assert z.shape == (0, 4) and ld.shape == (0,)


def test_weights_update_repacks_image():
    """The device image follows in-place weight updates and load_state_dict."""
    ref, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2))
    x = torch.randn(200, 4)
    before = flow.log_prob(x.to(DEV)).cpu()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(1.1)
    flow.load_state_dict(ref.state_dict())
    after = flow.log_prob(x.to(DEV)).cpu()
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
    assert not torch.equal(before, after)
    assert rel_err(after, lp_ref) <= LOG_PROB_RTOL


def test_spline_tails_and_knots():
    """Analytic known answers of the spline: identity outside
    [-B, B] with zero log-det; with all-zero conditioner output the bins are
    equal and the boundary derivative is 1."""
    _, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (2, 8, 1, 1),
                        dict(linear_transform=None))
    with torch.no_grad():
        for p in flow.parameters():
            p.zero_()
    x = torch.tensor([[0.3, 7.0], [-6.0, -0.2], [5.0, 5.0], [-5.0, 2.5]], device=DEV)
    z, ld = flow.forward(x)
    # features beyond the tail bound are untouched
    assert z[0, 1].item() == 7.0 and z[1, 0].item() == -6.0
    # zero parameters: knots at multiples of 10/8 map to themselves
    xk = torch.tensor([[1.25, 0.0], [-2.5, 0.0], [3.75, 0.0]], device=DEV)
    zk, _ = flow.forward(xk)
    np.testing.assert_allclose(zk.cpu().numpy()[:, 0], [1.25, -2.5, 3.75], atol=2e-6)


@pytest.mark.parametrize("bins", [8, 5, 2, 13])
def test_spline_input_exactly_at_the_tail_bound(bins):
    """x == +-tail_bound is still `inside` (nflows: ``inputs <= right``): it belongs to the LAST real bin, y = +-B and
    log|J| = 0 in both directions - also in the generic instance (5, 2, 13 bins on 16-bin padded parameter sets),
    where knot k_real = B is an interior entry of the searched table (round-5 advisor finding: NaN there)."""
    ref, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (2, 8, 1, 1), dict(linear_transform=None, num_bins=bins))
    x = torch.tensor([[0.3, 5.0], [-0.7, -5.0], [5.0, 0.1], [-5.0, -0.2], [5.0, 5.0], [-5.0, -5.0]])
    with torch.no_grad():
        z_ref, ld_ref = ref.forward(x)
        xi_ref, ldi_ref = ref.inverse(x)
    for prec in ("f32", "f16x3"):
        flow.precision = prec
        z, ld = flow.forward(x.to(DEV))
        xi, ldi = flow.inverse(x.to(DEV))
        for got, want in ((z, z_ref), (ld, ld_ref), (xi, xi_ref), (ldi, ldi_ref)):
            assert torch.isfinite(got).all()
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-5)


def test_cpu_tensors_fail_loudly():
    _, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        flow.log_prob(torch.randn(4, 4))


def test_configure_model_factory():
    """flows/utils.py:208-246 argument order and custom flow class plug-in
    (reference tests/test_flows/test_flow_utils.py:274-293)."""
    flow = configure_model(dict(n_inputs=4, n_neurons=8, n_blocks=3, n_layers=1, ftype="nsf",
                                activation="relu", num_bins=4))
    leaves = [t for t in flow._transform._transforms]
    assert sum("RationalQuadratic" in type(t).__name__ for t in leaves) == 3
    assert len(flow._transform._transforms[1].transform_net.blocks) == 1

    class Custom(RealNVP):
        pass

    flow = configure_model(dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=2, flow=Custom))
    assert isinstance(flow, Custom)
    with pytest.raises(RuntimeError, match="Must specify either 'flow' or 'ftype'"):
        configure_model(dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=2))


def test_precision_selection():
    """Flow API default: exact f32.  Populate 'auto': split-f16 first, split-bf16 behind it inside its
    validated envelope."""
    from nessai_amd import pack
    from nessai_amd.flows.base import auto_candidates, bf16x3_envelope, populate_precision

    _, small = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2))
    assert small.precision == "f32"
    header, _ = small.device_image()
    assert header[pack.H_PREC] == pack.PRECISIONS["f32"]
    small.precision = "bf16x3"
    header, _ = small.device_image()
    assert header[pack.H_PREC] == pack.PRECISIONS["bf16x3"]
    small.precision = "f16x3"
    header, _ = small.device_image()
    assert header[pack.H_PREC] == pack.PRECISIONS["f16x3"]
    assert populate_precision(small) == "f16x3" and auto_candidates(small) == ("f16x3",)
    _, c3 = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (32, 64, 6, 2))
    assert populate_precision(c3) == "f16x3" and auto_candidates(c3) == ("f16x3", "bf16x3")
    _, c2 = make_pair(RealNVP, O.RealNVP, (16, 32, 4, 2))
    assert auto_candidates(c2) == ("f16x3",) and not bf16x3_envelope(c2)
    _, c5 = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (15, 30, 4, 2))
    assert bf16x3_envelope(c5)
    _, narrow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (8, 64, 4, 2))  # small |log_prob|: 1.4e-5 in bf16x3
    assert not bf16x3_envelope(narrow)


@pytest.mark.parametrize("shape", [(16, 32, 6), (24, 48, 6), (32, 32, 6), (15, 30, 8)])
def test_split_bf16_envelope_of_the_populate_policy(shape):
    """Every flow shape inside the static split-bf16 envelope of the 'auto' policy stays inside the
    1e-5 bar on randomised weights (3 seeds x 3 input scales)."""
    from nessai_amd.flows.base import bf16x3_envelope

    D, H, B = shape
    worst = 0.0
    for seed in (1, 2, 3):
        ref, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (D, H, B, 2), {}, seed=seed)
        assert bf16x3_envelope(flow)
        flow.precision = "bf16x3"
        for scale in (0.5, 1.0, 3.0):
            x = torch.randn(3000, D, generator=torch.Generator().manual_seed(seed)) * scale
            with torch.no_grad():
                expected = ref.log_prob(x).numpy()
            worst = max(worst, rel_err(flow.log_prob(x.to(DEV)).cpu().numpy(), expected))
    assert worst < LOG_PROB_RTOL


def test_bench_configuration_meets_the_bar_in_split_bf16():
    """The headline configuration (trained 32-D NSF) in the precision the populate kernel uses."""
    from helpers import load_c3_fixture

    state, shift, scale, live = load_c3_fixture()
    ref = O.NeuralSplineFlow(32, 64, 6, 2)
    ref.load_state_dict(state)
    ref.eval()
    flow = NeuralSplineFlow(32, 64, 6, 2)
    flow.load_state_dict(state)
    flow.to(DEV)
    flow.eval()
    flow.precision = "bf16x3"
    x = torch.from_numpy(((live - shift) / scale).astype(np.float32))
    z = torch.randn(20000, 32)
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
        x_ref, ld_ref = ref.inverse(z)
        lq_ref = ref.base_distribution_log_prob(z) - ld_ref
    lp = flow.log_prob(x.to(DEV)).cpu()
    xi, ld = flow.inverse(z.to(DEV))
    lq = flow.base_distribution_log_prob(z.to(DEV)).cpu() - ld.cpu()
    assert rel_err(lp, lp_ref) <= LOG_PROB_RTOL
    assert rel_err(lq, lq_ref) <= LOG_PROB_RTOL
    assert rel_err(xi.cpu(), x_ref) <= X_ATOL


def test_full_size_properties_c3():
    """Size-independent properties at the benchmark batch (1e6 x 32): round trip and
    log-det antisymmetry hold for every row."""
    torch.manual_seed(1451)
    flow = NeuralSplineFlow(32, 64, 6, 2)
    flow.to(DEV)
    flow.eval()
    z = torch.randn(1_000_000, 32, device=DEV)
    x, ld = flow.inverse(z)
    z2, ld2 = flow.forward(x)
    assert torch.isfinite(x).all() and torch.isfinite(ld).all()
    assert (z - z2).abs().max().item() < 5e-4
    assert (ld + ld2).abs().max().item() < 5e-3


@pytest.mark.parametrize("name,n_flows,n", [("nsf_c4_d64", 5, 1000), ("nsf_d4", 3, 77), ("rnvp_c2", 4, 2500)])
def test_log_prob_many_is_the_per_flow_loop(name, n_flows, n):
    """One grouped launch == the loop of ImportanceFlowModel.log_prob_all
    (flowmodel/importance.py:114-129), bit for bit, and within the bar of the oracle."""
    from nessai_amd.flows.base import log_prob_many

    P, Oc, args, kw = CONFIGS[name]
    pairs = []
    for i in range(n_flows):
        pairs.append(make_pair(P, Oc, args, kw, seed=100 + i))
    x = torch.randn(n, args[0], generator=torch.Generator().manual_seed(3)).to(DEV)
    many = log_prob_many([f for _, f in pairs], x)
    assert many.shape == (n_flows, n)
    for i, (ref, flow) in enumerate(pairs):
        single = flow.log_prob(x)
        assert torch.equal(many[i], single)
        with torch.no_grad():
            expected = ref.log_prob(x.cpu()).numpy()
        assert rel_err(many[i].cpu().numpy(), expected) < LOG_PROB_RTOL


def test_log_prob_many_mixed_structures_falls_back_to_per_flow_launches():
    from nessai_amd.flows.base import log_prob_many

    _, a = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2), {})
    _, b = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (4, 16, 2, 2), {})
    x = torch.randn(50, 4).to(DEV)
    out = log_prob_many([a, b], x)
    assert torch.equal(out[0], a.log_prob(x)) and torch.equal(out[1], b.log_prob(x))
    assert log_prob_many([a], x[:0]).shape == (1, 0)


def test_auto_precision_is_measured_per_set_of_weights(monkeypatch):
    """``precision = "auto"`` (what ``FlowProposal.populate`` asks for): a probe of the flow's CURRENT
    weights through the float32 kernel and each fast mode decides.  Split-f16 - float32-class products -
    is taken at initialisation scale, and also for the same flow with its conditioner weights blown up
    (an ill-conditioned flow, where the float32 kernel itself is ~1e-4 from float64): there it is
    accepted by the float32 kernel's own one-ulp response, and it must be as close to the float64 twin
    as the float32 oracle is.  Split-bf16 (probed first here) is off by 1e-4..1e-2 on such a flow
    (``scripts/fuzz_precision.py``) and falls through; the choice follows the weights."""
    ref, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (32, 64, 6, 2), {})
    flow.precision = "auto"
    x = torch.randn(2000, 32)
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
        lp = flow.log_prob(x.to(DEV)).cpu()
    assert flow.effective_precision == "f16x3" and flow.probe_error <= flow.AUTO_PROBE_BAR
    assert rel_err(lp, lp_ref, floor=1e-3) <= LOG_PROB_RTOL

    monkeypatch.setenv("NESSAI_AMD_AUTO_ORDER", "bf16x3,f16x3")
    flow._image_key = None
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV)).cpu()
    assert flow.effective_precision == "bf16x3" and flow.probe_error <= flow.AUTO_PROBE_BAR
    assert rel_err(lp, lp_ref, floor=1e-3) <= LOG_PROB_RTOL

    with torch.no_grad():
        for r, f in zip(ref.parameters(), flow.parameters()):
            if r.ndim == 2:
                r.mul_(3.0)
                f.mul_(3.0)
        lp_ref = ref.log_prob(x)
        lp64 = copy.deepcopy(ref).double().log_prob(x.double())
        lp = flow.log_prob(x.to(DEV)).cpu()
    assert flow.probe_errors["bf16x3"] > flow.AUTO_PROBE_BAR  # split-bf16 fails the probe ...
    assert flow.effective_precision == "f16x3"  # ... split-f16 takes over
    noise = flow.probe_errors.get("f32_one_ulp", 0.0)
    assert flow.probe_errors["f16x3"] <= max(flow.AUTO_PROBE_BAR, flow.AUTO_NOISE_FACTOR * noise)
    # float32 itself is ~1e-4 from float64 on this flow: the device must be as good as the float32 oracle
    assert rel_err(lp, lp64, floor=1e-3) <= 3.0 * rel_err(lp_ref, lp64, floor=1e-3)
    flow.precision = "f32"
    with torch.no_grad():
        lp32 = flow.log_prob(x.to(DEV)).cpu()
    assert rel_err(lp, lp64, floor=1e-3) <= 3.0 * rel_err(lp32, lp64, floor=1e-3)

    # a flow outside the split-bf16 envelope never probes that mode
    monkeypatch.delenv("NESSAI_AMD_AUTO_ORDER")
    _, small = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2), {})
    small.precision = "auto"
    small.log_prob(torch.randn(10, 4, device=DEV))
    assert small.effective_precision == "f16x3" and "bf16x3" not in small.probe_errors


def test_split_f16_reports_operands_beyond_the_f16_range():
    """The split-f16 products hold for operands below 65520 only.  A call that meets a larger one is
    detected by the kernel (status[1]: such an operand turns every accumulator of its row into NaN): a flow pinned to the mode
    raises, ``precision = "auto"`` repeats the call in exact float32 and stops choosing the mode until the
    weights change.  Ordinary batches never trip the flag."""
    from nessai_amd.flows.base import F16RangeError

    for cls, ocls, args, kw in [(NeuralSplineFlow, O.NeuralSplineFlow, (6, 16, 3, 2), {}),
                                (RealNVP, O.RealNVP, (16, 32, 4, 2), {})]:
        ref, flow = make_pair(cls, ocls, args, kw)
        D = args[0]
        x = torch.randn(3000, D, generator=torch.Generator().manual_seed(2)) * 2.0
        flow.precision = "f16x3"
        flow.log_prob(x.to(DEV))
        assert flow._status.tolist()[1] == 0
        if cls is NeuralSplineFlow:  # (a random affine flow's inverse divides by scales down to 1e-3: it leaves
            flow.inverse(x.to(DEV))  # the f16 range by itself on wide inputs)
            assert flow._status.tolist()[1] == 0
        big = x.clone()
        big[1234, D - 1] = 7.0e4  # an identity feature of some coupling layer: goes into a matrix product
        big[77, 0] = -7.0e4
        with pytest.raises(F16RangeError, match="f16 range"):
            flow.log_prob(big.to(DEV))
        flow.precision = "f32"
        expected = flow.log_prob(big.to(DEV))
        flow.precision = "auto"
        got = flow.log_prob(x.to(DEV))
        assert flow.effective_precision == "f16x3"
        got = flow.log_prob(big.to(DEV))
        assert flow.effective_precision == "f32"  # blocked for these weights
        assert np.array_equal(got.cpu().numpy(), expected.cpu().numpy(), equal_nan=True)
        flow.log_prob(x.to(DEV))
        assert flow.effective_precision == "f32"
        with torch.no_grad():
            for p in flow.parameters():
                p.mul_(1.0001)  # new weights: probed again
        flow.log_prob(x.to(DEV))
        assert flow.effective_precision == "f16x3"


def test_images_rebuilt_on_the_device_equal_the_host_pack():
    """After a change of the weights the image of a couplings-and-permutations flow is rebuilt on the device
    (``nsa_repack_image``, one gather launch with the map of ``pack.repack_map``) instead of packed on the host:
    bit for bit the image ``pack_flow`` gives for the same weights, in every precision, also when the weights
    changed behind torch's version counters (the trainer's kernels: ``_image_key = None``); flows with LU-linear /
    batch-norm words keep the host pack."""
    from nessai_amd import pack
    from nessai_amd.flows.modules import describe

    for cls, ocls, args, kw, pure in [
        (NeuralSplineFlow, O.NeuralSplineFlow, (32, 64, 6, 2), {}, True),
        (NeuralSplineFlow, O.NeuralSplineFlow, (64, 128, 3, 2), {}, True),
        (RealNVP, O.RealNVP, (7, 33, 3, 1), dict(linear_transform="permutation", batch_norm_between_layers=False), True),
        (RealNVP, O.RealNVP, (16, 32, 4, 2), {}, False),
    ]:
        ref, flow = make_pair(cls, ocls, args, kw)
        x = torch.randn(300, args[0], device=DEV)
        for precision in ("f32", "bf16x3", "f16x3"):
            flow.precision = precision
            flow.log_prob(x)  # first image of this (structure, precision): host pack
            for round_ in range(2):
                with torch.no_grad():
                    for p in flow.parameters():
                        p.mul_(1.05)
                if round_ == 1:  # a change torch's version counters do not see
                    with torch.no_grad():
                        next(flow.parameters()).data.view(-1)[0] += 0.25
                    flow._image_key = None
                _, image = flow.device_image()
                expected = pack.pack_flow(dict(describe(flow), precision=precision))
                assert np.array_equal(image.cpu().numpy().view(np.uint32), expected), (cls.__name__, precision, round_)
            assert flow.__dict__.get("_repack_pure") is pure
        # the rebuilt image is what the kernels run: against the oracle with the final weights
        ref.load_state_dict({k: v.cpu() for k, v in flow.state_dict().items()})
        flow.precision = "f16x3"
        with torch.no_grad():
            assert rel_err(flow.log_prob(x).cpu(), ref.log_prob(x.cpu()), floor=1e-3) <= LOG_PROB_RTOL


def test_auto_noise_floor_branch_on_a_flow_captured_mid_run(monkeypatch):
    """The loosened branch of ``precision = "auto"`` on REAL weights: ``tests/golden/c3_ns_retrain_noise_floor.npz``
    is the flow of one retraining of a 60 000-iteration nested-sampling run on config 3 (``scripts/run_ns_c3.py
    --iterations 60000 --seed 42 --save-branch``; about one training in thirty of such runs lands here) whose probe
    shows split-f16 1.9e-5 from the float32 kernel - above the 5e-6 bar - while the float32 kernel itself moves
    by a comparable amount when every probe input moves by ONE ulp: the flow is ill-conditioned there, no float32
    evaluation is unique to 1e-5.  The documented rule: split-f16 is kept if it is within
    4 x that one-ulp response and never left the f16 range - and what that buys is checked here against a
    float64 twin of the oracle: split-f16 is as close to float64 as the exact-float32 kernel and as the float32
    oracle (``flow.log_prob`` semantics: src/nessai/flows/base.py:236-245).

    Round 5: with the spline's softmax evaluated without its shift (csrc/nsa_device.h, NsaRQ::knots) the SAME weights
    probe at 7e-7 - the two kernels no longer disagree through which of two nearly equal logits is the maximum -
    so this flow now passes the plain bar.  The branch is still the rule for flows that do not: it is exercised
    here by lowering the bar below the measured difference; the float64 comparison is unchanged."""
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_ns_retrain_noise_floor.npz"))
    state = {k: torch.from_numpy(gold[k]) for k in gold.files if not k.startswith("__")}
    torch.manual_seed(0)
    ref = O.NeuralSplineFlow(32, 64, 6, 2)
    ref.load_state_dict(state)
    ref.eval()
    flow = NeuralSplineFlow(32, 64, 6, 2)
    flow.load_state_dict(state)
    flow.to(DEV)
    flow.eval()
    flow.precision = "auto"
    flow.device_image()
    assert flow.effective_precision == "f16x3" and flow.probe_errors["f16x3"] <= flow.AUTO_PROBE_BAR
    assert flow.probe_errors["f16x3"] < 0.25 * float(gold["__probe_f16x3"])  # (1.9e-5 with the shifted softmax)
    # the rule of the loosened branch: above the (lowered) bar, inside the noise-floor allowance, kept
    monkeypatch.setattr(type(flow), "AUTO_PROBE_BAR", 0.5 * flow.probe_errors["f16x3"])
    flow._image_key = None
    flow.device_image()
    noise = flow.probe_errors.get("f32_one_ulp")
    assert flow.probe_errors["f16x3"] > flow.AUTO_PROBE_BAR and noise is not None
    assert flow.probe_errors["f16x3"] <= flow.AUTO_NOISE_FACTOR * noise
    assert flow.effective_precision == "f16x3" and flow._status.tolist()[1] == 0

    # what it means for the results: the probe's own latents through the oracle in float32 and in float64
    z = torch.randn(flow.AUTO_PROBE_ROWS, 32, generator=torch.Generator().manual_seed(20260930))
    ref64 = copy.deepcopy(ref).double()
    with torch.no_grad():
        x64, ld64 = ref64.inverse(z.double())
        lp64 = ref64.log_prob(x64)
        x32, ld32 = ref.inverse(z)
        lp32 = ref.log_prob(x64.float())
        out = {}
        for precision in ("f16x3", "f32"):
            flow.precision = precision
            x, ld = flow.inverse(z.to(DEV))
            lp = flow.log_prob(x64.float().to(DEV))
            out[precision] = (ld.cpu().double(), lp.cpu().double())
    log_q64 = (-0.5 * (z.double() ** 2).sum(1) - 16 * np.log(2 * np.pi)) - ld64

    def err(ld, lp):  # the probe's measure, against the float64 twin
        return max(float(((ld - ld64).abs() / log_q64.abs().clamp_min(1e-3)).max()),
                   float(((lp - lp64).abs() / lp64.abs().clamp_min(1e-3)).max()))

    e16, e32k, e32o = err(*out["f16x3"]), err(*out["f32"]), err(ld32.double(), lp32.double())
    print(f"against float64: f16x3 {e16:.2e}, f32 kernel {e32k:.2e}, f32 oracle {e32o:.2e}; probe {flow.probe_errors}")
    assert max(e32k, e32o) > 3e-6                      # float32 itself is not unique to the bar here ...
    assert e16 <= 3.0 * max(e32k, e32o)                # ... and split-f16 is float32-class: as close as either
    assert e16 <= 1e-4                                 # (and nowhere near the 1e-3 of plain 16-bit operands)


@pytest.mark.parametrize("act_name", ["tanh", "silu"])
@pytest.mark.parametrize("gain", [0.02, 1.0, 30.0])
def test_bare_unit_activations_over_their_whole_range(act_name, gain):
    """tanh / silu of the conditioner run on the bare v_exp_f32 / v_rcp_f32 units (csrc/nsa_device.h nsa_tanh_fast /
    nsa_silu_fast).  Weights scaled so that the pre-activations sit inside the polynomial branch of tanh (gain 0.02),
    around one, or deep in saturation where 2^(2x log2 e) overflows (gain 30): exact-f32 mode within the log_prob bar of
    the float32 oracle or, where that oracle is itself further from its float64 twin, within 3x its error; never a NaN."""
    import copy

    acts = {"tanh": torch.tanh, "silu": torch.nn.functional.silu}
    ref, flow = make_pair(RealNVP, O.RealNVP, (6, 32, 3, 2), dict(activation=acts[act_name], net="mlp"), seed=17)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith("weight") and p.ndim == 2 and p.shape[0] == 32:  # the layers in front of an activation
                p.mul_(gain)
    flow.load_state_dict(ref.state_dict())
    flow.precision = "f32"
    x = torch.randn(4097, 6, generator=torch.Generator().manual_seed(5)) * 2.0
    ref64 = copy.deepcopy(ref).double()
    with torch.no_grad():
        lp_ref, lp64 = ref.log_prob(x), ref64.log_prob(x.double())
        lp = flow.log_prob(x.to(DEV)).cpu()
    assert bool(torch.isfinite(lp).all())
    err = rel_err(lp, lp_ref, floor=1e-3)
    assert err <= 1e-5 or rel_err(lp, lp64, floor=1e-3) <= 3.0 * rel_err(lp_ref, lp64, floor=1e-3), err


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_affine_coupling_with_a_plain_sigmoid_scale_and_saturated_scales(precision):
    """``scale_activation = sigmoid(x + a)`` - the family's floor b = 0 (``flows/realnvp.py``: any 0 <= b < 1 is
    accepted) - with raw scales far in the negative: every scale is ~2e-22, a product of four Jacobian factors would
    underflow to 0 (log|det J| = -inf, the row dropped) where the reference's sum of logarithms is finite (~ -50 per
    feature).  The kernel takes one logarithm per feature when the floor is below 1e-9 (NsaTransformConsts::ld_group4)."""
    kw = dict(linear_transform="permutation", batch_norm_between_layers=False,
              scale_activation=lambda x: torch.sigmoid(x + 2.0))
    for blocks in (2, 1):
        ref, flow = make_pair(RealNVP, O.RealNVP, (8, 16, blocks, 2), kw)
        flow.precision = precision
        sd = ref.state_dict()
        for k in list(sd):
            if k.endswith("final_layer.bias"):
                half = sd[k].numel() // 2
                sd[k][half:] = -52.0  # nflows: the second half of the conditioner's outputs are the raw scales
            if k.endswith("final_layer.weight"):
                sd[k] *= 0.01
        ref.load_state_dict(sd)
        flow.load_state_dict(sd)
        x = torch.randn(2000, 8)
        with torch.no_grad():
            z_ref, ld_ref = ref.forward(x)
        z, ld = flow.forward(x.to(DEV))
        assert torch.isfinite(ld_ref).all() and float(ld_ref.max()) < -150.0
        assert torch.isfinite(ld).all()
        assert rel_err(ld.cpu(), ld_ref) <= 1e-5
        assert rel_err(flow.log_prob(x.to(DEV)).cpu(), ref.log_prob(x).detach()) <= LOG_PROB_RTOL
        if blocks == 1:  # the inverse of one coupling: x / 2e-22 stays finite, log|det J_inv| ~ +200
            with torch.no_grad():
                xi_ref, ldi_ref = ref.inverse(x)
            xi, ldi = flow.inverse(x.to(DEV))
            assert torch.isfinite(ldi).all() and rel_err(ldi.cpu(), ldi_ref) <= 1e-5
            assert rel_err(xi.cpu() / 1e22, xi_ref / 1e22) <= 1e-4
