"""Host-side logic of the product package that needs no GPU."""
import numpy as np
import pytest

from nessai_amd import livepoint, pack
from nessai_amd.flowmodel import FlowModel, update_config
from nessai_amd.model import GaussianModel, RosenbrockModel
from nessai_amd.utils import compute_radius, rescale_log_jacobian


def test_live_point_dtype_and_defaults():
    """Field layout of the reference (config.py:22-60, livepoint.py:74-155)."""
    names = ["a", "b"]
    dt = livepoint.get_dtype(names)
    assert dt.names == ("a", "b", "logP", "logL", "it")
    assert [dt[n].str for n in dt.names] == ["<f8", "<f8", "<f8", "<f8", "<i4"]
    x = livepoint.empty_structured_array(3, names=names)
    assert np.isnan(x["a"]).all() and np.isnan(x["logL"]).all() and (x["it"] == 0).all()
    assert livepoint.empty_structured_array(0, names=names).size == 0


def test_fast_live_point_builder_equals_field_loop():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(17, 5))
    names = [f"p{i}" for i in range(5)]
    fast = livepoint.array_to_live_points_fast(a, names)
    slow = livepoint.numpy_array_to_live_points(a, names)
    for n in names + ["it"]:
        assert np.array_equal(fast[n], slow[n])
    assert np.isnan(fast["logP"]).all()
    assert np.array_equal(livepoint.live_points_to_array(fast, names), a)
    assert livepoint.array_to_live_points_fast(np.empty((0, 5)), names).size == 0


def test_models_follow_the_reference_examples():
    m = RosenbrockModel(3)
    x = livepoint.numpy_array_to_live_points(np.array([[1.0, 1.0, 1.0], [0.0, 0.0, 6.0]]), m.names)
    assert m.in_bounds(x).tolist() == [True, False]
    lp = m.log_prior(x)
    assert lp[0] == pytest.approx(-3 * np.log(10.0)) and lp[1] == -np.inf
    assert m.log_likelihood(x)[0] == 0.0  # Rosenbrock minimum at (1, 1, 1)
    g = GaussianModel(2)
    y = livepoint.numpy_array_to_live_points(np.zeros((1, 2)), g.names)
    assert g.log_likelihood(y)[0] == pytest.approx(-np.log(2 * np.pi))
    assert g.bounds["x_0"] == [-10.0, 10.0]


def test_radius_and_rescale_jacobian():
    assert round(float(compute_radius(2, 0.8647)), 3) == 2.000
    assert abs(compute_radius(32, 0.95) - 6.7966) < 1e-3
    scales = [0.5, 2.0, 3.0]
    assert rescale_log_jacobian(scales) == pytest.approx(np.log(3.0))


def test_check_batch_size():
    """FlowModel.check_batch_size (flowmodel/base.py:195-236; reference
    tests/test_flowmodel/test_flowmodel_base.py:145-183)."""
    for data_size, batch_size in [(4010, 1000), (106, 21), (1000, 1000), (2000, 1000)]:
        out = FlowModel.check_batch_size(np.arange(data_size), batch_size)
        assert out >= int(0.1 * batch_size)
        if not data_size % batch_size:
            assert out == batch_size
    assert FlowModel.check_batch_size(np.arange(25), 10, min_fraction=0.8) == 7
    with pytest.raises(RuntimeError, match="Could not find a valid batch size"):
        FlowModel.check_batch_size(np.arange(3), 2, min_fraction=1.0)
    with pytest.raises(ValueError, match="Cannot use a batch size of 1"):
        FlowModel.check_batch_size(np.arange(2), 1, min_fraction=1.0)


def test_update_config_defaults():
    fc, tc = update_config(dict(n_inputs=16), None)
    assert fc["n_neurons"] == 32 and fc["n_blocks"] == 4 and fc["n_layers"] == 2
    assert tc["lr"] == 1e-3 and tc["batch_size"] == 1000 and tc["patience"] == 20
    assert tc["clip_grad_norm"] == 5.0 and tc["optimiser"] == "adamw"
    with pytest.raises(RuntimeError, match="noise_scale"):
        update_config(dict(n_inputs=2), dict(noise_type="constant"))


def test_transform_policy_table():
    """(tiles per chunk, features per lane per chunk, params per feature) the kernels are built for."""
    assert pack.transform_policy("rq", 8) == (3, 2, 23)
    assert pack.transform_policy("rq", 4) == (1, 1, 11)
    assert pack.transform_policy("rq", 10) == (2, 1, 29)
    assert pack.transform_policy("rq", 16) == (3, 1, 47)
    assert pack.transform_policy("affine", 0) == (1, 8, 2)
    assert pack.transform_policy("additive", 0) == (1, 16, 1)
    assert [pack.hidden_tiles(h) for h in (8, 32, 33, 64, 65, 96, 128)] == [1, 1, 2, 2, 4, 4, 4]


def test_flow_model_requires_rocm_device(tmp_path):
    fm = FlowModel(flow_config=dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=2, ftype="nsf"),
                   training_config=dict(device_tag="cpu"), output=str(tmp_path))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fm.initialise()


def test_importance_proposal_glue():
    """logit / sigmoid maps, their Jacobians and the meta-proposal logsumexp
    (src/nessai/proposal/importance.py:203-257, 382-441; utils/rescaling.py:290-336)."""
    from scipy.special import logsumexp

    from nessai_amd.proposal_importance import EPS, ImportanceFlowProposal, ins_dtype, logit, sigmoid

    x = np.array([[0.25, 0.5], [1e-12, 1.0]])
    xp, lj = logit(x, eps=EPS)
    assert np.isfinite(xp).all()
    np.testing.assert_allclose(xp[0], np.log(x[0] / (1 - x[0])))
    np.testing.assert_allclose(lj[0], -np.log(x[0]) - np.log1p(-x[0]))
    xb, ljb = sigmoid(xp)
    np.testing.assert_allclose(xb[0], x[0])
    np.testing.assert_allclose(ljb[0], -lj[0])
    assert ins_dtype(["a"]).names == ("a", "logP", "logL", "it", "logQ", "logW", "logU")

    m = RosenbrockModel(2)
    prop = ImportanceFlowProposal(m, output="/tmp")
    prop._weights = {-1: 0.2, 0: 0.3, 1: 0.5}
    xp = np.random.default_rng(0).normal(size=(7, 2))
    lj = np.random.default_rng(1).normal(size=7)
    log_q = np.concatenate([np.zeros((7, 1)), np.stack([-np.sum(xp**2, 1), -0.5 * np.sum(xp**2, 1)], 1) + lj[:, None]], 1)
    expect = logsumexp(log_q, b=np.array([0.2, 0.3, 0.5]), axis=1)
    np.testing.assert_allclose(prop.compute_meta_proposal_from_log_q(log_q), expect)
    with pytest.raises(RuntimeError, match="Weights must sum to 1"):
        prop.update_proposal_weights({1: 0.9})
    # compute_log_Q / draw run on the device: tests/test_gpu_importance.py
    xs, ljs = prop.rescale(np.zeros(3, dtype=ins_dtype(m.names)))
    assert xs.shape == (3, 2) and ljs.shape == (3,)


def test_separable_prior_terms_are_normalised():
    """Each device prior term integrates to one on its bounds (bilby-style densities)."""
    from nessai_amd.model import GWSurrogateModel, log_prior_term

    m = GWSurrogateModel()
    assert len(m.names) == 15
    for n in m.names:
        lo, hi = m.bounds[n]
        x = np.linspace(lo, hi, 200001)[1:-1]
        p = np.exp(log_prior_term(m.device_prior[n], x, lo, hi))
        assert abs(np.trapezoid(p, x) - 1.0) < 1e-3, n
    kinds, pars = m.device_prior_tables()
    assert kinds.dtype == np.int32 and pars.shape == (15, 3)
    pts = m.new_point(50, rng=np.random.default_rng(0))
    assert m.in_bounds(pts).all() and np.isfinite(m.log_prior(pts)).all()
    assert np.isfinite(m.log_likelihood(pts)).all()


def test_scale_activation_callables_are_identified():
    """RealNVP(scale_activation=callable): the callable is probed and must be sigmoid(x + a) + b
    (the reference's own test passes the default as a lambda, test_specific_flows.py:32)."""
    import torch

    from nessai_amd.flows import RealNVP
    from nessai_amd.flows.realnvp import identify_scale_activation

    assert identify_scale_activation(lambda x: torch.sigmoid(x + 2) + 1e-3) == (2.0, 0.001)
    assert identify_scale_activation(lambda x: torch.sigmoid(x - 0.5) + 0.05) == (-0.5, 0.05)
    for bad in (torch.exp, lambda x: (torch.nn.functional.softplus(x) + 1e-3).clamp(0, 3), lambda x: x[:3],
                lambda x: torch.sigmoid(2 * x) + 0.1):
        with pytest.raises(NotImplementedError):
            identify_scale_activation(bad)
    flow = RealNVP(2, 2, 2, 2, scale_activation=lambda x: torch.sigmoid(x + 2) + 1e-3)
    from nessai_amd.flows import modules as M

    desc = M.describe(flow)
    assert desc["affine_shift"] == 2.0 and desc["affine_eps"] == 1e-3
    flow = RealNVP(4, 8, 2, 2, scale_activation=lambda x: torch.sigmoid(x - 1.0) + 0.2)
    assert M.describe(flow)["affine_shift"] == -1.0
    RealNVP(4, 8, 2, 2, use_volume_preserving=True, scale_activation=torch.exp)  # unused for additive couplings


def test_parameter_block_view_and_record_copies_equal_the_field_loops():
    """The (N, D) view of the parameter fields and the record-wise copy / mask / concatenate helpers
    (livepoint.py) give exactly what numpy's per-field operations give, and step aside for layouts
    they do not cover."""
    from nessai_amd.proposal_importance import empty_ins_array

    names = [f"x_{i}" for i in range(7)]
    rng = np.random.default_rng(1)
    a = empty_ins_array(50, names)
    for n in names:
        a[n] = rng.normal(size=50)
    a["logL"] = rng.normal(size=50)
    ref = np.stack([a[n] for n in names], axis=-1)
    block = livepoint.parameter_block(a, names)
    assert block is not None and np.array_equal(block, ref)
    block[3, 2] = 42.0  # a writable view
    assert a["x_2"][3] == 42.0
    assert np.array_equal(livepoint.live_points_to_array(a, names), np.stack([a[n] for n in names], axis=-1))
    # layouts the view does not cover fall back to the field loop
    assert livepoint.parameter_block(a[::2], names) is None
    assert livepoint.parameter_block(a, names[::-1]) is None
    assert livepoint.parameter_block(a, ["x_0", "logL"]) is None
    assert np.array_equal(livepoint.live_points_to_array(a[::2], names), np.stack([a[n][::2] for n in names], axis=-1))
    assert np.array_equal(livepoint.live_points_to_array(a, names[::-1]), np.stack([a[n] for n in names[::-1]], axis=-1))
    assert livepoint.parameter_block(a[:0], names).shape == (0, 7)

    def same(u, v):
        return u.dtype == v.dtype and all(np.array_equal(u[f], v[f], equal_nan=True) for f in u.dtype.names)

    c = livepoint.copy_points(a)
    assert same(c, a.copy()) and c is not a
    c["x_0"][0] = -1.0
    assert a["x_0"][0] != -1.0
    mask = rng.random(50) > 0.4
    assert same(livepoint.take_points(a, mask), a[mask])
    idx = np.array([5, 1, 1, 40])
    assert same(livepoint.take_points(a, idx), a[idx])
    assert same(livepoint.concatenate_points([a[:10], a[20:25]]), np.concatenate([a[:10], a[20:25]]))
    assert same(livepoint.take_points(a[::2], mask[::2]), a[::2][mask[::2]])  # non-contiguous: plain numpy


def test_vectorised_model_maps_equal_the_per_field_forms():
    m = RosenbrockModel(5)
    pts = m.new_point(200, rng=np.random.default_rng(2))
    pts["x_1"][:5] = [6.0, -6.0, 5.0, -5.0, np.nan]
    slow_in = ~np.any([(pts[n] < m.bounds[n][0]) | (pts[n] > m.bounds[n][1]) for n in m.names], axis=0)
    assert np.array_equal(m.in_bounds(pts), slow_in)
    unit = m.to_unit_hypercube(pts)
    back = m.from_unit_hypercube(unit)
    for n in m.names:
        lo, hi = m.bounds[n]
        assert np.array_equal(unit[n], (pts[n] - lo) / (hi - lo), equal_nan=True)
        assert np.array_equal(back[n], lo + (hi - lo) * unit[n], equal_nan=True)
    slow_unit = ~np.any([(unit[n] < 0.0) | (unit[n] > 1.0) for n in m.names], axis=0)
    assert np.array_equal(m.in_unit_hypercube(unit), slow_unit)


def test_poolsize_scale_limits():
    """1 / acceptance inside [1, max_poolsize_scale]; the maximum when nothing was accepted
    (proposal/flowproposal/base.py:416-435)."""
    from nessai_amd.proposal import FlowProposal

    class Stub:
        max_poolsize_scale = 10.0
        _poolsize_scale = None

    s = Stub()
    for acceptance, want in ((0.0, 10.0), (None, 10.0), (0.5, 2.0), (0.01, 10.0), (2.0, 1.0), (1.0, 1.0)):
        FlowProposal.update_poolsize_scale(s, acceptance)
        assert s._poolsize_scale == want


def test_flow_factory_schema():
    """The configuration schema of src/nessai/flows/utils.py:112-246, 295-349: width rules, flow names, the
    refusals of what has no kernel, and the errors the reference raises for what it does not know."""
    from nessai_amd.flows import NeuralSplineFlow, RealNVP, modules
    from nessai_amd.flows import utils as U

    # width of the conditioner: (n_neurons, n_inputs) -> int  (the reference's table, utils.py:141-156)
    for n_neurons, n_inputs, want in [(None, None, 8), (16, None, 16), ("12", None, 12), (None, 5, 10), ("auto", 5, 10),
                                      ("double", 7, 14), ("equal", 7, 7), ("half", 7, 3), (24, 7, 24), (8.0, 3, 8)]:
        assert U.get_n_neurons(n_neurons, n_inputs) == want
    assert U.get_n_neurons(None, None, default=3) == 3
    with pytest.raises(ValueError, match="Could not get number of neurons"):
        U.get_n_neurons("auto")  # a rule without the number of inputs
    with pytest.raises(ValueError, match="Could not get number of neurons"):
        U.get_n_neurons("wide", 4)
    for name, cls in [("realnvp", RealNVP), ("FRealNVP", RealNVP), ("nsf", NeuralSplineFlow), ("Spline", NeuralSplineFlow)]:
        assert U.get_flow_class(name) is cls
    with pytest.raises(NotImplementedError):
        U.get_flow_class("maf")
    with pytest.raises(ValueError, match="Unknown flow"):
        U.get_flow_class("glow")
    assert isinstance(U.create_linear_transform("Permutation", 4), modules.RandomPermutation)
    lu = U.create_linear_transform("lu", 4)
    assert [type(t) for t in lu._transforms] == [modules.RandomPermutation, modules.LULinear]
    svd = U.create_linear_transform("svd", 6)  # flows/utils.py:316-323: ten Householder reflections per factor
    assert [type(t) for t in svd._transforms] == [modules.RandomPermutation, modules.SVDLinear]
    assert svd._transforms[1].orthogonal_1.q_vectors.shape == (10, 6)
    with pytest.raises(ValueError, match="Unknown linear transform"):
        U.create_linear_transform("qr", 4)
    assert isinstance(U.create_pre_transform("batch_norm", 4), modules.BatchNorm)
    with pytest.raises(NotImplementedError, match="logit"):
        U.create_pre_transform("logit", 4)
    with pytest.raises(ValueError, match="Unknown pre-transform"):
        U.create_pre_transform("whiten", 4)
    with pytest.raises(TypeError, match="must be an int"):
        U.configure_model(dict(n_inputs=4.0, n_neurons=8, n_blocks=2, n_layers=2, ftype="nsf"))
    with pytest.raises(RuntimeError, match="Must specify either 'flow' or 'ftype'"):
        U.configure_model(dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=2))
    with pytest.raises(NotImplementedError, match="Base distribution"):
        U.configure_model(dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=2, ftype="nsf", distribution="lars"))
    cfg = dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=2, ftype="realnvp", distribution="mvn",
               distribution_kwargs=dict(var=2.0))
    flow = U.configure_model(cfg)
    assert isinstance(flow, RealNVP) and flow._distribution.var == 2.0
    assert cfg["ftype"] == "realnvp" and "distribution_kwargs" in cfg  # the caller's dictionary is left alone


def test_eval_mode_switches_only_what_is_in_training_mode():
    """nessai_amd.utils.eval_mode == Module.eval() in effect (every flow and every submodule in eval mode), without
    the walk over the tree when there is nothing to switch - including a flow appended to a list that was switched
    before (ImportanceFlowModel.add_new_flow)."""
    import torch

    from nessai_amd.utils import eval_mode

    def flow():
        return torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.BatchNorm1d(2))

    models = torch.nn.ModuleList([flow(), flow()])
    eval_mode(models)
    assert not any(m.training for m in models.modules())
    calls = []
    models[0].eval = lambda: calls.append("walk")  # must not be called again
    models.append(flow())  # a new flow arrives in training mode
    assert models[2].training and models[2][1].training
    eval_mode(models)
    assert calls == [] and not any(m.training for m in models.modules())
    single = flow()
    assert eval_mode(single) is single and not single[1].training


def test_device_written_draw_records_only_for_the_frameworks_own_uniform_box_prior(monkeypatch):
    """``ImportanceFlowProposal.draw`` lets the device fill the prior fields of a batch (``nsa_importance_mix_points``)
    only when every prior method it would call on the model is the framework's own uniform-box implementation; a
    subclass that overrides one of them - a user's prior - is evaluated on the host as the reference does
    (``proposal/importance.py:488-503``).  The constant handed to the kernel is ``log_prior``'s own."""
    from nessai_amd.model import CorrelatedGaussianModel, UniformBoxModel
    from nessai_amd.proposal_importance import empty_ins_array, ins_dtype, uniform_box_of

    monkeypatch.delenv("NESSAI_AMD_DRAW_POINTS", raising=False)
    model = CorrelatedGaussianModel(6, rho=0.5, bound=3.0)
    dtype = ins_dtype(model.names)
    lower, upper, total = uniform_box_of(model, dtype)
    assert np.array_equal(lower, np.full(6, -3.0)) and np.array_equal(upper, np.full(6, 3.0))
    pts = empty_ins_array(4, model.names)
    for n in model.names:
        pts[n] = 0.25
    assert np.all(model.log_prior(pts) == total)  # bit for bit the value of an in-bounds point
    # the device's record layout is the dtype's: 8 D + 44 bytes, parameters first, fields at the offsets it writes
    assert dtype.itemsize == 8 * 6 + 44
    assert [dtype.fields[f][1] for f in ("logP", "logL", "it", "logQ", "logW", "logU")] == [48, 56, 64, 68, 76, 84]

    class OwnPrior(CorrelatedGaussianModel):
        def log_prior(self, x):
            return super().log_prior(x) - 1.0

    class OwnMap(UniformBoxModel):
        def from_unit_hypercube(self, x):
            return super().from_unit_hypercube(x)

    class OwnCube(UniformBoxModel):
        def batch_evaluate_log_prior_unit_hypercube(self, x):
            return super().batch_evaluate_log_prior_unit_hypercube(x)

    assert uniform_box_of(OwnPrior(6), dtype) is None
    assert uniform_box_of(OwnMap(6), ins_dtype(OwnMap(6).names)) is None
    assert uniform_box_of(OwnCube(6), ins_dtype(OwnCube(6).names)) is None
    assert uniform_box_of(RosenbrockModel(4), ins_dtype(RosenbrockModel(4).names)) is not None or not isinstance(RosenbrockModel(4), UniformBoxModel)
    monkeypatch.setenv("NESSAI_AMD_DRAW_POINTS", "0")
    assert uniform_box_of(model, dtype) is None


def test_float64_is_refused_not_downgraded():
    """The reference's ``torch_dtype="float64"`` (``utils/torchutils.py:12-25``, ``flowsampler.py:101,121``) builds and
    evaluates its flows in float64.  The gfx950 kernels compute in float32: a float64 flow or input must RAISE, never
    be cast down in silence (round-5 verdict) - like nflows' dtype mismatch for a float64 input to a float32 flow."""
    import torch

    from nessai_amd.flows import NeuralSplineFlow, RealNVP

    flow = RealNVP(4, 8, 2, 2)
    assert all(p.dtype == torch.float32 for p in flow.parameters())
    with pytest.raises(TypeError, match="float64"):
        flow.double()
    with pytest.raises(TypeError, match="input tensor is float64"):
        flow._refuse_float64(torch.zeros(3, 4, dtype=torch.float64))
    flow._refuse_float64(torch.zeros(3, 4))  # float32: fine
    torch.set_default_dtype(torch.float64)
    try:
        with pytest.raises(TypeError, match="float64 parameters"):
            NeuralSplineFlow(4, 8, 2, 2)
    finally:
        torch.set_default_dtype(torch.float32)
