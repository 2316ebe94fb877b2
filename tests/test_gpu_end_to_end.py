"""A minimal nested-sampling caller of the proposal protocol (style of the reference's
tests/test_sampling/test_standard_sampling.py: assert the flow got trained and the run resumes)."""
import pickle

import numpy as np
import pytest
import torch

from nessai_amd.model import GaussianModel
from nessai_amd.proposal import FlowProposal

pytestmark = pytest.mark.gpu


def test_proposal_drives_a_small_nested_sampling_loop(tmp_path):
    rng = np.random.default_rng(170817)
    torch.manual_seed(170817)
    model = GaussianModel(2)
    nlive = 200
    proposal = FlowProposal(
        model, rng=rng, poolsize=nlive, drawsize=20_000, output=str(tmp_path),
        flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1),
        training_config=dict(max_epochs=30, patience=10, batch_size=100),
    )
    proposal.initialise()
    live = model.new_point(nlive, rng=rng)
    live["logL"] = model.log_likelihood(live)
    live = np.sort(live, order="logL")
    first_worst = float(live["logL"][0])
    replaced = 0
    for it in range(300):
        if it % 100 == 0:
            proposal.train(live)
        worst = live[0]
        for _ in range(200):
            new = proposal.draw(worst)
            if new["logL"] > worst["logL"]:
                break
        else:
            pytest.fail("no point above the worst likelihood in 200 draws")
        idx = np.searchsorted(live["logL"], new["logL"])
        live[: idx - 1] = live[1:idx]
        live[idx - 1] = new
        replaced += 1
    assert replaced == 300
    assert proposal.training_count == 3
    assert proposal.populated_count >= 1
    assert model.in_bounds(live).all() and np.all(np.diff(live["logL"]) >= 0)
    assert live["logL"][0] > first_worst + 10.0  # the live set contracted towards the mode
    assert proposal.population_time.total_seconds() > 0
    # the proposal pickles without its device state (resume protocol, base.py:1286-1309)
    state = pickle.loads(pickle.dumps(proposal))
    assert state.flow is None and state.populated == proposal.populated
    assert state.weights_file is not None


@pytest.mark.timeout(900)
def test_config_1_two_dimensional_gaussian_with_the_reference_defaults(tmp_path):
    """BASELINE config 1 at its own size (``examples/2d_gaussian.py``): 2-D unit Gaussian on [-10, 10]^2,
    nlive = 1000, the reference's DEFAULT flow (RealNVP, 4 coupling layers, 2-block ResNet conditioners of
    2 x n_dims neurons, LU-linear + batch norm: ``flowmodel/config.py:17-21``, ``flows/realnvp.py:84-90``) and
    default proposal settings, driven by a plain nested-sampling loop to termination.  The evidence is known:
    Z = 1 / 400 (the likelihood integrates to one inside the box), log Z = -5.9915."""
    rng = np.random.default_rng(1234)
    torch.manual_seed(1234)
    model = GaussianModel(2)
    nlive = 1000
    proposal = FlowProposal(model, rng=rng, poolsize=nlive, output=str(tmp_path),
                            training_config=dict(max_epochs=100, patience=20))
    proposal.initialise()
    fc = proposal.flow.flow_config
    assert fc["n_blocks"] == 4 and fc["n_layers"] == 2 and fc["n_neurons"] == 4 and fc["ftype"].lower() == "realnvp"
    live = model.new_point(nlive, rng=rng)
    live["logL"] = model.log_likelihood(live)
    live = np.sort(live, order="logL")
    dead_log_l = []
    n_iter = 8000
    for it in range(n_iter):
        if it % 1000 == 0:
            proposal.train(live)
        worst = live[0]
        dead_log_l.append(float(worst["logL"]))
        for _ in range(2000):
            new = proposal.draw(worst)
            if new["logL"] > worst["logL"]:
                break
        else:
            pytest.fail("no point above the worst likelihood in 2000 draws")
        idx = np.searchsorted(live["logL"], new["logL"])
        live[: idx - 1] = live[1:idx]
        live[idx - 1] = new
    # log Z = log sum_i L_i (X_{i-1} - X_i) with X_i = exp(-i / nlive), plus the live points' share of X_final
    i = np.arange(1, n_iter + 1)
    log_w = -(i - 1) / nlive + np.log1p(-np.exp(-1.0 / nlive))
    terms = np.concatenate([np.asarray(dead_log_l) + log_w,
                            live["logL"] - n_iter / nlive - np.log(nlive)])
    log_z = float(np.logaddexp.reduce(terms))
    # statistical error of nested sampling: sqrt(H / nlive) ~ sqrt(4.2 / 1000) = 0.065
    assert abs(log_z - (-np.log(400.0))) < 0.3, log_z
    assert proposal.training_count == 8 and proposal.populated_count >= 8
    assert model.in_bounds(live).all()
    # the default flow runs the float32-class split-f16 kernels by the measured policy
    assert proposal.flow.model.effective_precision in ("f16x3", "f32")
