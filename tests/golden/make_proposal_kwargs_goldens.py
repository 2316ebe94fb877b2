#!/usr/bin/env python
"""Golden vectors for the rest of ``BaseFlowProposal``'s keyword surface, from the REFERENCE's own ``FlowProposal``:
``reverse_reparameterisations`` and ``map_to_unit_hypercube`` (``src/nessai/proposal/flowproposal/base.py:116-121,
277-292, 589, 744-797, 1053-1128``; ``flowproposal.py:345-389``).

Run in the build container only (needs /root/reference; the flow arithmetic under the ``glasflow`` names is the
oracle's, ``oracle/reference_harness.py``):

    python tests/golden/make_proposal_kwargs_goldens.py

Stores inputs and the reference's outputs of ``check_state`` + ``rescale`` / ``inverse_rescale`` (both orders), and for
the unit-hypercube proposal also ``prior_bounds``, ``compute_weights``, ``convert_to_samples`` and ``backward_pass`` on
fixed latents with the flow's weights.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import reference_harness  # noqa: E402

reference_harness.install()

import torch  # noqa: E402
from nessai.livepoint import empty_structured_array, numpy_array_to_live_points  # noqa: E402
from nessai.model import Model  # noqa: E402
from nessai.proposal import FlowProposal  # noqa: E402

NAMES = ["a", "b", "c", "d", "e"]
BOUNDS = {"a": [-3.0, 7.0], "b": [0.5, 2.5], "c": [-1.0, 1.0], "d": [10.0, 20.0], "e": [-4.0, 4.0]}
# one group per parameter, in model order: the reference then keeps the x' columns in model order (it orders them by
# group), which is the order the device path feeds its flow - the flow's weights can be shared for backward_pass
REPARAMETERISATIONS = {"a": "rescaletobounds", "b": "logit", "c": "z-score", "d": "rescaletobounds"}  # "e": the fallback
OUT = {}


class BoxModel(Model):
    """Uniform prior in a box with the affine unit-hypercube maps a user would write (model.py:603-615)."""

    def __init__(self):
        self.names = list(NAMES)
        self.bounds = {n: list(BOUNDS[n]) for n in NAMES}

    def log_prior(self, x):
        log_p = np.log(self.in_bounds(x), dtype="float")
        for lo, hi in self.bounds.values():
            log_p -= np.log(hi - lo)
        return log_p

    def log_likelihood(self, x):
        return -0.5 * np.sum(self.unstructured_view(x) ** 2, axis=-1)

    def to_unit_hypercube(self, x):
        out = x.copy()
        for n in self.names:
            lo, hi = self.bounds[n]
            out[n] = (x[n] - lo) / (hi - lo)
        return out

    def from_unit_hypercube(self, x):
        out = x.copy()
        for n in self.names:
            lo, hi = self.bounds[n]
            out[n] = lo + (hi - lo) * x[n]
        return out


def stack(x, names):
    return np.stack([np.asarray(x[n], dtype=np.float64) for n in names], axis=1)


def run(tag, live, xp_values, **kwargs):
    rng = np.random.default_rng(170817)
    torch.manual_seed(170817)
    model = BoxModel()
    model.set_rng(rng)
    with tempfile.TemporaryDirectory() as tmp:
        prop = FlowProposal(model, rng=rng, poolsize=50, drawsize=400, output=tmp, plot=False,
                            flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1,
                                             batch_norm_between_layers=False, linear_transform="permutation"),
                            reparameterisations=REPARAMETERISATIONS, **kwargs)
        prop.initialise()
        prop.check_state(live.copy())
        primes = list(prop.prime_parameters)
        OUT[f"{tag}/prime_parameters"] = np.array(primes)
        x_prime, log_j = prop.rescale(live.copy())
        OUT[f"{tag}/x_prime"] = stack(x_prime, primes)
        OUT[f"{tag}/log_j"] = np.asarray(log_j, dtype=np.float64)
        xp_in = empty_structured_array(xp_values.shape[0], names=primes)
        for n in NAMES:  # golden columns are stored in model order: "<name>_prime" (or the name itself)
            pp = n + "_prime" if n + "_prime" in primes else n
            xp_in[pp] = xp_values[:, NAMES.index(n)]
        x_back, log_j_inv = prop.inverse_rescale(xp_in.copy())
        OUT[f"{tag}/x_back"] = stack(x_back, NAMES)
        OUT[f"{tag}/log_j_inv"] = np.asarray(log_j_inv, dtype=np.float64)
        if kwargs.get("map_to_unit_hypercube"):
            OUT[f"{tag}/prior_bounds"] = np.array([prop.prior_bounds[n] for n in NAMES], dtype=np.float64)
            x_cube, log_j_cube = prop.inverse_rescale(xp_in.copy(), return_unit_hypercube=True)
            OUT[f"{tag}/x_back_cube"] = stack(x_cube, NAMES)
            OUT[f"{tag}/log_j_inv_cube"] = np.asarray(log_j_cube, dtype=np.float64)
            # weights and samples of cube points, some of them outside [0, 1)
            pts = rng.uniform(-0.05, 1.05, size=(64, len(NAMES)))
            pts[0, 0] = 1.0  # the upper face itself: outside the cube prior's half-open support
            pts[1, 1] = 0.0
            cube = numpy_array_to_live_points(pts, NAMES)
            log_q = rng.normal(size=64)
            OUT[f"{tag}/cube_points"] = pts
            OUT[f"{tag}/cube_log_q"] = log_q
            with np.errstate(divide="ignore", invalid="ignore"):
                OUT[f"{tag}/cube_log_w"] = np.asarray(prop.compute_weights(cube.copy(), log_q.copy()), dtype=np.float64)
                samples = prop.convert_to_samples(cube.copy(), plot=False)
            OUT[f"{tag}/cube_samples"] = stack(samples, NAMES)
            OUT[f"{tag}/cube_samples_logP"] = np.asarray(samples["logP"], dtype=np.float64)
            # backward pass on fixed latents, in the cube and mapped on (flowproposal.py:345-389)
            for k, v in prop.flow.model.state_dict().items():
                OUT[f"{tag}/state/{k}"] = v.detach().cpu().numpy()
            z = rng.normal(size=(300, len(NAMES))).astype(np.float32).astype(np.float64) * 1.3
            OUT[f"{tag}/bp_z"] = z
            x, log_q_bp, z_kept = prop.backward_pass(z.copy(), rescale=True, return_z=True)
            OUT[f"{tag}/bp_x"] = stack(x, NAMES)
            OUT[f"{tag}/bp_log_q"] = np.asarray(log_q_bp, dtype=np.float64)
            OUT[f"{tag}/bp_z_kept"] = np.asarray(z_kept, dtype=np.float64)
            xc, log_q_c, _ = prop.backward_pass(z.copy(), rescale=True, return_z=True, return_unit_hypercube=True)
            OUT[f"{tag}/bp_x_cube"] = stack(xc, NAMES)
            OUT[f"{tag}/bp_log_q_cube"] = np.asarray(log_q_c, dtype=np.float64)


def main():
    rng = np.random.default_rng(20261001)
    n, m = 211, 173
    live = empty_structured_array(n, names=NAMES)
    for p in NAMES:
        lo, hi = BOUNDS[p]
        live[p] = lo + (hi - lo) * rng.beta(2.0, 3.0, size=n)
    xp_values = rng.normal(0.0, 1.5, size=(m, len(NAMES)))
    xp_values[:4, 1] = [-40.0, 0.0, 36.0, 39.0]  # saturating logits
    OUT["live"] = stack(live, NAMES)
    OUT["xp_in"] = xp_values
    run("plain", live, xp_values)
    run("reversed", live, xp_values, reverse_reparameterisations=True)
    run("cube", live, xp_values, map_to_unit_hypercube=True)
    run("cube_reversed", live, xp_values, map_to_unit_hypercube=True, reverse_reparameterisations=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "proposal_kwargs_goldens.npz")
    np.savez_compressed(path, **OUT)
    print("wrote", path, len(OUT), "arrays")
    for tag in ("plain", "reversed", "cube", "cube_reversed"):
        print(tag, OUT[f"{tag}/prime_parameters"], float(OUT[f"{tag}/log_j"][0]), float(OUT[f"{tag}/log_j_inv"][5]))


if __name__ == "__main__":
    main()
