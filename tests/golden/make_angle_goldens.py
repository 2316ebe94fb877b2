#!/usr/bin/env python
"""Golden vectors for the angle reparameterisations, from the REFERENCE's own ``FlowProposal``
(``src/nessai/reparameterisations/angle.py:17-183``: an angle becomes a point of the plane, its radius an auxiliary
parameter drawn from chi(2); registry names ``angle-2pi``, ``angle-pi``, ``angle``, ``periodic``:
``reparameterisations/__init__.py:158-177``).

Run in the build container only (needs /root/reference):

    python tests/golden/make_angle_goldens.py

A 5-parameter box model with three angles of different ranges (one starting at 0: modulo 2 pi on the way back; one
on [-pi/2, pi/2] with scale 2; one "periodic" on [0, 4]: scale 2 pi / range, modulo 2 pi) and two ordinary parameters.  Stores the
live points, the proposal's column names, ``rescale`` (the radii come from the proposal's generator: numpy
``default_rng(4242)``'s state handed to scipy's ``chi(2).rvs``), ``inverse_rescale`` on given x', ``log_prior`` of the
results (the radial prior included), and ``backward_pass`` on fixed latents with the flow's weights.
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
from nessai.livepoint import empty_structured_array  # noqa: E402
from nessai.model import Model  # noqa: E402
from nessai.proposal import FlowProposal  # noqa: E402

NAMES = ["a", "phi", "psi", "c", "tau"]
BOUNDS = {"a": [-3.0, 7.0], "phi": [0.0, 2.0 * np.pi], "psi": [-np.pi / 2.0, np.pi / 2.0], "c": [0.5, 2.5],
          "tau": [0.0, 4.0]}
# one entry per parameter, in model order (the reference then keeps the x' columns in that order)
REPARAMETERISATIONS = {"a": "rescaletobounds", "phi": "angle-2pi", "psi": "angle-pi", "c": "z-score", "tau": "periodic"}


class BoxModel(Model):
    def __init__(self, names=NAMES, bounds=BOUNDS, discrete=()):
        self.names = list(names)
        self.bounds = {n: list(bounds[n]) for n in names}
        self.discrete = tuple(discrete)

    def new_point(self, N=1):
        x = super().new_point(N)
        for p in self.discrete:  # (the proposal checks its rescaling on points of the model: whole numbers here)
            x[p] = np.floor(x[p])
        return x

    def log_prior(self, x):
        log_p = np.log(self.in_bounds(x), dtype="float")
        for lo, hi in self.bounds.values():
            log_p -= np.log(hi - lo)
        return log_p

    def log_likelihood(self, x):
        return -0.5 * np.sum(self.unstructured_view(x) ** 2, axis=-1)


def stack(x, names):
    return np.stack([np.asarray(x[n], dtype=np.float64) for n in names], axis=1)


def main(NAMES=NAMES, BOUNDS=BOUNDS, REPARAMETERISATIONS=REPARAMETERISATIONS, filename="angle_goldens.npz", discrete=(), latent_scale=1.2):
    OUT = {}
    rng_data = np.random.default_rng(20261001)
    n, m = 157, 133
    live = empty_structured_array(n, names=NAMES)
    for p in NAMES:
        lo, hi = BOUNDS[p]
        live[p] = lo + (hi - lo) * rng_data.beta(2.0, 3.0, size=n)
        if p in discrete:  # integer values lo .. hi
            live[p] = np.floor(lo + (hi + 1 - lo) * rng_data.random(n))
    OUT["live"] = stack(live, NAMES)

    rng = np.random.default_rng(170817)
    torch.manual_seed(170817)
    model = BoxModel(NAMES, BOUNDS, discrete)
    model.set_rng(rng)
    with tempfile.TemporaryDirectory() as tmp:
        prop = FlowProposal(model, rng=rng, poolsize=50, drawsize=400, output=tmp, plot=False,
                            flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1,
                                             batch_norm_between_layers=False, linear_transform="permutation"),
                            reparameterisations=REPARAMETERISATIONS)
        prop.initialise()
        primes = list(prop.prime_parameters)
        OUT["prime_parameters"] = np.array(primes)
        OUT["parameters"] = np.array(list(prop.parameters))
        prop.check_state(live.copy())
        # (initialise() has drawn from the generator - verify_rescaling - so the state the radii come from is set here)
        rng.bit_generator.state = np.random.default_rng(4242).bit_generator.state
        x_prime, log_j = prop.rescale(live.copy())
        OUT["x_prime"] = stack(x_prime, primes)
        OUT["log_j"] = np.asarray(log_j, dtype=np.float64)
        xp_values = rng_data.normal(0.0, 1.2, size=(m, len(primes)))
        if latent_scale != 1.2:  # (x' of a Gaussian-CDF column lives in (0, 1))
            xp_values = 0.5 + 0.25 * np.tanh(xp_values)
        OUT["xp_in"] = xp_values
        xp_in = empty_structured_array(m, names=primes)
        for i, pp in enumerate(primes):
            xp_in[pp] = xp_values[:, i]
        x_back, log_j_inv = prop.inverse_rescale(xp_in.copy())
        back_names = [f for f in x_back.dtype.names if f not in ("logP", "logL", "it")]
        OUT["x_back_names"] = np.array(back_names)
        OUT["x_back"] = stack(x_back, back_names)
        OUT["log_j_inv"] = np.asarray(log_j_inv, dtype=np.float64)
        with np.errstate(divide="ignore"):
            OUT["log_prior_back"] = np.asarray(prop.log_prior(x_back), dtype=np.float64)
        # the fused path: backward_pass on fixed latents (x incl. the auxiliary radii, log_q), its weights
        for k, v in prop.flow.model.state_dict().items():
            OUT[f"state/{k}"] = v.detach().cpu().numpy()
        z = rng_data.normal(size=(400, len(primes))).astype(np.float32).astype(np.float64) * latent_scale
        OUT["bp_z"] = z
        x, log_q, z_kept = prop.backward_pass(z.copy(), rescale=True, return_z=True)
        names_x = [f for f in x.dtype.names if f not in ("logP", "logL", "it")]
        OUT["bp_names"] = np.array(names_x)
        OUT["bp_x"] = stack(x, names_x)
        OUT["bp_log_q"] = np.asarray(log_q, dtype=np.float64)
        OUT["bp_z_kept"] = np.asarray(z_kept, dtype=np.float64)
        with np.errstate(divide="ignore"):
            OUT["bp_log_w"] = np.asarray(prop.compute_weights(x, log_q), dtype=np.float64)
        samples = prop.convert_to_samples(x.copy(), plot=False)
        OUT["bp_sample_names"] = np.array([f for f in samples.dtype.names])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), filename)
    np.savez_compressed(path, **OUT)
    print("wrote", path)
    print(OUT["prime_parameters"], OUT["parameters"], OUT["x_back_names"], OUT["bp_names"], OUT["bp_sample_names"],
          OUT["bp_x"].shape)


if __name__ == "__main__":
    main()
