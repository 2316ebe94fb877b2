"""Generate golden vectors by running the REFERENCE's own code (build container only).

    python -B tests/golden/make_golden.py      # writes tests/golden/reference_goldens.npz

``/root/reference`` is imported with ``glasflow`` replaced by the oracle's
restatement (``oracle/reference_harness.py``): the reference's *assembly and glue*
— ``nessai.flows.RealNVP`` / ``NeuralSplineFlow``, ``FlowModel``, ``FlowProposal``
(backward_pass, populate), ``LatentRadiusTruncation``, the z-score
reparameterisation — run unmodified and their inputs/outputs are stored.  The
fixtures pin (a) the oracle's own assembly/glue restatement (``oracle/flows.py``,
``oracle/proposal.py``) in the CPU suite and (b) the device path in the GPU suite.
Only data is written; no reference source is copied.
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
from nessai.flowmodel import FlowModel  # noqa: E402
from nessai.flows import NeuralSplineFlow, RealNVP  # noqa: E402
from nessai.livepoint import numpy_array_to_live_points  # noqa: E402
from nessai.model import Model  # noqa: E402
from nessai.proposal import FlowProposal  # noqa: E402
from nessai.proposal.flowproposal.truncation import LatentRadiusTruncation  # noqa: E402
from nessai.utils import compute_radius  # noqa: E402

OUT = {}


def put(key, value):
    OUT[key] = np.array(value, copy=True)


def put_state(prefix, state):
    for k, v in state.items():
        put(f"{prefix}/state/{k}", v.detach().cpu().numpy())


class BoxModel(Model):
    """Uniform prior in a box, Rosenbrock likelihood (examples/rosenbrock.py)."""

    def __init__(self, dims, lo=-5.0, hi=5.0):
        self.names = [f"x_{d}" for d in range(dims)]
        self.bounds = {n: [lo, hi] for n in self.names}

    def log_prior(self, x):
        log_p = np.log(self.in_bounds(x), dtype="float")
        for bounds in self.bounds.values():
            log_p -= np.log(bounds[1] - bounds[0])
        return log_p

    def log_likelihood(self, x):
        x = self.unstructured_view(x)
        return -np.sum(100.0 * (x[..., 1:] - x[..., :-1] ** 2.0) ** 2.0 + (1.0 - x[..., :-1]) ** 2.0, axis=-1)


def flows():
    """G0: the reference's flow classes, seeded; inputs and outputs."""
    cases = {
        "nsf_d4": (NeuralSplineFlow, (4, 8, 2, 2), {}),
        "nsf_d6_k10": (NeuralSplineFlow, (6, 16, 3, 1), dict(num_bins=10)),
        "rnvp_d4": (RealNVP, (4, 8, 2, 2), {}),
        "rnvp_d5_mlp": (RealNVP, (5, 16, 3, 2), dict(net="mlp", linear_transform="permutation")),
    }
    for name, (cls, args, kw) in cases.items():
        torch.manual_seed(170817)
        flow = cls(*args, **kw)
        # move batch-norm / LU parameters off their trivial initial values
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for pname, p in flow.named_parameters():
                if "unconstrained" in pname or "entries" in pname or pname.endswith(".bias"):
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
            for bname, b in flow.named_buffers():
                if "running_mean" in bname:
                    b.add_(0.3 * torch.randn(b.shape, generator=g))
                if "running_var" in bname:
                    b.mul_(torch.rand(b.shape, generator=g) + 0.5)
        flow.eval()
        x = torch.randn(256, args[0], generator=g) * 2.0
        with torch.no_grad():
            z, ld = flow.forward(x)
            xi, ldi = flow.inverse(x)
            lp = flow.log_prob(x)
        put_state(f"flows/{name}", flow.state_dict())
        put(f"flows/{name}/x", x.numpy())
        put(f"flows/{name}/z", z.numpy())
        put(f"flows/{name}/logdet", ld.numpy())
        put(f"flows/{name}/x_inv", xi.numpy())
        put(f"flows/{name}/logdet_inv", ldi.numpy())
        put(f"flows/{name}/log_prob", lp.numpy())


def truncation():
    """G1: constant-volume radius and the latent mask."""
    dims = [2, 15, 16, 32, 64]
    put("truncation/dims", dims)
    put("truncation/radius", [compute_radius(d, 0.95) for d in dims])
    rule = LatentRadiusTruncation(radius_mode="fixed", fixed_radius=2.5)
    rule.prepare(None, None)
    z = np.random.default_rng(3).normal(size=(500, 4))
    put("truncation/z", z)
    put("truncation/kept", rule.apply_latent(None, z))


def proposal():
    """G2/G3/G5: FlowProposal.backward_pass and populate with injected draws."""
    dims = 4
    rng = np.random.default_rng(170817)
    torch.manual_seed(170817)
    model = BoxModel(dims)
    model.set_rng(rng)
    with tempfile.TemporaryDirectory() as tmp:
        prop = FlowProposal(
            model, rng=rng, poolsize=100, drawsize=1000, output=tmp, plot=False,
            flow_config=dict(ftype="nsf", n_blocks=2, n_neurons=8, n_layers=2),
        )
        prop.initialise()
        live = numpy_array_to_live_points(rng.uniform(-2.0, 3.0, size=(200, dims)), model.names)
        prop.check_state(live)  # z-score constants from the live points
        put_state("proposal", prop.flow.model.state_dict())
        put("proposal/live_points", np.stack([live[n] for n in model.names], axis=1))

        # G5: z-score forward / inverse
        x_prime, log_j = prop.rescale(live)
        put("proposal/rescaled", np.stack([x_prime[n] for n in prop.prime_parameters], axis=1))
        put("proposal/rescale_log_j", log_j)
        x_back, log_j_inv = prop.inverse_rescale(x_prime)
        put("proposal/inverse_rescaled", np.stack([x_back[n] for n in model.names], axis=1))
        put("proposal/inverse_rescale_log_j", log_j_inv)

        # forward pass
        z_f, lp_f = prop.forward_pass(live)
        put("proposal/forward_z", z_f)
        put("proposal/forward_log_prob", lp_f)

        # G2: backward pass on fixed latents
        z = rng.normal(size=(512, dims)).astype(np.float32).astype(np.float64) * 1.5
        x, log_q, z_kept = prop.backward_pass(z, rescale=True, return_z=True)
        put("proposal/bp_z", z)
        put("proposal/bp_x", np.stack([x[n] for n in model.names], axis=1))
        put("proposal/bp_log_q", log_q)
        put("proposal/bp_z_kept", z_kept)

        # G3: populate with injected latents and uniforms
        for accumulate in (False, True):
            batches_z = [rng.normal(size=(1000, dims)).astype(np.float32).astype(np.float64) for _ in range(60)]
            batches_u = [rng.random(1000) for _ in range(60)]
            state = dict(k=0, used=[])

            def fake_latent(n, _z=batches_z, _s=state):
                out = _z[_s["k"]]
                _s["k"] += 1
                return out

            orig_bp = prop.backward_pass

            def tracking_bp(zt, *a, _s=state, _z=batches_z, **k):
                res = orig_bp(zt, *a, **k)
                zb = _z[_s["k"] - 1]
                lookup = {row.tobytes(): i for i, row in enumerate(zb)}
                _s["used"].append(np.array([lookup[r.tobytes()] for r in res[2]], dtype=np.int64))
                return res

            class FakeRng:
                def __init__(self, real):
                    self.real = real
                    self.calls = 0

                def random(self, n, _s=state, _u=batches_u):
                    if accumulate:
                        idx = np.concatenate(_s["used"])
                        u = np.concatenate([_u[b][ix] for b, ix in enumerate(_s["used"])])
                        assert len(u) == n
                        return u
                    b = len(_s["used"]) - 1
                    return _u[b][_s["used"][b]]

                def permutation(self, n):
                    return np.arange(n)

            prop.sample_latent_distribution = fake_latent
            prop.backward_pass = tracking_bp
            prop.rng = FakeRng(rng)
            prop.accumulate_weights = accumulate
            prop.populate(live[0], n_samples=100, plot=False)
            tag = "acc" if accumulate else "std"
            put(f"populate_{tag}/z", np.stack(batches_z[: state["k"]]))
            put(f"populate_{tag}/u", np.stack(batches_u[: state["k"]]))
            put(f"populate_{tag}/x", np.stack([prop.x[n] for n in model.names], axis=1))
            put(f"populate_{tag}/logP", prop.samples["logP"])
            put(f"populate_{tag}/logL", prop.samples["logL"])
            put(f"populate_{tag}/n_passes", state["k"])
            put(f"populate_{tag}/population_acceptance", prop.population_acceptance)
            put(f"populate_{tag}/radius", prop.r if hasattr(prop, "r") and prop.r is not None else np.nan)
            del prop.sample_latent_distribution, prop.backward_pass
            prop.rng = rng
        put("proposal/radius", prop._get_latent_radius_rule().fixed_radius)


def flowmodel():
    """G4: batch-size table, prep_data split, one optimiser step."""
    table = [(4010, 1000), (106, 21), (1000, 1000), (2000, 1000), (1800, 1000), (1005, 1000), (25, 10)]
    put("flowmodel/batch_table", table)
    put("flowmodel/batch_out", [FlowModel.check_batch_size(np.arange(n), b) for n, b in table])
    with tempfile.TemporaryDirectory() as tmp:
        torch.manual_seed(170817)
        rng = np.random.default_rng(170817)
        fm = FlowModel(
            flow_config=dict(n_inputs=4, n_neurons=8, n_blocks=2, n_layers=2, ftype="nsf"),
            training_config=dict(batch_size=100, val_size=0.1, max_epochs=1),
            output=tmp, rng=rng,
        )
        fm.initialise()
        put_state("flowmodel/before", fm.model.state_dict())
        data = np.random.default_rng(5).normal(size=(250, 4))
        train_data, val_data, bs = fm.prep_data(data, val_size=0.1, batch_size=100)
        put("flowmodel/data", data)
        put("flowmodel/train_data", train_data.numpy())
        put("flowmodel/val_data", val_data.numpy())
        put("flowmodel/batch_size", bs)
        # one full-batch step (deterministic: no randperm dependence on batch order)
        x = train_data[:100]
        fm.model.train()
        for p in fm.model.parameters():
            p.grad = None
        loss = -fm.model.log_prob(x).mean()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in fm.model.named_parameters()}
        total_norm = torch.nn.utils.clip_grad_norm_(fm.model.parameters(), 5.0)
        fm._optimiser.step()
        put("flowmodel/step_x", x.numpy())
        put("flowmodel/step_loss", loss.item())
        put("flowmodel/step_grad_norm", float(total_norm))
        for k, g in grads.items():
            put(f"flowmodel/grad/{k}", g.numpy())
        put_state("flowmodel/after", fm.model.state_dict())
        fm.model.eval()
        with torch.no_grad():
            put("flowmodel/val_loss_after", (-fm.model.log_prob(val_data).mean()).item())


def main():
    flows()
    truncation()
    proposal()
    flowmodel()
    path = os.path.join(ROOT, "tests", "golden", "reference_goldens.npz")
    np.savez_compressed(path, **OUT)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(OUT), "arrays")


if __name__ == "__main__":
    main()
