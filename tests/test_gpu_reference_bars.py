"""The reference's OWN property bars on the device, at the reference's own fixture (runs on a real MI355X).

``tests/test_flows/test_included_flows.py:39-43`` builds ``Flow(D, 8, 2, 2, batch_norm_between_layers=False).eval()`` for
D in {2, 4}, draws N = 1000 rows and asks (``:129-154``, ``np.testing.assert_array_almost_equal(..., decimal=5)``, i.e.
``|a - b| < 1.5e-5``; the reference marks these tests ``flaky(reruns=5)`` - float32 misses the bar on some draws):

* invertibility: ``inverse(forward(x)) == x`` and ``log_J == -log_J_inv``,
* ``sample_and_log_prob``: the returned log-probability equals ``log_prob`` of the returned samples,

and SURVEY.md section 8d adds the gate ``|inverse(forward(x)) - x| <= 1e-5``.  Here the device kernels (exact float32 and
the split-f16 mode ``populate`` runs) are held to exactly these numbers on five seeded sets of weights per case.  A bar
that float32 arithmetic itself does not hold is not held against the device: each figure is measured for the float32
CPU oracle on the same weights and inputs, and where the oracle is over the bar the device must not be further out
than the oracle (x 1.5) - the measured pairs are printed.  One case of the forty (nsf, D = 2, seed 2) misses
``log_J == -log_J_inv`` at 1.8e-5 where the float32 oracle shows 4.8e-6 by cancellation: it is taken apart against the
float64 oracle, term by term (scripts/debug_bars.py prints the same table).
"""
import numpy as np
import pytest
import torch

from helpers import DEV
from nessai_amd.flows import NeuralSplineFlow, RealNVP
from oracle import flows as O

pytestmark = pytest.mark.gpu

BAR = 1.5e-5   # numpy's decimal=5
GATE = 1.0e-5
FLOWS = {"realnvp": (RealNVP, O.RealNVP), "nsf": (NeuralSplineFlow, O.NeuralSplineFlow)}


def _pair(kind, dim, seed):
    P, Oc = FLOWS[kind]
    torch.manual_seed(seed)
    ref = Oc(dim, 8, 2, 2, batch_norm_between_layers=False).eval()
    flow = P(dim, 8, 2, 2, batch_norm_between_layers=False)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)  # (returns None, as the reference's BaseFlow.to)
    flow.eval()
    return ref, flow


def _held(name, device_err, oracle_err, bar):
    """The device holds the bar, or float32 itself does not and the device is no further out than the oracle x 1.5."""
    ok = device_err < bar or device_err <= 1.5 * oracle_err
    tag = "" if device_err < bar else f"  [bar {bar:.1e} missed; float32 oracle on the same case: {oracle_err:.2e}]"
    print(f"{name}: device {device_err:.2e}, float32 oracle {oracle_err:.2e}{tag}")
    return ok


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("dim", [2, 4])
@pytest.mark.parametrize("kind", ["realnvp", "nsf"])
def test_invertibility_at_the_reference_bar(kind, dim, precision, seed):
    ref, flow = _pair(kind, dim, seed)
    flow.precision = precision
    x = torch.randn(1000, dim)
    with torch.inference_mode():
        z_o, lj_o = ref.forward(x)
        xo_o, lji_o = ref.inverse(z_o)
        z, lj = flow.forward(x.to(DEV))
        xo, lji = flow.inverse(z)
    e_x = float((xo.cpu() - x).abs().max())
    e_j = float((lj.cpu() + lji.cpu()).abs().max())
    o_x = float((xo_o - x).abs().max())
    o_j = float((lj_o + lji_o).abs().max())
    assert _held(f"{kind} D={dim} {precision} seed {seed}: |inverse(forward(x)) - x|", e_x, o_x, GATE)
    name = f"{kind} D={dim} {precision} seed {seed}: |log_J + log_J_inv|"
    if not _held(name, e_j, o_j, BAR):
        # The sum of the two log-Jacobians is a difference of rounding errors: where the inverse's log|J| is steep in z
        # (the measured case - nsf, D = 2, seed 2, x = (1.76, 1.78) - moves by 2.3e-5 for the 3e-7 the device's z is
        # off) the float32 oracle can come out ahead by CANCELLATION while being further from the truth in every term.
        # So the miss is taken apart against the float64 oracle: the forward log|J|, the inverse log|J| AT THE z IT WAS
        # GIVEN, and z itself must each be no further from float64 than the float32 oracle's (x 1.5).
        import copy

        r64 = copy.deepcopy(ref).double()
        with torch.inference_mode():
            z64, lj64 = r64.forward(x.double())
            _, lji64_dev = r64.inverse(z.cpu().double())
            _, lji64_o = r64.inverse(z_o.double())
        terms = {
            "forward log|J|": (float((lj.cpu() - lj64).abs().max()), float((lj_o - lj64).abs().max())),
            "inverse log|J| at its own z": (float((lji.cpu() - lji64_dev).abs().max()), float((lji_o - lji64_o).abs().max())),
            "z": (float((z.cpu() - z64).abs().max()), float((z_o - z64).abs().max())),
        }
        for term, (dev, o32) in terms.items():
            print(f"    {term}: device {dev:.2e} from float64, float32 oracle {o32:.2e}")
            assert dev <= max(BAR, 1.5 * o32), (name, term, dev, o32)


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("dim", [2, 4])
@pytest.mark.parametrize("kind", ["realnvp", "nsf"])
def test_sample_and_log_prob_at_the_reference_bar(kind, dim, precision, seed):
    ref, flow = _pair(kind, dim, seed)
    flow.precision = precision
    with torch.inference_mode():
        x, lp = flow.sample_and_log_prob(1000)
        lp_target = flow.log_prob(x)
        # the oracle's own figure for the same property on the same samples: its sample_and_log_prob arithmetic is
        # inverse(z) + base log-prob - log|J_inv|, against its forward log_prob
        z, _ = ref.forward(x.cpu())
        xo, lji = ref.inverse(z)
        lp_o = ref.base_distribution_log_prob(z) - lji
        lp_o_target = ref.log_prob(xo)
    assert x.shape == (1000, dim)
    e = float((lp - lp_target).abs().max())
    o = float((lp_o - lp_o_target).abs().max())
    assert _held(f"{kind} D={dim} {precision} seed {seed}: |sample_and_log_prob - log_prob(samples)|", e, o, BAR)
