"""Signed error of the inverse-flow log-det (the populate direction) against the oracle evaluated
in float64, on the trained config-3 flow: mean (bias), rms and max, per precision mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import DEV, load_c3_fixture
from nessai_amd.flows import NeuralSplineFlow
from oracle import flows as O

state, shift, scale, live = load_c3_fixture()
ref = O.NeuralSplineFlow(32, 64, 6, 2); ref.load_state_dict(state); ref.eval(); ref.double()
flow = NeuralSplineFlow(32, 64, 6, 2); flow.load_state_dict(state); flow.to(DEV); flow.eval()
torch.manual_seed(3)
z = torch.randn(50000, 32)
x = torch.from_numpy(((live - shift) / scale).astype(np.float32))
with torch.no_grad():
    xi_ref, ld_ref = ref.inverse(z.double())
    lp_ref = ref.log_prob(x.double())
    for prec in ("f32", "bf16x3"):
        flow.precision = prec
        xi, ld = flow.inverse(z.to(DEV))
        e = ld.cpu().double() - ld_ref
        ok = torch.isfinite(e)
        lp = flow.log_prob(x.to(DEV)).cpu().double()
        e2 = lp - lp_ref
        print(f"{prec}: inverse logdet err mean {e[ok].mean():+.3e} rms {e[ok].pow(2).mean().sqrt():.3e} max {e[ok].abs().max():.3e}"
              f" | log_prob err mean {e2.mean():+.3e} rms {e2.pow(2).mean().sqrt():.3e} max {e2.abs().max():.3e}"
              f" | x err max {(xi.cpu().double() - xi_ref)[ok].abs().max():.3e}")
