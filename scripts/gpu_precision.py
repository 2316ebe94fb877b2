"""Relative log_prob error of each precision mode against the oracle, per configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
torch.set_grad_enabled(False)
from helpers import DEV, make_pair, rel_err, load_c3_fixture
from test_gpu_flows import CONFIGS
from nessai_amd.flows import NeuralSplineFlow
from oracle import flows as O

for name in sorted(CONFIGS):
    P, Oc, args, kw = CONFIGS[name]
    ref, flow = make_pair(P, Oc, args, kw)
    out = []
    for scale in (0.5, 1.0, 3.0):
        x = torch.randn(4000, args[0]) * scale
        with torch.no_grad():
            lp_ref = ref.log_prob(x); xi_ref, ldi_ref = ref.inverse(x)
        for prec in ("f32", "bf16x3", "f16x3"):
            flow.precision = prec
            try:
                lp = flow.log_prob(x.to(DEV)).cpu()
                xi, ldi = flow.inverse(x.to(DEV))
                out.append(f"{prec}@{scale}: lp {rel_err(lp, lp_ref):.1e} x {rel_err(xi.cpu(), xi_ref):.1e}")
            except FloatingPointError:
                out.append(f"{prec}@{scale}: left the f16 range")
    print(f"{name:24s}", " | ".join(out))
state, shift, scale, live = load_c3_fixture()
ref = O.NeuralSplineFlow(32, 64, 6, 2); ref.load_state_dict(state); ref.eval()
flow = NeuralSplineFlow(32, 64, 6, 2); flow.load_state_dict(state); flow.to(DEV); flow.eval()
x = torch.from_numpy(((live - shift) / scale).astype(np.float32))
z = torch.randn(20000, 32)
with torch.no_grad():
    lp_ref = ref.log_prob(x); xi_ref, ld_ref = ref.inverse(z)
for prec in ("f32", "bf16x3", "f16x3"):
    flow.precision = prec
    lp = flow.log_prob(x.to(DEV)).cpu(); xi, ld = flow.inverse(z.to(DEV))
    print("trained C3", prec, f"log_prob rel {rel_err(lp, lp_ref):.2e}  inverse x {rel_err(xi.cpu(), xi_ref):.2e} logdet {rel_err(ld.cpu(), ld_ref):.2e}")
