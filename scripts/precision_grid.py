"""split-bf16 vs f32 log_prob error over a grid of NSF shapes (randomised weights, 3 seeds x 3 input scales)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import DEV, make_pair, rel_err
from nessai_amd.flows import NeuralSplineFlow
from oracle import flows as O
print("D  H  blocks  f32  bf16x3   (max over scales 0.5,1,3 and 3 seeds; trained-like weights x1 and x3)")
for D, H, B in [(4, 32, 4), (8, 32, 4), (8, 64, 6), (12, 30, 4), (16, 32, 6), (16, 64, 6), (24, 48, 6), (32, 32, 6), (6, 40, 4), (10, 30, 8)]:
    worst = {"f32": 0.0, "bf16x3": 0.0}
    for seed in (1, 2, 3):
        for wscale in (1.0,):
            ref, flow = make_pair(NeuralSplineFlow, O.NeuralSplineFlow, (D, H, B, 2), {}, seed=seed)
            with torch.no_grad():
                for p in ref.parameters():
                    p.mul_(wscale)
            flow.load_state_dict(ref.state_dict()); flow.to(DEV); flow.eval()
            for scale in (0.5, 1.0, 3.0):
                x = torch.randn(3000, D, generator=torch.Generator().manual_seed(seed)) * scale
                with torch.no_grad():
                    lp_ref = ref.log_prob(x).numpy()
                for prec in ("f32", "bf16x3"):
                    flow.precision = prec
                    lp = flow.log_prob(x.to(DEV)).cpu().numpy()
                    worst[prec] = max(worst[prec], rel_err(lp, lp_ref))
    print(D, H, B, f"{worst['f32']:.2e}", f"{worst['bf16x3']:.2e}")
