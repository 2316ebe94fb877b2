import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from helpers import DEV, make_pair
from test_gpu_flows import CONFIGS
for name in sorted(CONFIGS):
    P, Oc, args, kw = CONFIGS[name]
    ref, flow = make_pair(P, Oc, args, kw)
    for prec in ("f32", "bf16x3"):
        flow.precision = prec
        out = []
        for scale in (0.5, 1.0, 3.0):
            torch.manual_seed(1)
            x = torch.randn(4000, args[0]) * scale
            with torch.no_grad():
                b = ref.log_prob(x).double().numpy()
                a = flow.log_prob(x.to(DEV)).cpu().double().numpy()
            d = np.abs(a - b)
            big = np.abs(b) >= 0.1
            r = d[big] / np.abs(b[big])
            out.append(f"s{scale}: floor1 {np.max(d/np.maximum(np.abs(b),1)):.1e} true p99 {np.quantile(r,0.99):.1e} max {r.max():.1e} minabs {np.abs(b).min():.2f}")
        print(f"{name:22s} {prec:6s}", " | ".join(out))
