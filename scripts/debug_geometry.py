import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_c3_fixture
from test_gpu_populate import build
state, shift, scale, _ = load_c3_fixture()
n = 300_000
with torch.no_grad():
    prop, _ = build(32, "nsf", dict(n_neurons=64, n_blocks=6), poolsize=2000, drawsize=n, state=state, shift=shift, scale=scale)
    prop.populate(None, n_samples=50, max_samples=10**9)
    radius = prop.fixed_radius if prop.fixed_radius is not None else 1.0e9
    buf = prop._get_buffers(n, 2000)
    outs = {}
    for occ in ("3", "2", "2x2"):
        os.environ["NESSAI_AMD_OCC"] = occ
        for prio in ("0", "1"):
            os.environ["NESSAI_AMD_PRIO"] = prio
            buf["x64"].zero_(); buf["log_q"].zero_()
            prop._launch_draw(n, None, 0, radius, buf["x64"], buf["log_q"], None)
            torch.cuda.synchronize()
            outs[(occ, prio)] = (buf["x64"][:n].clone(), buf["log_q"][:n].clone())
    x3, q3 = outs[("3", "0")]
    for k, (x, q) in outs.items():
        dx = (x - x3).abs().max().item()
        a, b = torch.nan_to_num(q, nan=12345.0), torch.nan_to_num(q3, nan=12345.0)
        bad = torch.nonzero(a != b).reshape(-1)
        print(k, "max|dx|", dx, "rows with different log_q", bad.numel(), bad[:8].tolist(),
              [(float(a[i]), float(b[i])) for i in bad[:4]], "precision", prop.flow.model.effective_precision)
