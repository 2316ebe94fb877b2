"""One launch of the config-3 draw kernel (1e6 rows, fixed seed): dump log_q and x to an .npz
(A/B comparison of kernel builds: NESSAI_AMD_LIBRARY=... python scripts/dump_draw.py out.npz)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
prop = bench.build_proposal(0, 1, dev)
prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
n = int(os.environ.get("DUMP_ROWS", bench.DRAWSIZE))
buf = prop._get_buffers(bench.DRAWSIZE, bench.NLIVE)
prepared = prop._prepare_flow()
prop._launch_draw(n, None, 0, prop.fixed_radius, buf["x64"], buf["log_q"], None, prepared=prepared)
torch.cuda.synchronize()
np.savez(sys.argv[1], log_q=buf["log_q"][:n].cpu().numpy(), x=buf["x64"][:n].cpu().numpy())
