import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nessai_amd.flows import NeuralSplineFlow
dev = "cuda:0"
torch.manual_seed(1451)
flow = NeuralSplineFlow(32, 64, 6, 2); flow.to(dev); flow.eval()
n = 1_000_000
z = torch.randn(n, 32, device=dev)
def t(prec, wpg=0):
    flow.precision = prec; flow.waves_per_group = wpg
    for _ in range(2): flow.inverse(z)
    torch.cuda.synchronize()
    ev = []
    for _ in range(5):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); flow.inverse(z); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"{prec:7s} wpg={wpg}: median {ms[2]:.2f} ms  min {ms[0]:.2f}")
for prec in ("f32", "bf16x3", "f32", "bf16x3"):
    t(prec)
for wpg in (2, 4, 8):
    t("bf16x3", wpg)
import subprocess
print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout[-600:])
