"""First-contact GPU check: parity of the fused flow kernel vs the oracle + rough timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nessai_amd.flows import NeuralSplineFlow, RealNVP
from oracle import flows as O

dev = "cuda:0"
print("device", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)

def randomise(flow):
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if 'unconstrained' in name or 'lower_entries' in name or 'upper_entries' in name or name.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
        for name, b in flow.named_buffers():
            if 'running_mean' in name: b.add_(0.3 * torch.randn_like(b))
            if 'running_var' in name: b.mul_(torch.rand_like(b) + 0.5)

def check(tag, P, Oc, args, kw, n=5000, scale=2.0):
    torch.manual_seed(5)
    ref = Oc(*args, **kw); randomise(ref); ref.eval()
    flow = P(*args, **kw); flow.load_state_dict(ref.state_dict()); flow.to(dev); flow.eval()
    D = args[0]
    x = torch.randn(n, D) * scale
    with torch.no_grad():
        z_r, ld_r = ref.forward(x); xi_r, ldi_r = ref.inverse(x); lp_r = ref.log_prob(x)
    xd = x.to(dev)
    try:
        z, ld = flow.forward(xd); xi, ldi = flow.inverse(xd); lp = flow.log_prob(xd)
        z2, lp2 = flow.forward_and_log_prob(xd)
        torch.cuda.synchronize()
    except Exception as e:
        print(tag, "FAILED", repr(e)); return
    e = lambda a, b: (a.cpu() - b).abs().max().item()
    rel = ((lp.cpu() - lp_r).abs() / lp_r.abs().clamp_min(1e-3)).max().item()
    print(f"{tag:28s} fwd z {e(z,z_r):.2e} ld {e(ld,ld_r):.2e} | inv x {e(xi,xi_r):.2e} ld {e(ldi,ldi_r):.2e} | log_prob abs {e(lp,lp_r):.2e} rel {rel:.2e} | fwd+lp {e(z2,z_r):.2e} {e(lp2,lp_r):.2e}")

check("nsf D4 H8", NeuralSplineFlow, O.NeuralSplineFlow, (4, 8, 2, 2), {})
check("nsf D2 H8", NeuralSplineFlow, O.NeuralSplineFlow, (2, 8, 2, 2), {})
check("nsf D5 H40 K10 bn", NeuralSplineFlow, O.NeuralSplineFlow, (5, 40, 2, 1), dict(num_bins=10, batch_norm_between_layers=True))
check("nsf D32 H64 C3", NeuralSplineFlow, O.NeuralSplineFlow, (32, 64, 6, 2), {}, n=20000)
check("nsf D15 H30", NeuralSplineFlow, O.NeuralSplineFlow, (15, 30, 4, 2), {})
check("nsf D64 H128", NeuralSplineFlow, O.NeuralSplineFlow, (64, 128, 3, 2), {}, n=3000)
check("nsf D8 H16 K4 lu tanh", NeuralSplineFlow, O.NeuralSplineFlow, (8, 16, 2, 2), dict(num_bins=4, linear_transform='lu', activation=torch.tanh))
check("rnvp D16 H32 C2 resnet", RealNVP, O.RealNVP, (16, 32, 4, 2), {}, n=20000)
check("rnvp D16 H32 mlp", RealNVP, O.RealNVP, (16, 32, 4, 2), dict(net='mlp'))
check("rnvp D6 additive actnorm", RealNVP, O.RealNVP, (6, 16, 2, 2), dict(use_volume_preserving=True, actnorm=True, batch_norm_between_layers=False, linear_transform=None))
check("rnvp D7 H33 silu perm", RealNVP, O.RealNVP, (7, 33, 3, 1), dict(linear_transform='permutation', activation=torch.nn.functional.silu))

# timing: C3 inverse, 1e6 rows
torch.manual_seed(1451)
flow = NeuralSplineFlow(32, 64, 6, 2); flow.to(dev); flow.eval()
n = 1_000_000
z = torch.randn(n, 32, device=dev)
for wpg in (4, 8, 2):
    flow.waves_per_group = wpg
    for _ in range(2): flow.inverse(z)
    torch.cuda.synchronize()
    t = time.time(); reps = 5
    for _ in range(reps): x, ld = flow.inverse(z)
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    print(f"C3 inverse 1e6 rows, waves/group={wpg}: {dt*1e3:.2f} ms  -> {n/dt/1e6:.1f} Mrows/s, {n*491520/dt/1e12:.1f} TFLOP/s")
