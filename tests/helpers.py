"""Shared test helpers (CPU and GPU suites)."""
import numpy as np
import torch

DEV = "cuda:0"


def randomise(flow, seed=0):
    """Move LU / batch-norm / bias parameters off their trivial initial values."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if (
                "unconstrained" in name
                or "lower_entries" in name
                or "upper_entries" in name
                or name.endswith(".bias")
                or "log_scale" in name
                or name.endswith(".shift")
            ):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif "batch_norm_layers" in name and name.endswith(".weight"):  # batch norm within the residual blocks
                p.mul_(0.5 + torch.rand(p.shape, generator=g))
            elif "q_vectors" in name:  # Householder vectors of an SVD-linear layer: off their identity pairs
                p.add_(0.3 * torch.randn(p.shape, generator=g))
        for name, b in flow.named_buffers():
            if "running_mean" in name:
                b.add_(0.3 * torch.randn(b.shape, generator=g))
            if "running_var" in name:
                b.mul_(torch.rand(b.shape, generator=g) + 0.5)


def make_pair(product_cls, oracle_cls, args, kwargs=None, seed=5):
    """(oracle flow on CPU, product flow on the GPU) with identical weights."""
    kwargs = dict(kwargs or {})
    torch.manual_seed(seed)
    ref = oracle_cls(*args, **kwargs)
    randomise(ref, seed)
    ref.eval()
    flow = product_cls(*args, **kwargs)
    flow.load_state_dict(ref.state_dict())
    flow.to(DEV)
    flow.eval()
    return ref, flow


def rel_err(a, b, floor=1.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def load_c3_fixture():
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_trained.npz")
    data = np.load(path)
    state = {k[len("state/"):]: torch.from_numpy(data[k]) for k in data.files if k.startswith("state/")}
    return state, data["shift"], data["scale"], data["live_points"]
