import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (ROCm device)")
    # the suites need the in-tree HIP library (hipcc cross-compiles without a GPU)
    lib_path = os.path.join(ROOT, "nessai_amd", "libnessai_amd.so")
    if not os.path.exists(lib_path):
        import __graft_entry__

        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


#: GPU suites that drive the flows under autograd; every other GPU suite exercises the inference
#: kernels and runs with autograd off, as the reference's callers do
#: (``torch.inference_mode()`` in src/nessai/flowmodel/base.py:782-948)
GRAD_MODULES = {"test_gpu_autograd", "test_gpu_train", "test_gpu_conditional_train", "test_gpu_train_modes"}


@pytest.fixture(autouse=True)
def _seed(request):
    import numpy as np
    import torch

    torch.manual_seed(170817)
    np.random.seed(170817)
    if "gpu" in request.keywords and request.module.__name__ not in GRAD_MODULES:
        with torch.no_grad():
            yield
    else:
        yield
