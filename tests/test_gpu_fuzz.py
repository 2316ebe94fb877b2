"""Randomised parity sweeps (scripts/fuzz_*.py) at a size that fits the GPU suite: random supported
configurations instead of the hand-picked ones of the other files, fixed seeds."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script, iterations, seed", [
    ("fuzz_parity.py", 30, 2),     # flows: forward / inverse / log_prob, float64 arbiter
    ("fuzz_populate.py", 30, 4),   # fused populate pass, proposal precision f32 / auto
    ("fuzz_reparam.py", 25, 6),    # reparameterisation table of the draw epilogue
    ("fuzz_train.py", 40, 5),      # loss and gradients of one minibatch
])
def test_randomised_sweep(script, iterations, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), "--iterations", str(iterations),
                        "--seed", str(seed)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
