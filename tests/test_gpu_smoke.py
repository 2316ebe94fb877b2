"""The driver's round-end entry point must keep working as the package evolves."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke_runs():
    import torch

    import __graft_entry__ as entry

    with torch.enable_grad():  # conftest runs GPU tests under no_grad; smoke() is called without it
        entry.smoke()
