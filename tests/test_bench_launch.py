"""``python bench.py --gpus N`` without a launcher (the command the round-end driver runs on an 8-GPU node):
``bench.self_launch`` re-runs the command line as N ranks under ``torch.distributed.run`` on 127.0.0.1 and
forwards rank 0's ONE JSON line.  No multi-GPU node was ever available to a round, so the code path is pinned
here with the device count and the child process mocked: argument forwarding, the rendezvous address, the
IPC environment, the one-JSON-line rule and the "needs N devices" exit."""
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def launch(monkeypatch, devices, stdout, returncode=0, argv=("--gpus", "8", "--steps", "3", "--warmup", "1")):
    import subprocess

    seen = {}

    def fake_run(cmd, env=None, stdout=None, **kw):
        seen.update(cmd=list(cmd), env=dict(env), stdout=stdout)
        return types.SimpleNamespace(returncode=returncode, stdout=seen["payload"])

    seen["payload"] = stdout
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: devices)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    r, w = os.pipe()
    try:
        bench.self_launch(8, w)
        os.close(w)
        w = None
        return seen, os.read(r, 1 << 16).decode()
    finally:
        if w is not None:
            os.close(w)
        os.close(r)


def test_self_launch_runs_n_ranks_and_forwards_rank_zero_s_line(monkeypatch):
    line = json.dumps({"metric": "accepted live points/sec", "value": 4.0e5, "n_gpus": 8})
    seen, out = launch(monkeypatch, 8, ("library chatter\n" + line + "\n").encode())
    assert out == line + "\n" and json.loads(out)["n_gpus"] == 8
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"       # the container's hostname may not resolve
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]  # the ranks see the same arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # dmabuf IPC: RCCL fails without it here


def test_self_launch_needs_n_devices_on_the_node(monkeypatch):
    with pytest.raises(SystemExit, match="needs 8 devices on this node, found 2"):
        launch(monkeypatch, 2, b"")


@pytest.mark.parametrize("payload, code", [(b'{"a": 1}\n{"b": 2}\n', 0), (b"", 0), (b'{"a": 1}\n', 3)])
def test_self_launch_insists_on_one_json_line_and_a_clean_exit(monkeypatch, payload, code):
    with pytest.raises(SystemExit, match="8-rank run failed"):
        launch(monkeypatch, 8, payload, returncode=code)


def test_without_a_device_there_is_no_fallback(monkeypatch):
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    with pytest.raises(SystemExit, match="no CPU fallback"):
        bench.self_launch(2, 1)
