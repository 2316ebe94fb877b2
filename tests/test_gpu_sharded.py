"""Sharded populate end to end: two ranks (one process each) against one rank.

The draw kernel's Philox streams are indexed by the GLOBAL row, so two ranks that each take half
of every draw batch must reproduce the single-rank run exactly: same accepted rows in the same
order, same counts.  The GPU box has one device, so both ranks share it and the collectives run
on the ``gloo`` backend (the product code is backend-agnostic; ``bench.py`` uses RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRAW, NS = 40_000, 300


def _proposal(world, shard, **extra):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_c3_fixture
    from nessai_amd.model import RosenbrockModel
    from nessai_amd.proposal import FlowProposal

    state, shift, scale, _ = load_c3_fixture()
    prop = FlowProposal(
        RosenbrockModel(32), rng=np.random.default_rng(3),
        flow_config=dict(ftype="nsf", n_neurons=64, n_blocks=6, n_layers=2, num_bins=8),
        poolsize=NS, drawsize=DRAW, device_seed=77, shard=shard, volume_fraction=0.5, **extra,
    )
    prop.initialise()
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    prop.set_rescaling_constants(shift, scale)
    return prop


def _worker(rank, world, port, out_dir, extra=None, calls=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    prop = _proposal(world, shard=True, **(extra or {}))
    for _ in range(calls):  # the second call runs its trains from the acceptance the first one saw
        prop.populate(None, n_samples=NS, max_samples=10**9)
    x = np.stack([prop.x[n] for n in prop.names], axis=1)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    with open(os.path.join(out_dir, f"c{rank}.txt"), "w") as f:
        f.write(f"{prop.n_accepted} {prop.n_proposed}")
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_two_ranks_reproduce_one_rank(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    single = _proposal(1, shard=False)
    single.populate(None, n_samples=NS, max_samples=10**9)
    expected = np.stack([single.x[n] for n in single.names], axis=1)
    xs = [np.load(tmp_path / f"x{r}.npy") for r in range(2)]
    assert np.array_equal(xs[0], xs[1])  # both ranks end with the same pool
    counts = [(tmp_path / f"c{r}.txt").read_text().split() for r in range(2)]
    assert counts[0] == counts[1] == [str(single.n_accepted), str(single.n_proposed)]
    assert np.array_equal(xs[0], expected)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra", [dict(accumulate_weights=True),
                                   dict(enforce_likelihood_threshold=True)],
                         ids=["accumulate_weights", "rules_after_the_flow"])
def test_two_ranks_reproduce_one_rank_in_the_other_modes(tmp_path, extra):
    """``accumulate_weights=True`` sharded (the running log_constant and the expected count exchanged per
    train of passes; flowproposal.py:471-490) and the truncation rules that act
    after the flow on the device-side loop (truncation.py:385-392, 423-430): two ranks end with the pool
    of one rank, bit for bit, over two consecutive populations."""
    worst = None
    if "enforce_likelihood_threshold" in extra:
        probe = _proposal(1, shard=False)
        probe.populate(None, n_samples=NS, max_samples=10**9)
        worst = probe.samples[np.argsort(probe.samples["logL"])[NS // 4]]  # a quarter of the pool lies below it

    def run(prop):
        for _ in range(2):
            prop.populate(worst, n_samples=NS, max_samples=10**9)

    mp.spawn(_mode_worker, args=(2, _free_port(), str(tmp_path), extra, worst), nprocs=2, join=True)
    single = _proposal(1, shard=False, **extra)
    run(single)
    expected = np.stack([single.x[n] for n in single.names], axis=1)
    xs = [np.load(tmp_path / f"x{r}.npy") for r in range(2)]
    assert np.array_equal(xs[0], xs[1])
    counts = [(tmp_path / f"c{r}.txt").read_text().split() for r in range(2)]
    assert counts[0] == counts[1] == [str(single.n_accepted), str(single.n_proposed)]
    assert len(expected) == NS and np.array_equal(xs[0], expected)
    if worst is not None:  # every row of the pool passed the likelihood rule
        assert (single.samples["logL"] > worst["logL"]).all()


def _mode_worker(rank, world, port, out_dir, extra, worst):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    prop = _proposal(world, shard=True, **extra)
    for _ in range(2):
        prop.populate(worst, n_samples=NS, max_samples=10**9)
    x = np.stack([prop.x[n] for n in prop.names], axis=1)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    with open(os.path.join(out_dir, f"c{rank}.txt"), "w") as f:
        f.write(f"{prop.n_accepted} {prop.n_proposed}")
    dist.destroy_process_group()


def _train_worker(rank, world, port, out_dir, replicate):
    """Rank 0 fits the live points (``replicate``: both ranks do - the fused trainer is not
    bit-reproducible), then populate sharded: the flow, the pool and the host generators must agree
    afterwards."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sys.path.insert(0, ROOT)
    from nessai_amd.model import GaussianModel
    from nessai_amd.proposal import FlowProposal

    torch.manual_seed(11)
    model = GaussianModel(6)
    prop = FlowProposal(
        model, rng=np.random.default_rng(5),
        flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2, n_layers=1, num_bins=4),
        training_config=dict(max_epochs=40, patience=40, batch_size=250),
        poolsize=200, drawsize=20_000, device_seed=9, shard=True, replicate_training=replicate,
        output=os.path.join(out_dir, "run"),
    )
    prop.initialise()
    live = model.new_point(500, rng=np.random.default_rng(1))
    if rank == 1:  # desynchronise the replicas on purpose: rank 1 starts from other weights
        with torch.no_grad():
            for p_ in prop.flow.model.parameters():
                p_.add_(0.01 * torch.randn_like(p_))
    prop.train(live)
    assert prop.training_count == 1 and not prop.flow.model.training
    flat = prop.flow.model.trainer().flat.detach().cpu().numpy()
    prop.populate(live[0], n_samples=200, max_samples=10**8)
    x = np.stack([prop.x[n] for n in prop.names], axis=1)
    np.savez(os.path.join(out_dir, f"t{rank}.npz"), flat=flat, x=x, indices=np.asarray(prop.indices),
             rng=prop.rng.random(4), torch_rng=torch.rand(4).numpy(), weights_file=prop.flow.weights_file)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("replicate", [False, True])
def test_sharded_train_leaves_one_flow_on_every_rank(tmp_path, replicate):
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path), replicate), nprocs=2, join=True)
    a, b = (np.load(tmp_path / f"t{r}.npz") for r in range(2))
    assert np.array_equal(a["flat"], b["flat"])          # rank 0's weights everywhere
    assert np.array_equal(a["x"], b["x"]) and len(a["x"]) == 200
    assert np.array_equal(a["indices"], b["indices"])    # the replicated samplers stay in lock step
    assert np.array_equal(a["rng"], b["rng"]) and np.array_equal(a["torch_rng"], b["torch_rng"])
    assert str(a["weights_file"]) != str(b["weights_file"])  # one weights file per rank
    for r, d in enumerate((a, b)):
        state = torch.load(str(d["weights_file"]), map_location="cpu", weights_only=True)
        assert all(torch.isfinite(v).all() for v in state.values() if v.is_floating_point()), r


def _gauss_proposal(shard, lean=True):
    """Default-style sizes: ``drawsize == poolsize`` (the reference's default), big enough that a rank's share of a
    pass (20000 rows) takes the lean trains - a rank then routinely accepts MORE than drawsize / world rows, which the
    replay's per-row outputs must hold (round-5 advisor finding)."""
    sys.path.insert(0, ROOT)
    from nessai_amd.model import GaussianModel
    from nessai_amd.proposal import FlowProposal

    torch.manual_seed(0)
    prop = FlowProposal(GaussianModel(4), rng=np.random.default_rng(3), poolsize=40_000, drawsize=40_000,
                        flow_config=dict(ftype="realnvp", n_neurons=8, n_blocks=2), device_seed=5, shard=shard,
                        lean=lean)
    prop.initialise()
    prop.check_state(GaussianModel(4).new_point(300, rng=np.random.default_rng(1)))
    return prop


def _gauss_worker(rank, world, port, out_dir, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    prop = _gauss_proposal(shard=True)
    stats = []
    for _ in range(2):
        prop.populate(None, n_samples=40_000, max_samples=10**9)
        assert prop.last_populate_path == "trains_sharded" and prop.last_populate_lean
        stats.append(dict(prop.sharded_stats))
    x = np.stack([prop.x[n] for n in prop.names], axis=1)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), x)
    with open(os.path.join(out_dir, f"c{rank}.txt"), "w") as f:
        f.write(f"{prop.n_accepted} {prop.n_proposed}")
    import json

    with open(os.path.join(out_dir, f"s{rank}.json"), "w") as f:
        json.dump(stats, f)
    if backend == "nccl":
        torch.cuda.synchronize()
    dist.destroy_process_group()


def _gauss_single():
    single = _gauss_proposal(shard=False)
    for _ in range(2):
        single.populate(None, n_samples=40_000, max_samples=10**9)
    assert single.last_populate_path == "trains"
    return single, np.stack([single.x[n] for n in single.names], axis=1)


@pytest.mark.timeout(600)
def test_two_ranks_with_drawsize_equal_to_poolsize(tmp_path):
    """``drawsize == poolsize`` sharded over two ranks: each rank draws 20000 rows per pass and replays up to 40000
    accepted ones - the pool equals the one-rank pool bit for bit."""
    mp.spawn(_gauss_worker, args=(2, _free_port(), str(tmp_path), "gloo"), nprocs=2, join=True)
    single, expected = _gauss_single()
    xs = [np.load(tmp_path / f"x{r}.npy") for r in range(2)]
    assert np.array_equal(xs[0], xs[1]) and xs[0].shape == (40_000, 4)
    counts = [(tmp_path / f"c{r}.txt").read_text().split() for r in range(2)]
    assert counts[0] == counts[1] == [str(single.n_accepted), str(single.n_proposed)]
    assert np.array_equal(xs[0], expected)


@pytest.mark.timeout(600)
def test_rccl_one_rank_sharded_populate_equals_unsharded(tmp_path):
    """The RCCL path under the driver every round: ONE rank on the ``nccl`` backend (= RCCL on ROCm; what
    ``NESSAI_AMD_FORCE_SHARD=1`` does in ``bench.py``) runs the sharded trains - heads, all-reduce(max) of the
    maxima on the device, tails, all-reduce(sum) of the counts, commit - and must end with the pool of the unsharded
    loop; ``sharded_stats`` counts two all-reduces per train (the two all-gathers exist from world 2 on)."""
    import json

    mp.spawn(_gauss_worker, args=(1, _free_port(), str(tmp_path), "nccl"), nprocs=1, join=True)
    single, expected = _gauss_single()
    x = np.load(tmp_path / "x0.npy")
    assert (tmp_path / "c0.txt").read_text().split() == [str(single.n_accepted), str(single.n_proposed)]
    assert np.array_equal(x, expected)
    for st in json.loads((tmp_path / "s0.json").read_text()):
        assert st["trains"] >= 1 and st["all_reduce"] == 2 * st["trains"]
        assert sum(st["train_lengths"]) >= st["passes"] >= 1
        assert st["all_gather"] == 0  # world 1: nothing to gather; 2 from world 2 on (test_distributed_gloo.py)
