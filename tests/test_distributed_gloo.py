"""The multi-rank exchange of the sharded populate, world_size 2 on CPU (gloo).

The draw batch is split over ranks; the exchange is one all-reduce(max) of the
encoded ``max(log_w)`` and one all-gather of the accepted rows.
The per-rank compute needs the GPU, so here the rows come from the numpy Philox
restatement and the oracle; what is exercised is the collective logic of
``nessai_amd.proposal`` itself.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nessai_amd import lib
    from nessai_amd.proposal import _gather_accepted
    from oracle import philox

    L = lib.load()
    dims, n_local, capacity = 5, 4000, 50
    # every rank owns the rows [rank*n_local, (rank+1)*n_local) of one global batch
    z = philox.latents(7, rank * n_local, n_local, dims).astype(np.float64)
    log_w = -0.5 * np.sum(z**2, axis=1) * 0.3
    # (1) global maximum through the monotone int64 key
    key = torch.tensor([L.nsa_max_key_encode(float(log_w.max()))], dtype=torch.int64)
    dist.all_reduce(key, op=dist.ReduceOp.MAX)
    gmax = L.nsa_max_key_decode(int(key.item()))
    # (2) local accept against the GLOBAL maximum with the global-row-indexed uniforms
    u = philox.uniforms(7, rank * n_local, n_local)
    accept = (log_w - gmax) > np.log(u)
    x_acc = torch.zeros((capacity, dims), dtype=torch.float64)  # the rank's compaction buffer
    rows = z[accept][:capacity]
    x_acc[: len(rows)] = torch.from_numpy(rows.copy())
    n_acc = torch.tensor([int(accept.sum())], dtype=torch.int64)  # may exceed the capacity
    total, gathered = _gather_accepted(x_acc, n_acc, capacity, dims)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), gathered.numpy())
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{total} {gmax!r}")
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_exchange_equals_single_rank(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import philox

    dims, n_local, capacity = 5, 4000, 50
    z = philox.latents(7, 0, world * n_local, dims).astype(np.float64)
    log_w = -0.5 * np.sum(z**2, axis=1) * 0.3
    u = philox.uniforms(7, 0, world * n_local)
    accept = (log_w - log_w.max()) > np.log(u)
    # rank-major then draw order, each rank contributing at most `capacity` rows
    expected = np.concatenate(
        [z[r * n_local : (r + 1) * n_local][accept[r * n_local : (r + 1) * n_local]][:capacity] for r in range(world)]
    )
    outs = [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]
    assert np.array_equal(outs[0], outs[1])  # every rank ends with the same pool
    assert np.array_equal(outs[0], expected)
    for r in range(world):
        total, gmax = (tmp_path / f"rank{r}.txt").read_text().split()
        assert int(total) == int(accept.sum())
        assert float(gmax) == float(log_w.max())


def test_row_indexed_streams_do_not_depend_on_the_sharding():
    """Counter-based draws: 1 x 8000 rows == 2 x 4000 rows with row offsets."""
    sys.path.insert(0, ROOT)
    from oracle import philox

    whole = philox.latents(11, 0, 8000, 6)
    parts = np.concatenate([philox.latents(11, 0, 4000, 6), philox.latents(11, 4000, 4000, 6)])
    assert np.array_equal(whole, parts)
    assert np.array_equal(philox.uniforms(11, 0, 8000),
                          np.concatenate([philox.uniforms(11, 0, 4000), philox.uniforms(11, 4000, 4000)]))


def _pool_worker(rank, world, port, out_dir):
    """The end-of-populate exchange of the sharded trains (``FlowProposal._assemble_sharded_pool``): every rank
    holds its accepted rows appended pass after pass and its per-pass counts."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nessai_amd.proposal import FlowProposal

    counts, rows = _pool_case(rank)
    from types import SimpleNamespace

    prop = SimpleNamespace(device=torch.device("cpu"), dims=rows.shape[1])  # all the method reads of `self`
    x_acc = torch.zeros((POOL_N, rows.shape[1]), dtype=torch.float64)  # capacity = n_samples rows per rank
    kept = rows[:POOL_N]
    x_acc[: len(kept)] = torch.from_numpy(kept)
    pool = FlowProposal._assemble_sharded_pool(prop, torch.from_numpy(counts), x_acc, POOL_N, world)
    np.save(os.path.join(out_dir, f"pool{rank}.npy"), pool)
    dist.destroy_process_group()


POOL_N = 40  # n_samples


def _pool_case(rank):
    """Per-pass accepted counts and the rows (tagged pass, rank, index) of one rank; rank 1 accepts more than the
    capacity in total, pass 2 of rank 0 is empty."""
    counts = np.array([[7, 12, 0, 9, 30], [5, 20, 11, 25, 2]], dtype=np.int64)[rank]
    rows = np.array([[p, rank, k, p * 1000 + rank * 100 + k] for p, c in enumerate(counts) for k in range(c)],
                    dtype=np.float64)
    return counts, rows


@pytest.mark.timeout(120)
def test_sharded_pool_is_assembled_in_pass_rank_draw_order(tmp_path):
    world = 2
    mp.spawn(_pool_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    per_rank = [_pool_case(r) for r in range(world)]
    expected = []
    for p in range(5):
        for r in range(world):
            counts, rows = per_rank[r]
            start = int(counts[:p].sum())
            block = rows[start : start + int(counts[p])]
            expected.append(block[: max(0, POOL_N - start)])  # a rank keeps its first n_samples rows only
    expected = np.concatenate(expected)[:POOL_N]
    pools = [np.load(tmp_path / f"pool{r}.npy") for r in range(world)]
    assert np.array_equal(pools[0], pools[1]) and np.array_equal(pools[0], expected)
    assert len(expected) == POOL_N and expected[0].tolist() == [0.0, 0.0, 0.0, 0.0]
    # one rank: the same code without collectives
    sys.path.insert(0, ROOT)
    from nessai_amd.proposal import FlowProposal

    counts, rows = per_rank[0]
    from types import SimpleNamespace

    prop = SimpleNamespace(device=torch.device("cpu"), dims=4)
    x_acc = torch.zeros((POOL_N, 4), dtype=torch.float64)
    x_acc[: min(len(rows), POOL_N)] = torch.from_numpy(rows[:POOL_N])
    single = FlowProposal._assemble_sharded_pool(prop, torch.from_numpy(counts), x_acc, POOL_N, 1)
    assert np.array_equal(single, rows[:POOL_N])
