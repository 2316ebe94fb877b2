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


# ---- the train-of-passes loop of the sharded populate on 8 ranks -----------------------------------------------------
# FlowProposal._populate_trains_sharded itself - train lengths from the acceptance seen so far, ONE all-reduce(max)
# and ONE all-reduce(sum) per train, the commit's stop test on the GLOBAL counts, the per-rank row lists, the final
# all-gathers and the (pass, rank, draw) interleave - driven by numpy restatements of the contracts of its three
# device steps (include/nessai_amd.h: nsa_populate_train_heads_lean / _tails_lean / _commit, nsa_populate_replay).
# What a GPU adds to this is the arithmetic of a row; what 8 ranks add is here.

TRAIN_WORLD = 8
TRAIN_DIMS = 3
TRAIN_SEED = 23
INT64_MAX = 0x7FFFFFFFFFFFFFFF


def _row_weights(first_row, n):
    """(log_w, log u) of the global rows [first_row, first_row + n): a stand-in for flow + prior whose acceptance
    differs from one block of rows to the next, so that the ranks' per-pass counts are uneven."""
    from oracle import philox

    z = philox.latents(TRAIN_SEED, first_row, n, TRAIN_DIMS).astype(np.float64)
    g = np.arange(first_row, first_row + n)
    log_w = -0.5 * np.sum(z**2, axis=1) * (10.0 + 8.0 * ((g // 700) % 5))
    log_w[(g % 97) == 0] = -np.inf  # dropped rows (outside the bounds / the radius)
    return log_w, np.log(philox.uniforms(TRAIN_SEED, first_row, n)), z


def _stopped(state):
    return bool(state[0] >= state[2] or state[1] >= state[5])


def _make_emulated(rank, world, n_local, acceptance):
    from types import SimpleNamespace

    from nessai_amd import lib
    from nessai_amd.proposal import FlowProposal

    L = lib.load()

    class Emulated(FlowProposal):
        def __init__(self):  # (no flow, no device: only what the sharded loop reads)
            self.device = torch.device("cpu")
            self.model = SimpleNamespace(names=[f"x{i}" for i in range(TRAIN_DIMS)])
            self._reparam = SimpleNamespace(names=self.model.names, prime_names=self.model.names)
            self.population_acceptance = acceptance
            self._draw_counter = 0
            self._ll_spec = None
            self.shard = True
            self.sharded_stats = None
            self.result = None

        dims = TRAIN_DIMS
        names = property(lambda self: self.model.names)

        def _sharded_setup(self, n, n_samples):
            buf = dict(x_acc=torch.zeros((n_samples, TRAIN_DIMS), dtype=torch.float64), log_q=None, log_p=None)
            return dict(prepared=None, buf=buf, rules=None, lean=True, separable=False, log_prior_const=0.0,
                        key_init=int(L.nsa_max_key_init()), stream=None)

        def _sharded_heads(self, ctx, n, train, off, stride, threshold, keys, state):
            arena = ctx["buf"]["log_w_many"]
            for i in range(train):
                if _stopped(state):  # (the head kernels of a train that is already over return at once)
                    continue
                log_w, _, _ = _row_weights(off + i * stride, n)
                arena[i] = torch.from_numpy(log_w)
                keys[i] = int(L.nsa_max_key_encode(float(log_w.max())))

        def _sharded_tails(self, ctx, n, train, off, stride, n_samples, keys, state, lcount):
            buf = ctx["buf"]
            scratch = buf["scratch"]
            scratch[:] = torch.tensor([0, 0, INT64_MAX, int(state[3]), 0, INT64_MAX, 0, 0])  # nsa_train_scratch_kernel
            for i in range(train):
                first = off + i * stride
                _, log_u, _ = _row_weights(first, n)
                gmax = L.nsa_max_key_decode(int(keys[i]))
                accept = (buf["log_w_many"][i].numpy() - gmax) > log_u
                rows = first + np.nonzero(accept)[0]
                written = int(scratch[3])
                room = max(0, n_samples - written)
                buf["idx_acc"][written: written + min(room, len(rows))] = torch.from_numpy(rows[:room])
                scratch[4] = len(rows)
                # nsa_pass_commit_kernel on the scratch state, history = lcount
                lcount[int(scratch[1])] = len(rows)
                scratch[0] += len(rows)
                scratch[1] += 1
                scratch[3] = min(written + len(rows), n_samples)

        def _sharded_commit(self, ctx, train, n_samples, state, lcount, gcount, history):
            for i in range(train):  # nsa_train_commit_kernel
                if _stopped(state):
                    return
                got = int(lcount[i])
                history[int(state[1])] = got
                state[0] += int(gcount[i])
                state[1] += 1
                state[3] = min(int(state[3]) + got, n_samples)

        def _replay(self, idx, m, radius, x64, log_q, log_prior, prepared):
            for k in range(m):  # nsa_populate_replay: the row again from its global Philox row
                x64[k] = torch.from_numpy(_row_weights(int(idx[k]), 1)[2][0])

        def _finish_population(self, st, n_accepted, n_proposed, log_l=None, f16_flag=None):
            self.result = (np.stack([self.x[n] for n in self.names], axis=1), int(n_accepted), int(n_proposed))
            self.population_acceptance = n_accepted / n_proposed

    return Emulated()


def _trains_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, n_samples = 2500, 1500
    # first call: the acceptance guess is twice the truth, so the first train undershoots and a SHORT second train
    # follows; second call: the acceptance measured by the first
    prop = _make_emulated(rank, world, n_local, acceptance=0.02)
    outs = []
    for call in range(2):
        prop._populate_trains_sharded(None, np.inf, n_samples, 10**9, n_local, world, rank)
        outs.append((prop.result, dict(prop.sharded_stats), prop._draw_counter))
    # third call: the max_samples exit inside a train
    prop._populate_trains_sharded(None, np.inf, 10**6, 9 * n_local * world, n_local, world, rank)
    outs.append((prop.result, dict(prop.sharded_stats), prop._draw_counter))
    np.save(os.path.join(out_dir, f"trains{rank}.npy"), np.array(outs, dtype=object), allow_pickle=True)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_trains_of_sharded_passes_on_eight_ranks_equal_the_one_rank_loop(tmp_path):
    world, n_local = TRAIN_WORLD, 2500
    mp.spawn(_trains_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    per_rank = [np.load(tmp_path / f"trains{r}.npy", allow_pickle=True) for r in range(world)]

    # the reference loop: one pass at a time over the whole batch (flowproposal.py:431-518)
    stride = n_local * world
    counter = 0
    for call, (n_samples, pass_limit) in enumerate([(1500, 10**9), (1500, 10**9), (10**6, 9 * stride // stride + 1)]):
        pool, accepted, passes = [], 0, 0
        uneven = []
        while accepted < n_samples and passes < pass_limit:
            log_w, log_u, z = _row_weights(counter + passes * stride, stride)
            keep = (log_w - log_w.max()) > log_u
            pool.append(z[keep])
            uneven.append([int(keep[r * n_local:(r + 1) * n_local].sum()) for r in range(world)])
            accepted += int(keep.sum())
            passes += 1
        counter += passes * stride
        want = np.concatenate(pool)[:n_samples]
        for r in range(world):
            (x, n_acc, n_prop), stats, draw_counter = per_rank[r][call]
            assert n_acc == accepted and n_prop == passes * stride and draw_counter == counter
            assert np.array_equal(x, want)
            assert stats["passes"] == passes and sum(stats["train_lengths"]) >= passes
            assert stats["all_reduce"] == 2 * stats["trains"] and stats["all_gather"] == 2
        stats = per_rank[0][call][1]
        counts = np.array(uneven)
        assert counts.min() < counts.max()  # the ranks' per-pass counts are uneven
        if call == 0:
            assert stats["trains"] >= 2 and stats["train_lengths"][-1] < stats["train_lengths"][0]  # a short final train
        if call == 1:
            assert stats["trains"] <= 3  # the measured acceptance sizes the train: 2-3 rounds of collectives
        if call == 2:
            assert passes == 10 and accepted < n_samples  # stopped by max_samples, mid-train
