#!/usr/bin/env python
"""Headline benchmark: accepted live points / second in ``FlowProposal.populate()``.

Workload (BASELINE.json configs[2]): 32-D Rosenbrock, uniform
prior on [-5, 5]^32, NeuralSplineFlow (8 RQ bins, 6 coupling layers, 2 residual
blocks, 64 hidden), nlive = poolsize = 2000, populate draw batch = 1e6, default
constant-volume truncation, z-score reparameterisation.  Weights: the flow
trained with the reference schedule on 2000 synthetic live points
(tests/golden/make_c3_weights.py) so that the acceptance rate is realistic.

A *step* is one ``populate(n_samples=2000)`` call: draw -> truncate -> inverse
flow -> rescale -> prior bounds -> weights -> rejection -> compaction -> live
points, looping over 1e6-row draw batches until 2000 points are accepted.  The
timed window is the reference's ``population_time`` window
(flowproposal.py:400-518): the likelihood evaluation of the pool is excluded.

``python bench.py --gpus N --steps K --warmup W`` (N > 1 under torchrun: one
rank per GPU, every draw batch is sharded over the ranks with one all-reduce(max) + one
all-reduce(sum) per pass and one all-gather per step).  ``--scaling strong`` (default) keeps the
1e6-row batch of BASELINE config 3 and gives each rank 1e6 / N rows, so the acceptance - and with
it the metric's meaning - is that of the 1-GPU run; ``--scaling weak`` gives every rank 1e6 rows
(an N x 1e6 batch: the acceptance per row falls with the batch size because the weights are
normalised by the batch maximum, so accepted points/s grows much more slowly than rows/s).

Environment knobs (tuning / diagnostics only): ``NESSAI_AMD_FORCE_SHARD=1`` runs the sharded
code path over RCCL on a single rank; ``NESSAI_AMD_WPG=<n>`` sets the waves per workgroup of
the flow kernel; ``NESSAI_AMD_LIBRARY=<path>`` loads another build of the library.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DIMS = 32
NLIVE = 2000
DRAWSIZE = 1_000_000
FLOPS_PER_ROW = 491_520          # conditioner GEMM FLOPs per proposed row
PEAK_TFLOPS = {"f32": 157.3,     # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)
               "bf16x3": 2500.0,  # v_mfma_f32_32x32x16_bf16 dense peak
               "f16x3": 2500.0}   # v_mfma_f32_32x32x16_f16: the same rate
DTYPE_NAMES = {"f32": "f32",
               "bf16x3": "bf16x3 (split-bf16 operands on bf16 MFMA, f32 accumulate)",
               "f16x3": "f16x3 (split-f16 operands - float32-class products - on f16 MFMA, f32 accumulate)"}
FLOW_CONFIG = dict(ftype="nsf", n_neurons=64, n_blocks=6, n_layers=2, num_bins=8)


def load_fixture():
    data = np.load(os.path.join(ROOT, "tests", "golden", "c3_trained.npz"))
    state = {k[len("state/"):]: torch.from_numpy(data[k]) for k in data.files if k.startswith("state/")}
    return state, data["shift"], data["scale"]


def build_proposal(rank, world, device, scaling="strong"):
    from nessai_amd.model import RosenbrockModel
    from nessai_amd.proposal import FlowProposal

    state, shift, scale = load_fixture()
    model = RosenbrockModel(DIMS)
    prop = FlowProposal(
        model,
        rng=np.random.default_rng(1451),
        flow_config=FLOW_CONFIG,
        training_config=dict(device_tag=str(device)),
        poolsize=NLIVE,
        drawsize=DRAWSIZE * (world if scaling == "weak" else 1),  # the TOTAL draw batch of one pass
        device_seed=1451,
        shard=world > 1,
        output=os.path.join("/tmp", f"nessai_amd_bench_{os.getpid()}"),
    )
    prop.initialise()
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    prop.set_rescaling_constants(shift, scale)
    if os.environ.get("NESSAI_AMD_WPG"):  # tuning knob: waves per workgroup of the flow kernel
        prop.flow.model.waves_per_group = int(os.environ["NESSAI_AMD_WPG"])
    return prop


def time_draw_kernel(prop, launches=10, precision=None, rows=None):
    """Average duration of the dominant kernel (fused draw+inverse-flow) over ``rows`` rows (default: the
    whole 1e6-row batch; for N > 1 the rank's share of it - what a launch of the sharded run really
    processes), HIP events on the stream it is launched on (torch's current stream)."""
    n = DRAWSIZE if rows is None else int(rows)
    keep = prop.precision
    if precision is not None:
        prop.precision = precision
    try:
        return _time_draw_kernel(prop, n, launches)
    finally:
        prop.precision = keep


def _time_draw_kernel(prop, n, launches):
    buf = prop._get_buffers(n, NLIVE)
    prepared = prop._prepare_flow()
    # the launch populate() makes: since round 5 the lean pass (no candidate rows: log_q alone is written)
    lean = prop.last_populate_lean

    def launch(offset):
        if lean:  # the head of a lean pass: log_w and its maximum, 8 bytes per row
            prop._launch_draw_weights(n, offset, prop.fixed_radius, buf["log_w"], buf["max_key"], log_prior=buf["log_p"],
                                      prepared=prepared)
        else:
            prop._launch_draw(n, None, offset, prop.fixed_radius, buf["x64"], buf["log_q"], None, prepared=prepared)

    for _ in range(2):
        launch(0)
    torch.cuda.synchronize()
    events = []
    for i in range(launches):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        launch(i * n)
        e.record()
        events.append((s, e))
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in events]
    global LAST_GEOMETRY
    LAST_GEOMETRY = last_geometry()
    return float(np.mean(ms)), float(np.min(ms))


LAST_GEOMETRY = None


def last_geometry():
    """Launch geometry the library chose for the last flow kernel (nsa_last_geometry)."""
    import ctypes

    from nessai_amd import lib

    v = [ctypes.c_int(0) for _ in range(3)]
    lib.load().nsa_last_geometry(*[ctypes.addressof(x) for x in v])
    return {"waves_per_simd": v[0].value, "waves_per_group": v[1].value, "workgroups": v[2].value,
            "row_tiles_per_wave": int(lib.load().nsa_last_row_tiles())}


def traffic_probe(device):
    """``--traffic-probe``: a few launches of the draw kernel and nothing else (the command that
    ``measure_traffic`` runs under rocprofv3)."""
    prop = build_proposal(0, 1, device)
    prop.populate(None, n_samples=NLIVE, max_samples=10**12)
    time_draw_kernel(prop, launches=4)


def measure_traffic(waves_per_simd=None):
    """HBM bytes per launch of the draw kernel from the PMC counters, collected live: two
    ``rocprofv3 --kernel-trace --pmc`` passes over ``bench.py --traffic-probe`` (FETCH_SIZE and
    WRITE_SIZE do not fit one pass), corrected as MI355X_MICROARCH.md (HBM) prescribes: both counters
    are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes
    (x2).  Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    per_launch = {}
    tmp = tempfile.mkdtemp(prefix="nsa_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    if waves_per_simd in (2, 3, "2x2"):
        # the probe is another process: pin it to the geometry THIS run's launches used (the library
        # times its candidates once per process; their launches would be in the probe's trace too)
        env["NESSAI_AMD_OCC"] = str(waves_per_simd)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "pmc",
                   "--", sys.executable, os.path.abspath(__file__), "--traffic-probe"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            values = []
            for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as f:
                    for row in csv.DictReader(f):
                        if "nsa_flow_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            values.append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
            if not values:
                return None, f"no {counter} rows for the flow kernel in the rocprofv3 output"
            full = max(g for g, _ in values)  # the 1e6-row launches (the populate call also makes them)
            vals = [v for g, v in values if g == full]
            # median: the populate call's switched-off surplus passes have the same grid and move nothing
            per_launch[counter] = float(np.median(vals)) * 1024.0
        total = 2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]
        return total, {"fetch_size_bytes_x2": 2.0 * per_launch["FETCH_SIZE"], "write_size_bytes": per_launch["WRITE_SIZE"],
                       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, this run"}
    except Exception as exc:  # the counters are a cross-check, never the reason a bench run fails
        return None, f"{type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _oracle_proposal(drawsize, chunk_rows=None):
    from oracle import proposal as OP

    state, shift, scale = load_fixture()
    fm = OP.OracleFlowModel(
        dict(n_inputs=DIMS, n_neurons=64, n_blocks=6, n_layers=2, ftype="nsf"),
        rng=np.random.default_rng(1451),
    )
    fm.model.load_state_dict(state)
    fm.model.eval()
    fm.chunk_rows = chunk_rows
    model = OP.RosenbrockModel(DIMS)
    prop = OP.OracleFlowProposal(model, fm, rng=np.random.default_rng(1451), poolsize=NLIVE, drawsize=drawsize)
    for i, n in enumerate(model.names):
        prop.shift[n] = float(shift[i])
        prop.scale[n] = float(scale[i])
    return prop


ALL_CORES_SAMPLE = 250_000
ALL_CORES_THREADS = 32   # eager torch on ~100 us operators stops scaling long before a 256-thread host is full
ALL_CORES_TIMEOUT = 120  # seconds


def cpu_all_cores_leg():
    """Leg (ii) of SURVEY.md section 8d: the same oracle pass with many torch threads.  Run in a
    child process under a timeout: intra-op threading of the eager restatement is slower, not
    faster, on hosts with hundreds of cores (1e6 draws took 1060 s on 256 threads of the GPU box
    against 53 s on one), and a baseline leg must not decide how long the benchmark runs."""
    import subprocess

    cores = min(os.cpu_count() or 1, ALL_CORES_THREADS)
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", str(cores)], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=ALL_CORES_TIMEOUT)
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"value": None, "cores": cores, "kind": "port", "host_cpus": os.cpu_count(),
                "sample": f"one populate pass of {ALL_CORES_SAMPLE} draws did not finish in {ALL_CORES_TIMEOUT} s "
                          f"on {cores} torch threads (slower than the single-thread leg)"}
    except Exception as exc:
        return {"value": None, "cores": cores, "kind": "port", "sample": f"failed: {type(exc).__name__}: {exc}"}


def cpu_leg_child(cores):
    """``--cpu-leg N``: one bounded oracle pass on N torch threads, one JSON line (child of the above)."""
    torch.set_num_threads(cores)
    prop = _oracle_proposal(ALL_CORES_SAMPLE, chunk_rows=50_000)
    torch.manual_seed(1451)
    t = time.perf_counter()
    prop.populate(n_samples=NLIVE, max_samples=ALL_CORES_SAMPLE - 1)
    dt = time.perf_counter() - t
    print(json.dumps({
        "value": prop.n_accepted / dt, "unit": "accepted points/s", "cores": cores, "kind": "port",
        "host_cpus": os.cpu_count(),
        "sample": f"one populate pass of {ALL_CORES_SAMPLE} draws (flow inverse in chunks of 50000 rows), "
                  f"{prop.n_accepted} accepted in {dt:.1f} s",
        "proposed_rows_per_s": prop.n_proposed / dt, "acceptance": prop.population_acceptance,
    }))


def train_leg(epochs=40):
    """Secondary figure (not the metric): `FlowModel.train` on this config's flow - the optimiser step time a
    nested-sampling run pays every ~nlive iterations (src/nessai/flowmodel/base.py:530-696).  The committed
    live points of config 3 (tests/golden/c3_trained.npz), 2000 rows -> minibatches of 1000 + 800 and a
    200-row validation pass per epoch; a two-epoch fit first (workspace, first launches, the epoch graph)."""
    import tempfile

    from nessai_amd.flowmodel import FlowModel

    data = np.load(os.path.join(ROOT, "tests", "golden", "c3_trained.npz"))
    x = (data["live_points"] - data["shift"]) / data["scale"]
    cfg = dict(n_inputs=32, n_neurons=64, n_blocks=6, n_layers=2, ftype="nsf")
    torch.manual_seed(1451)
    with tempfile.TemporaryDirectory() as tmp:
        fm = FlowModel(flow_config=cfg, training_config=dict(max_epochs=epochs, patience=10**6), output=tmp,
                       rng=np.random.default_rng(1451))
        fm.initialise()
        fm.train(x, max_epochs=2)
        fm.reset_model()
        torch.cuda.synchronize()
        t = time.perf_counter()
        hist = fm.train(x, max_epochs=epochs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    steps = 2 * len(hist["loss"])
    return {"what": "FlowModel.train, config 3 flow, 2000 live points", "ms_per_optimiser_step": 1e3 * dt / steps,
            "epochs": len(hist["loss"]), "steps": steps, "final_loss": float(hist["loss"][-1])}


def _cpu_pass(sample_draw, chunk_rows):
    """One populate pass of the oracle over ``sample_draw`` draws on the calling thread count."""
    prop = _oracle_proposal(sample_draw, chunk_rows=chunk_rows)
    torch.manual_seed(1451)
    t = time.perf_counter()
    prop.populate(n_samples=NLIVE, max_samples=sample_draw - 1)  # exactly one pass
    dt = time.perf_counter() - t
    return prop, dt


def cpu_baseline(full=True):
    """The oracle (CPU restatement of the reference path) timed on this host.

    The reported baseline (``value``): ONE torch thread - the reference's default
    (src/nessai/flowsampler.py:98) - over ONE pass of the metric's own configuration, the 1e6-draw batch
    (flow inverse in chunks of 1e5 rows; ~50 s on the GPU box's host).  Secondary figures in the same object:
    ``bounded_sample`` - the 2.5e5-draw pass earlier rounds reported (its acceptance per row is ~2.5x the full
    batch's because the weights are normalised by the batch maximum: flattering to the CPU) - and ``all_cores``
    (SURVEY.md section 8d ii): the bounded pass on many threads (``cpu_all_cores_leg``).  ``full=False``
    (``--cpu-bounded``): the bounded sample alone, as before round 5."""
    threads = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        prop_b, dt_b = _cpu_pass(250_000, None)
        bounded = {
            "value": prop_b.n_accepted / dt_b, "unit": "accepted points/s", "cores": 1, "kind": "port",
            "sample": f"one populate pass of 250000 draws (of the 1e6-row batch), {prop_b.n_accepted} accepted in {dt_b:.1f} s",
            "proposed_rows_per_s": prop_b.n_proposed / dt_b, "acceptance": prop_b.population_acceptance,
        }
        if full:
            prop, dt = _cpu_pass(DRAWSIZE, 100_000)
            out = {
                "value": prop.n_accepted / dt, "unit": "accepted points/s", "cores": 1, "kind": "port",
                "sample": (f"one populate pass of the metric's own {DRAWSIZE}-draw batch (flow inverse in chunks of "
                           f"100000 rows), {prop.n_accepted} accepted in {dt:.1f} s"),
                "proposed_rows_per_s": prop.n_proposed / dt, "acceptance": prop.population_acceptance,
                "host_cpus": os.cpu_count(), "bounded_sample": bounded,
            }
        else:
            out = dict(bounded, host_cpus=os.cpu_count())
        out["all_cores"] = cpu_all_cores_leg()
        return out
    finally:
        torch.set_num_threads(threads)


def self_launch(n_gpus, json_fd):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks."""
    import socket
    import subprocess

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU fallback)")
    have = torch.cuda.device_count()
    if have < n_gpus:
        raise SystemExit(f"bench.py --gpus {n_gpus} needs {n_gpus} devices on this node, found {have}")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the children's stdout carries rank 0's JSON line only (every rank sends library chatter to stderr)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or len(lines) != 1:
        sys.stderr.write(proc.stdout.decode())
        raise SystemExit(f"bench.py: the {n_gpus}-rank run failed (exit code {proc.returncode}, {len(lines)} JSON lines)")
    os.write(json_fd, (lines[0] + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the secondary FlowModel.train timing")
    ap.add_argument("--cpu-full", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--cpu-bounded", action="store_true",
                    help="time the single-thread CPU baseline on the bounded 2.5e5-draw sample only (~15 s) instead "
                         "of one pass of the metric's own 1e6-draw batch (~50 s more)")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-leg", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: 'strong' shards the 1e6-row draw batch of BASELINE config 3 over the ranks "
                         "(1e6 / N rows each, same acceptance as at N = 1); 'weak' gives every rank 1e6 rows "
                         "(an N x 1e6 batch: more rows per pass but a lower acceptance per row)")
    args = ap.parse_args()
    if args.cpu_leg:
        cpu_leg_child(args.cpu_leg)
        return

    # stdout carries exactly ONE JSON line: libraries that write to the C-level stdout (RCCL prints a
    # version banner there, flushed at exit) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run, rendezvous on 127.0.0.1); rank 0 of the children prints the JSON line
        self_launch(args.gpus, json_fd)
        return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.traffic_probe:
        traffic_probe(device)
        return
    force_shard = os.environ.get("NESSAI_AMD_FORCE_SHARD") == "1"  # exercise the RCCL path on 1 rank
    if world > 1 or force_shard:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if force_shard and "RANK" not in os.environ:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group("nccl", device_id=device)

    if DRAWSIZE % world:
        raise SystemExit(f"--gpus {world} does not divide the draw batch of {DRAWSIZE} rows")
    prop = build_proposal(rank, world, device, args.scaling)
    batch = prop.drawsize
    if force_shard:
        prop.shard = True

    def barrier():
        if world > 1 or force_shard:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        prop.populate(None, n_samples=NLIVE, max_samples=10**12)
    barrier()
    t0 = time.perf_counter()
    accepted = proposed = 0
    for _ in range(args.steps):
        prop.populate(None, n_samples=NLIVE, max_samples=10**12)
        accepted += prop.n_accepted
        proposed += prop.n_proposed
    barrier()
    elapsed = time.perf_counter() - t0
    # likelihood evaluation of the pool is outside the reference's window: subtract nothing,
    # report the conservative wall time of whole populate() calls (it includes logL of 2000 rows)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # the roofline kernel is timed on what ONE launch of this run processes: the rank's share of the batch
    rows_per_launch = batch // world
    kernel_ms, kernel_ms_min = time_draw_kernel(prop, rows=rows_per_launch)
    geometry = LAST_GEOMETRY
    precision = prop.flow.model.effective_precision  # what populate() actually ran ("auto" resolved by the flow)
    achieved = FLOPS_PER_ROW * rows_per_launch / (kernel_ms * 1e-3) / 1e12
    f32_ms, _ = time_draw_kernel(prop, precision="f32", rows=rows_per_launch)
    f32_achieved = FLOPS_PER_ROW * rows_per_launch / (f32_ms * 1e-3) / 1e12

    # N > 1: every rank's kernel time and what the sharded populate did (trains, passes, collectives), so that the
    # first real multi-GPU run explains itself
    kernel_ms_per_rank = [kernel_ms]
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, float(kernel_ms))
        kernel_ms_per_rank = [float(v) for v in gathered]
    sharded_stats = getattr(prop, "sharded_stats", None) if (world > 1 or force_shard) else None

    traffic, traffic_detail = (None, "skipped (--no-pmc or N > 1)")
    if rank == 0 and world == 1 and not args.no_pmc:
        traffic, traffic_detail = measure_traffic("2x2" if (geometry or {}).get("row_tiles_per_wave") == 2
                                                  else (geometry or {}).get("waves_per_simd"))
    if rank == 0:
        out = {
            "metric": "accepted live points/sec in FlowProposal.populate(), 32-D Rosenbrock nlive=2000",
            "value": accepted / elapsed,
            "unit": "accepted points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong",
            "vs_baseline": None,
            "dtype": DTYPE_NAMES[precision],
            # populate() asks for "auto": the first 16-bit matrix-core mode (split-f16, then split-bf16) that a
            # 4096-row probe of THIS flow shows to agree with the exact-float32 kernels to 5e-6 relative in
            # both directions (HipFlow._pack_auto)
            "precision_policy": {"requested": prop.precision, "effective": precision,
                                 "probe_rel_err_vs_f32": prop.flow.model.probe_error,
                                 "probe_rel_errs": prop.flow.model.probe_errors},
            "data": "synthetic",
            "config": {
                "workload": "32-D Rosenbrock, NeuralSplineFlow (8 RQ bins, 6 coupling layers, "
                            f"2x64 residual conditioner), nlive=poolsize=2000, populate batch={batch:.0e} "
                            f"({batch // world} rows per GPU)",
                "drawsize": batch,
                "drawsize_per_gpu": batch // world,
                "n_samples": NLIVE,
                "weights": "trained on 2000 synthetic live points (tests/golden/c3_trained.npz)",
                "parallelism": f"draw batch sharded over {world} GPU(s)",
            },
            "proposed_rows_per_s": proposed / elapsed,
            # the last sharded populate of the timed region: trains of passes (two all-reduces each), passes,
            # all-gathers (per-pass counts + accepted rows) - FlowProposal.sharded_stats; null at N = 1
            "sharded_populate": sharded_stats,
            "acceptance": accepted / proposed,
            "passes_per_step": proposed / batch / args.steps,
            "roofline": {
                "kernel": f"nsa_flow_kernel<2, NsaRQ<8>, inverse, {precision}> "
                          "(fused draw + inverse flow + rescale + bounds + log_q" +
                          ("; lean: no candidate rows written)" if prop.last_populate_lean else ")"),
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_TFLOPS[precision],
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_TFLOPS[precision],
                # HBM bytes per launch: PMC counters read in THIS run (measure_traffic); algorithmic bytes per
                # row: 8 (log_w) for the lean pass populate() runs since round 5 - only accepted rows are ever
                # used, and those are computed again from their Philox rows - 264 (x as float64 + log_q) before
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "lean_pass": bool(prop.last_populate_lean),
                "algorithmic_bytes": (8.0 if prop.last_populate_lean else 264.0) * rows_per_launch,
                "kernel_ms_avg": kernel_ms,
                "kernel_ms_min": kernel_ms_min,
                # chosen by the library (big batches of the 16-bit modes: round-count estimate of nsa_big_batch_geometry, nsa_api.hip)
                "kernel_ms_per_rank": kernel_ms_per_rank,
                "launch_geometry": geometry,
                "rows_per_launch": rows_per_launch,
                "rows_per_launch_note": "the whole batch at N = 1; for N > 1 this rank's share of it (rank 0 times its own kernel)",
                "flops_per_row": FLOPS_PER_ROW,
                "note": "algorithmic FLOPs (one product per weight); the split 16-bit modes issue 3 MFMAs "
                        "per product and are bound by the spline's VALU/transcendental work, not by the matrix pipe",
            },
            "roofline_f32_mode": {
                "kernel": "nsa_flow_kernel<2, NsaRQ<8>, inverse, f32> (exact float32 products)",
                "bound": "mfma",
                "achieved": f32_achieved,
                "peak": PEAK_TFLOPS["f32"],
                "unit": "TFLOP/s",
                "frac": f32_achieved / PEAK_TFLOPS["f32"],
                "kernel_ms_avg": f32_ms,
            },
        }
        if world == 1 and not args.no_train_leg:
            try:
                out["train"] = train_leg()
            except Exception as exc:  # a secondary figure must not cost the line
                out["train"] = {"error": repr(exc)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(full=not args.cpu_bounded)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            # one CPU pass is one draw of a heavy-tailed quantity (the pass maximum sets its acceptance): the
            # CPU's row rate at the acceptance the GPU averaged over its many passes is
            out["cpu_baseline"]["value_at_full_batch_acceptance"] = (
                out["cpu_baseline"]["proposed_rows_per_s"] * out["acceptance"]
            )
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_shard:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
