#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) outputs of scripts/profile_bench.sh into text.

    python scripts/summarise_prof.py gpurun_out/prof_r01 > profiles/r01_bench_c3.txt
"""
import glob
import os
import sqlite3
import sys


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), "
        "max(sgpr_count), max(scratch_size) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    out = []
    out.append(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}  grid  wg  lds  vgpr agpr sgpr scratch")
    for r in rows:
        name = r[0][:70]
        out.append(
            f"{name:70s} {r[1]:6d} {r[2]/1e6:10.3f} {r[3]/1e3:10.2f} {r[4]/1e3:10.2f} {r[5]/1e3:10.2f} "
            f"{100*r[2]/total:6.2f}  {r[6]} {r[7]} {r[8]} {r[9]} {r[10]} {r[11]} {r[12]}"
        )
    return "\n".join(out)


def counter_stats(db_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    rows = db.execute(
        "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
        "group by kernel_name, counter_name order by kernel_name, counter_name"
    ).fetchall() if "kernel_name" in cols else []
    out = [f"{'kernel':60s} {'counter':28s} {'launches':>8s} {'sum':>18s} {'avg/launch':>18s}"]
    for k, c, n, s, a in rows:
        out.append(f"{k[:60]:60s} {c:28s} {n:8d} {s:18.0f} {a:18.1f}")
    return "\n".join(out), cols


def main():
    root = sys.argv[1]
    trace = os.path.join(root, "trace", "trace_results.db")
    if os.path.exists(trace):
        print("== rocprofv3 --kernel-trace --stats (per-kernel) ==")
        print(kernel_stats(trace))
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "pmc_results.db"))):
        print()
        print(f"== rocprofv3 --pmc  [{os.path.basename(os.path.dirname(db))}] ==")
        text, cols = counter_stats(db)
        print(text if text.count("\n") else f"(no counters_collection rows; columns {cols})")


if __name__ == "__main__":
    main()


def gap_stats(db_path, big_us=20.0):
    """Idle time between consecutive kernels of a trace: how much of the wall time is not kernels, and how it
    splits into short gaps (launch-to-launch inside a graph or a stream) and long ones (host round trips)."""
    db = sqlite3.connect(db_path)
    rows = db.execute("select start, end, name from kernels order by start").fetchall()
    if len(rows) < 2:
        return "no kernels"
    busy = sum(r[1] - r[0] for r in rows)
    gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
    short = [g for g in gaps if g < big_us * 1e3]
    long_ = [g for g in gaps if g >= big_us * 1e3]
    span = rows[-1][1] - rows[0][0]
    out = [f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %)",
           f"gaps < {big_us:.0f} us: {len(short)}  total {sum(short) / 1e6:.3f} ms  mean {sum(short) / max(len(short), 1) / 1e3:.2f} us",
           f"gaps >= {big_us:.0f} us: {len(long_)}  total {sum(long_) / 1e6:.3f} ms  mean {sum(long_) / max(len(long_), 1) / 1e3:.2f} us"]
    return "\n".join(out)
