#!/usr/bin/env python
"""Per-basic-block MFMA / VALU / DS statistics of a gfx950 .s file (scheduling audit)."""
import re
import statistics
import sys

lines = open(sys.argv[1]).read().split("\n")
blocks, cur, name, kern = [], [], "entry", ""
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern = m.group(1)[:60]
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((kern, name, cur))
        name, cur = l.split(":")[0], []
    else:
        cur.append(l)
blocks.append((kern, name, cur))
for kern, name, b in blocks:
    insts = [l.strip().split()[0] for l in b if l.startswith("\t") and not l.strip().startswith((".", ";", "//"))]
    nm = sum(i.startswith("v_mfma") for i in insts)
    nv = sum(i.startswith("v_") and not i.startswith("v_mfma") for i in insts)
    if nm >= 16 or nv > 300:
        gaps, g, seen = [], 0, False
        for i in insts:
            if i.startswith("v_mfma"):
                if seen:
                    gaps.append(g)
                g, seen = 0, True
            elif i.startswith(("v_", "ds_", "s_")):
                g += 1
        print(f"{kern[-28:]:28s} {name:12s} insts {len(insts):5d} mfma {nm:4d} valu {nv:5d} ds {sum(i.startswith('ds_') for i in insts):4d} "
              f"gap mean {statistics.mean(gaps) if gaps else 0:5.1f} max {max(gaps) if gaps else 0:4d} zero {sum(g == 0 for g in gaps):3d}")
