#!/usr/bin/env python
"""Print the instruction-class sequence (M=mfma v=valu d=lds s=salu w=waitcnt B=barrier G=vmem)
of the basic blocks of a .s file that hold at least N MFMAs."""
import re
import sys
import textwrap

path, n_min = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 90
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(lines)]
for a, b in zip(starts[:-1], starts[1:]):
    seq = ""
    for l in lines[a + 1 : b]:
        t = l.strip()
        if not l.startswith("\t") or t.startswith((".", ";")):
            if "sched_barrier" in t:
                seq += "|"
            continue
        op = t.split()[0]
        seq += ("M" if op.startswith("v_mfma") else "v" if op.startswith("v_") else "d" if op.startswith("ds_")
                else "w" if op.startswith("s_waitcnt") else "B" if op.startswith("s_barrier")
                else "s" if op.startswith("s_") else "G" if op.startswith(("global", "buffer")) else "?")
    if seq.count("M") >= n_min:
        print(lines[a], "mfma", seq.count("M"), "valu", seq.count("v"))
        print("\n".join(textwrap.wrap(seq, 150)))
