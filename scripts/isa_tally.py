#!/usr/bin/env python
"""Tally the instruction mix of a loop of a gfx950 assembly listing (hipcc -S).

usage: isa_tally.py FILE.s KERNEL_SUBSTRING MARKER_OPCODE
Finds the kernel whose mangled name contains KERNEL_SUBSTRING, then the innermost loop (by the
compiler's "Loop Header" comments / backward branches) that contains MARKER_OPCODE, and prints the
opcode histogram of that loop body grouped into categories."""
import collections
import re
import sys


def main():
    path, kernel, marker = sys.argv[1:4]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and kernel in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    marks = [i for i, l in enumerate(body) if re.match(r"\s+" + marker, l)]
    print(f"kernel lines {len(body)}; {len(loops)} loops; {len(marks)} x {marker}")
    cands = [(b - a, a, b) for a, b in loops if any(a <= m <= b for m in marks)]
    cands.sort()
    seen = set()
    for size, a, b in cands[: int(sys.argv[4]) if len(sys.argv) > 4 else 1]:
        tally(body[a : b + 1], f"loop lines {a}-{b} ({size} lines)")


CATS = [
    ("mfma", r"v_mfma"), ("trans", r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_"), ("pk", r"v_pk_"),
    ("cvt", r"v_cvt"), ("cndmask", r"v_cndmask"), ("cmp", r"v_cmp"), ("fma", r"v_(fma|fmac|mad|mac)_"),
    ("mul", r"v_mul_"), ("addsub", r"v_(add|sub|subrev)_"), ("minmax", r"v_(max|min|max3|min3|med3)"),
    ("mov", r"v_(mov|accvgpr|readlane|readfirstlane|swap)"), ("bitop", r"v_(and|or|xor|lshl|lshr|ashr|bfe|bfi|perm|not|lshlrev|lshrrev|alignbit|and_or|lshl_or|lshl_add|add_lshl)"),
    ("valu_other", r"v_"), ("ds_read", r"ds_read|ds_load"), ("ds_write", r"ds_write|ds_store"), ("ds_other", r"ds_"),
    ("vmem", r"(global|buffer|flat|scratch)_"), ("smem", r"s_load|s_buffer"), ("waitcnt", r"s_waitcnt"),
    ("barrier", r"s_barrier"), ("nop", r"s_nop"), ("salu", r"s_"),
]


def tally(block, title):
    hist = collections.Counter()
    ops = collections.Counter()
    for l in block:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m or l.strip().startswith((".", ";")):
            continue
        op = m.group(1)
        ops[op] += 1
        for name, pat in CATS:
            if re.match(pat, op):
                hist[name] += 1
                break
    print("==", title)
    valu = sum(v for k, v in hist.items() if k in ("trans", "pk", "cvt", "cndmask", "cmp", "fma", "mul", "addsub", "minmax", "mov", "bitop", "valu_other"))
    print("   VALU (non-MFMA):", valu)
    for k, v in hist.most_common():
        print(f"   {k:12s} {v}")
    print("   top opcodes:", ", ".join(f"{k}:{v}" for k, v in ops.most_common(28)))


main()
