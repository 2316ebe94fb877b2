#!/usr/bin/env python
"""Where a kernel waits on the vector-memory counter: for every s_waitcnt vmcnt(N) of a kernel of a build object, the
vector-memory instruction in front of it and the first LDS / memory / barrier / MFMA instruction behind it, tallied.
A compiler-inserted vmcnt(0) right behind a global_load_lds drains the stage copies (see NsaRingPipe).

usage: isa_waits.py build/inst_h4_t1_8.o MANGLED_FRAGMENT [--show N]"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(obj, tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", os.path.join(tmp, os.path.basename(obj))], check=True,
                       capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, co[0])], check=True,
                              capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp)


def op(line):
    p = line.split()
    return p[0] if p else ""


def main():
    obj, frag = sys.argv[1:3]
    show = int(sys.argv[sys.argv.index("--show") + 1]) if "--show" in sys.argv else 0
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", disassemble(obj))
    for b in blocks:
        head = b.split("\n", 1)[0]
        if frag not in head:
            continue
        lines = b.split("\n")
        cnt = collections.Counter()
        shown = 0
        for i, l in enumerate(lines):
            m = re.search(r"s_waitcnt.*(vmcnt\(\d+\))", l)
            if not m:
                continue
            prev = next((j for j in range(i - 1, max(i - 4000, 0), -1) if re.match(r"\s*(global_|buffer_|flat_|scratch_)", lines[j])), None)
            nxt = next((j for j in range(i + 1, min(i + 40, len(lines))) if re.match(r"\s*(ds_|global_|buffer_|s_barrier|v_mfma|scratch_)", lines[j])), None)
            key = (m.group(1), op(lines[prev]) if prev else "none", i - prev if prev else -1, op(lines[nxt]) if nxt else "none")
            cnt[(key[0], key[1], key[3])] += 1
            if show and prev and "global_load_lds" in lines[prev] and m.group(1) == "vmcnt(0)" and shown < show:
                shown += 1
                print(f"--- vmcnt(0) {i - prev} lines behind a global_load_lds:")
                for l2 in lines[max(prev - 2, 0):(nxt or i) + 4]:
                    print("    ", l2.split("//")[0].rstrip())
        print(head[-70:], f"{len(lines)} lines, {sum('global_load_lds' in l for l in lines)} global_load_lds, "
              f"{sum('s_barrier' in l for l in lines)} s_barrier")
        for k, v in cnt.most_common(14):
            print("   ", v, k)


if __name__ == "__main__":
    main()
