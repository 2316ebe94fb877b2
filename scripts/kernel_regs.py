#!/usr/bin/env python
"""Register / scratch metadata of the kernels in a hipcc object or shared library.

usage: kernel_regs.py FILE.o [SUBSTRING ...]
Extracts the gfx950 code object (llvm-objdump --offloading) into a temporary directory and prints, per
kernel whose demangled name contains every SUBSTRING: vgpr_count, agpr_count, vgpr_spill_count,
private_segment_fixed_size (scratch bytes per lane), sgpr_spill_count and the text size."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(path):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        out = []
        for f in co:
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                block = ".agpr_count:" + block
                get = lambda k: re.search(r"\.%s:\s+(\S+)" % k, block)
                name = get("name").group(1)
                out.append(dict(name=name, vgpr=int(get("vgpr_count").group(1)), agpr=int(get("agpr_count").group(1)),
                                spill=int(get("vgpr_spill_count").group(1)), sgpr_spill=int(get("sgpr_spill_count").group(1)),
                                scratch=int(get("private_segment_fixed_size").group(1)),
                                lds=int(get("group_segment_fixed_size").group(1))))
        return out
    finally:
        shutil.rmtree(tmp)


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    rows = kernels(path)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True,
                           text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        n = n.replace("void ", "").replace("(NsaLaunch)", "")
        if all(s in n for s in subs):
            print(f"{n:70s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} spilled {r['spill']:3d} scratch {r['scratch']:4d} B "
                  f"sgpr spilled {r['sgpr_spill']:3d}")


if __name__ == "__main__":
    main()
