#!/usr/bin/env python
"""Hash of the instruction stream of every kernel in a hipcc object / library (refactors that must not change code).

usage: isa_hash.py FILE.o [SUBSTRING ...]
Extracts the gfx950 code object (llvm-objdump --offloading), disassembles it and prints, per kernel whose mangled name
contains every SUBSTRING, the md5 of its instructions (addresses and encodings stripped) and their count."""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        for f in sorted(x for x in os.listdir(tmp) if "amdgcn" in x):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            name, ins = None, []
            out = {}
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    name, ins = m.group(1), []
                    out[name] = ins
                elif name and line.startswith("\t"):
                    ins.append(re.sub(r"\s*//.*$", "", line.strip()))
            for name, ins in out.items():
                if all(s in name for s in subs) and ins:
                    print(hashlib.md5("\n".join(ins).encode()).hexdigest(), len(ins), name)
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
