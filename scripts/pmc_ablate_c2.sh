#!/bin/bash
# VALU / SALU / LDS instruction counts of the C2 draw kernel per section: one counter pass over each ablation
# build (scripts/ablate_c2.sh), full minus ablated = what the section executes.  GPU box, repo root.
out=${1:-gpurun_out/pmc_ablate_c2}
mkdir -p "$out"
root=$(pwd)
export PYTHONPATH="$root:$PYTHONPATH" NESSAI_AMD_RESIDENT=4x1
for f in variants/libnessai_amd_c2a*.so; do
  b=$(basename $f .so | sed s/libnessai_amd_c2a//)
  (cd /tmp && TMPDIR=/tmp NESSAI_AMD_LIBRARY=$root/$f rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d "$root/$out/a$b" -o pmc -- python $root/scripts/bench_resident.py --geometry 4x1 --net resnet 1000000 > "$root/$out/a$b.log" 2>&1)
done
python - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
table = {}
for d in sorted(glob.glob(out + "/a*/")):
    bits = int(os.path.basename(d.rstrip("/"))[1:])
    acc = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(path, newline="")) if "nsa_flow_kernel" in r["Kernel_Name"] and "Lb1EEv" in r["Kernel_Name"].replace(" ", "") or "true>" in r["Kernel_Name"]]
        rows = [r for r in rows if "nsa_flow_kernel" in r["Kernel_Name"]]
        full = max((int(r["Grid_Size"]) for r in rows), default=0)
        for r in rows:
            if int(r["Grid_Size"]) == full:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    table[bits] = {k: sum(v) / len(v) for k, v in acc.items()}
tiles = 31250.0
names = {0: "complete kernel", 4: "affine transform", 16: "latent draw (Philox, Box-Muller)", 64: "populate epilogue loop", 128: "populate prologue (norm, density)",
         512: "LU-linear ops", 1024: "hidden layers (4 per coupling)", 2048: "last layer + transform"}
with open(out + "/summary.txt", "w") as f:
    base = table.get(0, {})
    for bits in sorted(table):
        t = table[bits]
        line = f"ablate {bits:5d} {names.get(bits, ''):36s} VALU/tile {t.get('SQ_INSTS_VALU', 0) / tiles:8.1f}  SALU/tile {t.get('SQ_INSTS_SALU', 0) / tiles:7.1f}  LDS/tile {t.get('SQ_INSTS_LDS', 0) / tiles:7.1f}"
        if bits and base:
            line += f"   -> section: VALU {(base['SQ_INSTS_VALU'] - t.get('SQ_INSTS_VALU', 0)) / tiles:7.1f}  SALU {(base['SQ_INSTS_SALU'] - t.get('SQ_INSTS_SALU', 0)) / tiles:7.1f}  LDS {(base['SQ_INSTS_LDS'] - t.get('SQ_INSTS_LDS', 0)) / tiles:7.1f}"
        print(line); f.write(line + "\n")
PY
rm -rf "$out"/a*/
