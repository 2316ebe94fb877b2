#!/bin/bash
# SQ counters of the fused training kernels (one rocprofv3 --pmc pass per group).
# usage (GPU box): bash scripts/pmc_train.sh OUTDIR
out=${1:-gpurun_out/pmc_train}
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/$out/g$i" -o pmc -- python "$root/scripts/bench_train.py" --epochs 10 --cpu-epochs 1 > "$root/$out/g$i.log" 2>&1 || echo "group $i failed: $grp"
done
cd "$root"
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for path in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"]
        if "trf_forward" in k or "trf_backward" in k or "trf2_forward" in k or "trf2_backward" in k:
            acc[("fwd" if "forward" in k else "bwd", r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    for k in sorted(acc):
        v = acc[k]
        line = f"{k[0]} {k[1]:28s} mean {sum(v)/len(v):16.1f}  launches {len(v)}"
        print(line); f.write(line + "\n")
PY
