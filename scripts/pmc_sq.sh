#!/bin/bash
# SQ counters of the draw kernel (one rocprofv3 --pmc pass per group; no other trace domains).
# usage (GPU box): bash scripts/pmc_sq.sh OUTDIR
out=${1:-gpurun_out/pmc_sq}
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$root/$out/avail.txt" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_VALU" \
           "SQ_INSTS_WAVE32_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/$out/g$i" -o pmc -- python "$root/bench.py" --traffic-probe > "$root/$out/g$i.log" 2>&1 || echo "group $i failed: $grp"
done
cd "$root"
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for path in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(path, newline="")))
    full = max((int(r["Grid_Size"]) for r in rows if "nsa_flow_kernel" in r["Kernel_Name"]), default=0)
    for r in rows:
        if "nsa_flow_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == full:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:36s} mean {sum(v)/len(v):16.1f}  launches {len(v)}"
        print(line); f.write(line + "\n")
PY
