#!/bin/bash
# SQ / SQC counters of one kernel of an arbitrary command (one rocprofv3 --pmc pass per group of four; no
# other trace domains).  The kernel is picked by a substring of its name; of its launches only those with
# the largest grid are averaged (the measured launches, not probes on small batches).
# usage (GPU box, from the repo root): bash scripts/pmc_kernel.sh OUTDIR KERNEL_SUBSTRING -- command ...
out=$1; pat=$2; shift 3
mkdir -p "$out"
root=$(pwd)
export PYTHONPATH="$root:$PYTHONPATH"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INSTS_VALU_ADD_F64" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL"; do
  i=$((i+1))
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$root/$out/g$i" -o pmc -- "$@" > "$root/$out/g$i.log" 2>&1) || echo "group $i failed: $grp"
done
python - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
name = None
for path in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(path, newline="")) if pat in r["Kernel_Name"]]
    full = max((int(r["Grid_Size"]) for r in rows), default=0)
    for r in rows:
        if int(r["Grid_Size"]) == full:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
with open(out + "/summary.txt", "w") as f:
    f.write(f"kernel {name}\n")
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:36s} mean {sum(v)/len(v):16.1f}  launches {len(v)}"
        print(line); f.write(line + "\n")
PY
rm -rf "$out"/g[0-9]*  # the raw per-dispatch tables (tens of MB) stay on the box
