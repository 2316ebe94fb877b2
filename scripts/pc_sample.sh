#!/bin/bash
# PC sampling of a command (rocprofv3 beta feature; no counters, no other trace domains): where the waves of
# a kernel are, instruction by instruction.  usage (GPU box, repo root): bash scripts/pc_sample.sh OUTDIR METHOD UNIT INTERVAL -- command ...
out=$1; method=$2; unit=$3; interval=$4; shift 5
mkdir -p "$out"
root=$(pwd)
export PYTHONPATH="$root:$PYTHONPATH"
(cd /tmp && TMPDIR=/tmp ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method \
   --pc-sampling-unit $unit --pc-sampling-interval $interval --kernel-trace --output-format csv -d "$root/$out/raw" -o pcs -- "$@" > "$root/$out/log.txt" 2>&1)
echo "rc=$?" >> "$out/log.txt"
ls -la "$out"/raw/* 2>/dev/null | head
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for path in glob.glob(out + "/raw/**/*pc_sampling*.csv", recursive=True):
    rows = csv.DictReader(open(path, newline=""))
    print(path, rows.fieldnames)
    hist = collections.Counter()
    n = 0
    for r in rows:
        n += 1
        key = (r.get("Instruction_Comment") or "", r.get("Instruction") or "", r.get("Code_Object_Offset") or r.get("Code_Object_Id") or "")
        hist[key] += 1
    with open(out + "/" + path.split("/")[-1].replace(".csv", "_hist.txt"), "w") as f:
        f.write(f"samples {n}\n")
        for k, v in hist.most_common():
            f.write(f"{v}\t{k[2]}\t{k[1]}\t{k[0]}\n")
    print("samples", n)
PY
rm -rf "$out/raw"
