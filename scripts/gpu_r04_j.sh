#!/bin/bash
# round 4, session J: config 2 at the reference's default drawsize = nlive - where a populate() goes
cd "$(dirname "$0")/.."
root=$(pwd)
o=gpurun_out/r4j
mkdir -p $o
TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/prof -o c2 -- python $root/scripts/profile_c2_small.py > $root/$o/prof_log.txt 2>&1
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); cp "$f" $o/c2_small_kernel_stats.csv
t=$(find $o/prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last populate: the last 12 kernels
print(list(rows[0].keys())); last = rows[-16:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>8s} wg {r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")):>5s}  {r["Kernel_Name"][:90]}')
PY
rm -rf $o/prof
grep -i 'small\|flow_kernel\|copyBuffer' $o/c2_small_kernel_stats.csv | cut -c1-160
