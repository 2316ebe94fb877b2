#!/bin/bash
# round 4, session E: leaner prologue / epilogue / dense write-back, kernel args re-read at use - parity + time
cd "$(dirname "$0")/.."
o=gpurun_out/r4e
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | tail -8 > $o/test_resident.txt
timeout 600 python scripts/bench_resident.py 1000000 > $o/resident_geometries.jsonl 2> $o/resident_err.txt
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_resident.py 2>&1 | tail -25 > $o/test_all.txt
python bench.py --no-pmc --no-train-leg > $o/bench.json 2> $o/bench.err
python scripts/bench_configs.py > $o/secondary_configs.jsonl 2> $o/secondary_err.txt
cat $o/test_resident.txt; python - <<'PY'
import json
for l in open("gpurun_out/r4e/resident_geometries.jsonl"):
    d = json.loads(l); print(d["config"], d["geometry_env"], round(d["kernel_ms"], 4), d["kept_rows"], d["checksum"])
d = json.loads(open("gpurun_out/r4e/bench.json").read()); print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"])
for l in open("gpurun_out/r4e/secondary_configs.jsonl"):
    d = json.loads(l); r = d.get("roofline") or {}
    print(d["config"], d.get("n_flows", ""), "kernel_ms", r.get("kernel_ms_avg"), "frac", r.get("frac"), "ms/populate", d.get("gpu_ms_per_populate"), d.get("draw_ms"), d.get("update_log_q_ms_20000_rows"), d.get("gpu_ms"))
PY
cat $o/test_all.txt; tail -n 3 $o/bench.err $o/secondary_err.txt $o/resident_err.txt
