#!/bin/bash
# round 4, session G: single-reciprocal affine + log of products, zero-cell gather tables (image format 8); SLP variant
cd "$(dirname "$0")/.."
o=gpurun_out/r4g
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_flows.py tests/test_gpu_row_tiles.py -x -q 2>&1 | tail -8 > $o/test_flows.txt
timeout 600 python scripts/bench_resident.py 1000000 > $o/resident_geometries.jsonl 2> $o/resident_err.txt
NESSAI_AMD_RESIDENT=4x1 NESSAI_AMD_LIBRARY=variants/libnessai_amd_slp.so python scripts/bench_resident.py --geometry 4x1 1000000 > $o/resident_slp.jsonl 2>> $o/resident_err.txt
python bench.py --no-pmc --no-train-leg > $o/bench.json 2> $o/bench.err
cat $o/test_flows.txt; python - <<'PY'
import json
for f in ("resident_geometries", "resident_slp"):
    for l in open(f"gpurun_out/r4g/{f}.jsonl"):
        d = json.loads(l); print(f, d["config"], d["geometry_env"], round(d["kernel_ms"], 4), d["kept_rows"], d["checksum"])
d = json.loads(open("gpurun_out/r4g/bench.json").read()); print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"])
PY
tail -n 3 $o/bench.err $o/resident_err.txt
