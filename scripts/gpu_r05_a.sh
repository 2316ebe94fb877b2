#!/bin/bash
# round 5, first GPU pass: lean passes + replay (parity, then the bench with and without them on the same box)
set -x
OUT=gpurun_out/r05b
mkdir -p $OUT
python -m pytest tests/test_gpu_lean.py tests/test_gpu_populate.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
python bench.py --steps 10 --warmup 2 --no-train-leg --cpu-bounded > $OUT/bench_lean.json 2> $OUT/bench_lean.err
NESSAI_AMD_LEAN=0 python bench.py --steps 10 --warmup 2 --no-train-leg --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err
NESSAI_AMD_FORCE_SHARD=1 python bench.py --steps 10 --warmup 2 --no-train-leg --no-cpu-baseline --no-pmc > $OUT/bench_shard.json 2> $OUT/bench_shard.err
python - <<'PY'
import json
for n in ("lean", "full", "shard"):
    try:
        d = json.load(open(f"gpurun_out/r05b/bench_{n}.json"))
        r = d["roofline"]
        print(n, round(d["value"]), "acc/s", round(d["ms_per_step"], 2), "ms/step kernel", round(r["kernel_ms_avg"], 4), "frac", round(r["frac"], 4), "traffic", r["traffic"], r.get("traffic_detail"))
    except Exception as e:
        print(n, "failed", e)
PY
