#!/bin/bash
# round 5: the whole GPU suite, the A/B of the round's spline / priority changes on one box, and the bench
OUT=gpurun_out/r05e
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
grep -E "bar .* missed" $OUT/pytest.log | head -20
run() { timeout 150 python scripts/bench_variants_r05.py "$@" >> $OUT/variants.jsonl 2>> $OUT/err.log || echo "run failed: $* ($NESSAI_AMD_LIBRARY)" >> $OUT/err.log; }
run
NESSAI_AMD_PRIO=0 run
NESSAI_AMD_LIBRARY=variants/libnessai_amd_shifted.so run
NESSAI_AMD_LIBRARY=variants/libnessai_amd_shifted.so NESSAI_AMD_PRIO=0 run
run
python -c "
import sys, json
for l in open('$OUT/variants.jsonl'):
    d = json.loads(l); print(d['library'], 'prio', d['prio'], 'skew', d['skew'], d['ms_avg'], d['ms_min'], d['kept'], d['log_w_sum'], d['status'])
"
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05e/bench.json"))
r = d["roofline"]
print(round(d["value"]), "acc/s", round(d["ms_per_step"], 2), "ms/step kernel", round(r["kernel_ms_avg"], 4), "frac", round(r["frac"], 4), "traffic", r["traffic"])
PY
