#!/bin/bash
# round 5: experiment variants of the lean draw kernel on one box (every run under its own timeout)
OUT=gpurun_out/r05c2
mkdir -p $OUT
run() { timeout 150 python scripts/bench_variants_r05.py "$@" >> $OUT/variants.jsonl 2>> $OUT/err.log || echo "run failed: $* ($NESSAI_AMD_LIBRARY)" >> $OUT/err.log; }
export NESSAI_AMD_LIBRARY=variants/libnessai_amd_base.so
run
for v in a1 a2 a3; do NESSAI_AMD_LIBRARY=variants/libnessai_amd_$v.so run --no-populate; done
run --no-populate
NESSAI_AMD_PRIO=1 run
NESSAI_AMD_PRIO=2 run
NESSAI_AMD_SKEW=0 run
NESSAI_AMD_SKEW=0 NESSAI_AMD_PRIO=1 run
run
python -c "
import sys, json
for l in open('$OUT/variants.jsonl'):
    d = json.loads(l); print(d['library'], 'prio', d['prio'], 'skew', d['skew'], d['ms_avg'], d['ms_min'], d['kept'], d['log_w_sum'], d['status'])
"
tail -5 $OUT/err.log
