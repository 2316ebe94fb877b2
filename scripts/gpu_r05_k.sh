#!/bin/bash
# round 5: A/B of the stage-table / LDS-DMA variants of the headline draw kernel on ONE box, interleaved twice
OUT=gpurun_out/r05k
mkdir -p $OUT
for rep in 1 2; do
for v in base tabs raw tabsraw; do
  NESSAI_AMD_LIBRARY=variants/libnessai_amd_$v.so timeout 200 python scripts/bench_variants_r05.py 2>/dev/null | tail -1 >> $OUT/ab.jsonl
done
done
cat $OUT/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['library'], d['ms_avg'], d['ms_min'], d['kept'], d['log_w_sum'], d['status'])
"
