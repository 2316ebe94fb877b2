#!/bin/bash
# round 5: scalar stage table with / without the readfirstlane on the loaded words (headline kernel, one box, interleaved),
# then the loader geometry's timeline again
OUT=gpurun_out/r05s
mkdir -p $OUT
for rep in 1 2; do
for v in tabs1 tabs2; do
  NESSAI_AMD_LIBRARY=variants/libnessai_amd_$v.so timeout 200 python scripts/bench_variants_r05.py 2>/dev/null | tail -1 >> $OUT/ab.jsonl
done
done
cat $OUT/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['library'], d['ms_avg'], d['ms_min'], d['kept'], d['log_w_sum'], d['status'])
"
for cfg in "1 1" "2 1" "0 1"; do
  set -- $cfg
  NSA_TL=1 NESSAI_AMD_LIBRARY=variants/libnessai_amd_tl4.so NESSAI_AMD_TEAM=$1 NESSAI_AMD_LOADER=$2 timeout 120 python scripts/bench_small_batch.py 2>&1 | tail -1 >> $OUT/tl.jsonl
done
cat $OUT/tl.jsonl
