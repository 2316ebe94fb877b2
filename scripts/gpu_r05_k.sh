#!/bin/bash
# round 5: A/B of headline-kernel library builds on ONE box, interleaved twice (set VARIANTS; "main" = the shipped library)
OUT=gpurun_out/r05k2
mkdir -p $OUT
rm -f $OUT/ab.jsonl
for rep in 1 2; do
for v in ${VARIANTS:-base w512}; do
  if [ $v = main ]; then L=nessai_amd/libnessai_amd.so; else L=variants/libnessai_amd_$v.so; fi
  NESSAI_AMD_LIBRARY=$L timeout 200 python scripts/bench_variants_r05.py 2>/dev/null | tail -1 >> $OUT/ab.jsonl
done
done
cat $OUT/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['library'], d['ms_avg'], d['ms_min'], d['kept'], d['log_w_sum'], d['status'])
"
