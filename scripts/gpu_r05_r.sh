#!/bin/bash
# round 5: is the loader geometry bound by its loader or by its computing wave?  (timeline build, one process each)
OUT=gpurun_out/r05r
mkdir -p $OUT
for cfg in "1 1" "1 0" "2 1" "4 1" "0 1"; do
  set -- $cfg
  NSA_TL=1 NESSAI_AMD_LIBRARY=variants/libnessai_amd_tl4.so NESSAI_AMD_TEAM=$1 NESSAI_AMD_LOADER=$2 timeout 120 python scripts/bench_small_batch.py 2>&1 | tail -1 >> $OUT/tl.jsonl
done
cat $OUT/tl.jsonl
