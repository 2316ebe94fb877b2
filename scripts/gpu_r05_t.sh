#!/bin/bash
# round 5: how many loader waves (HT = 4 variant library, one process each, same box)
OUT=gpurun_out/r05t
mkdir -p $OUT
for cfg in "1 2" "1 3" "1 4" "2 3" "2 4" "1 1"; do
  set -- $cfg
  NESSAI_AMD_LIBRARY=variants/libnessai_amd_ld.so NESSAI_AMD_TEAM=$1 NESSAI_AMD_LOADER=$2 timeout 120 python scripts/bench_small_batch.py 2>&1 | tail -1 >> $OUT/ld2.jsonl
  INVERSE=1 NESSAI_AMD_LIBRARY=variants/libnessai_amd_ld.so NESSAI_AMD_TEAM=$1 NESSAI_AMD_LOADER=$2 timeout 120 python scripts/bench_small_batch.py 2>&1 | tail -1 >> $OUT/ld2.jsonl
done
cat $OUT/ld2.jsonl
