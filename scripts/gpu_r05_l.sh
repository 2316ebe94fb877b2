#!/bin/bash
# round 5: config 4's small launch per geometry, one process each (same box), then the timeline build
OUT=gpurun_out/r05l
mkdir -p $OUT
for t in 0 1 4 8; do for r in 2 8; do
  NESSAI_AMD_TEAM=$t NESSAI_AMD_RING=$r timeout 120 python scripts/bench_small_batch.py 2>/dev/null | tail -1 >> $OUT/small.jsonl
done; done
timeout 120 python scripts/bench_small_batch.py 2>/dev/null | tail -1 >> $OUT/small.jsonl


for t in 0 1 4; do for r in 2 8; do
  NSA_TL=1 NESSAI_AMD_LIBRARY=variants/libnessai_amd_tl4.so NESSAI_AMD_TEAM=$t NESSAI_AMD_RING=$r timeout 120 python scripts/bench_small_batch.py 2>&1 | tail -1 >> $OUT/small_tl.jsonl
done; done
cat $OUT/small.jsonl; cat $OUT/small_tl.jsonl
