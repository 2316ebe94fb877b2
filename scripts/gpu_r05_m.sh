#!/bin/bash
# round 5: the loader wave - team / ring tests, then config 4's small launch per geometry (one process each, same box)
OUT=gpurun_out/r05m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_team.py tests/test_gpu_reference_bars.py tests/test_gpu_batch_norm_within.py tests/test_gpu_conditional.py -m gpu -q -x > $OUT/team.log 2>&1
tail -12 $OUT/team.log
for l in 1 0; do for t in 1 2 4 8; do
  NESSAI_AMD_LOADER=$l NESSAI_AMD_TEAM=$t timeout 120 python scripts/bench_small_batch.py 2>/dev/null | tail -1 >> $OUT/small.jsonl
done; done
NESSAI_AMD_TEAM=0 timeout 120 python scripts/bench_small_batch.py 2>/dev/null | tail -1 >> $OUT/small.jsonl
for r in 2 3 4; do NESSAI_AMD_RING=$r timeout 120 python scripts/bench_small_batch.py 2>/dev/null | tail -1 >> $OUT/small.jsonl; done
INVERSE=1 timeout 120 python scripts/bench_small_batch.py 2>/dev/null | tail -1 >> $OUT/small.jsonl
cat $OUT/small.jsonl
timeout 300 python scripts/bench_configs.py --c4 > $OUT/c4.jsonl 2> $OUT/c4.err; cut -c1-300 $OUT/c4.jsonl
