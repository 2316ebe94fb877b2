#!/bin/bash
# round 5: which small-batch geometry per flow shape (one process each, same box): config 4 and config 3 / 5 / 2 shapes
OUT=gpurun_out/r05n
mkdir -p $OUT
run() { "$@" timeout 120 python scripts/bench_small_batch.py $ROWS $SHAPE 2>/dev/null | tail -1 >> $OUT/small.jsonl; }
for cfg in "5050 64 128 6 2" "2000 32 64 6 2" "4000 15 30 4 2" "5050 96 192 2 1"; do
  set -- $cfg; ROWS=$1; SHAPE="$2 $3 $4 $5"
  for t in 0 1 2 4; do run env NESSAI_AMD_TEAM=$t; done
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_LOADER=0
  run env NESSAI_AMD_TEAM=4 NESSAI_AMD_LOADER=0
  run env INVERSE=1 NESSAI_AMD_TEAM=0
  run env INVERSE=1 NESSAI_AMD_TEAM=1
done
cat $OUT/small.jsonl
