#!/bin/bash
# round 5: the loader geometry on BIG batches of the 128-wide flow (several rounds of workgroups) against the staged one
OUT=gpurun_out/r05cc
mkdir -p $OUT
run() { "$@" timeout 150 python scripts/bench_small_batch.py $ROWS $SHAPE 2>/dev/null | tail -1 >> $OUT/big.jsonl; }
for cfg in "200000 64 128 6 2" "1000000 64 128 6 2" "200000 32 64 6 2"; do
  set -- $cfg; ROWS=$1; SHAPE="$2 $3 $4 $5"
  run env NESSAI_AMD_TEAM=0
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=5 NESSAI_AMD_LOADER=3
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=6 NESSAI_AMD_LOADER=2
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=7 NESSAI_AMD_LOADER=1
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=4 NESSAI_AMD_LOADER=3
done
cat $OUT/big.jsonl
