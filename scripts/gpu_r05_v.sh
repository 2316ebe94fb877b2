#!/bin/bash
# round 5: the loader geometry with several tiles per workgroup on mid-size batches (one process each, same box)
OUT=gpurun_out/r05v
mkdir -p $OUT
run() { "$@" timeout 120 python scripts/bench_small_batch.py $ROWS $SHAPE 2>/dev/null | tail -1 >> $OUT/mid.jsonl; }
for cfg in "10000 64 128 6 2" "20000 64 128 6 2" "40400 64 128 6 2" "20000 32 64 6 2"; do
  set -- $cfg; ROWS=$1; SHAPE="$2 $3 $4 $5"
  run env NESSAI_AMD_TEAM=0
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=2 NESSAI_AMD_LOADER=3
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=4 NESSAI_AMD_LOADER=3
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=4 NESSAI_AMD_LOADER=2
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=5 NESSAI_AMD_LOADER=3
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=6 NESSAI_AMD_LOADER=2
  run env NESSAI_AMD_TEAM=1 NESSAI_AMD_TILES=1 NESSAI_AMD_LOADER=3
done
cat $OUT/mid.jsonl
