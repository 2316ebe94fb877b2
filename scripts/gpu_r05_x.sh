#!/bin/bash
# round 5: teams working in rounds - bit-identity tests, then config 4's small launch per team size / loaders
OUT=gpurun_out/r05x
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_team.py -m gpu -q -x > $OUT/team.log 2>&1; tail -6 $OUT/team.log
run() { "$@" timeout 120 python scripts/bench_small_batch.py $ROWS $SHAPE 2>/dev/null | tail -1 >> $OUT/rounds.jsonl; }
for cfg in "5050 64 128 6 2" "2000 32 64 6 2" "5050 96 192 2 1"; do
  set -- $cfg; ROWS=$1; SHAPE="$2 $3 $4 $5"
  run env
  run env NESSAI_AMD_TEAM=2 NESSAI_AMD_LOADER=3
  run env NESSAI_AMD_TEAM=2 NESSAI_AMD_LOADER=2
  run env NESSAI_AMD_TEAM=2 NESSAI_AMD_LOADER=3 NESSAI_AMD_ROUNDS=0
  run env NESSAI_AMD_TEAM=4 NESSAI_AMD_LOADER=3
  run env NESSAI_AMD_TEAM=4 NESSAI_AMD_LOADER=4
  run env INVERSE=1 NESSAI_AMD_TEAM=2 NESSAI_AMD_LOADER=3
done
cat $OUT/rounds.jsonl
