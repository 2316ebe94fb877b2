#!/bin/bash
# round 5: 256-wide instances at one wave per SIMD (512 registers) against the two-waves build (variant), same box;
# then the tests that run 256-wide flows
OUT=gpurun_out/r05y
mkdir -p $OUT
run() { "$@" timeout 150 python scripts/bench_small_batch.py $ROWS $SHAPE 2>/dev/null | tail -1 >> $OUT/wide.jsonl; }
for cfg in "200000 96 192 2 1" "5050 96 192 2 1" "200000 100 200 2 1"; do
  set -- $cfg; ROWS=$1; SHAPE="$2 $3 $4 $5"
  for lib in "" variants/libnessai_amd_w8old.so; do
    for prec in f16x3 f32; do
      run env NESSAI_AMD_LIBRARY=$lib PRECISION=$prec
      run env NESSAI_AMD_LIBRARY=$lib PRECISION=$prec INVERSE=1
    done
  done
done
cat $OUT/wide.jsonl
timeout 1500 python -m pytest tests/test_gpu_flows.py tests/test_gpu_team.py tests/test_gpu_resident.py tests/test_gpu_conditional.py -m gpu -q -k "h192 or d100 or h128 or team" > $OUT/wide_tests.log 2>&1; tail -5 $OUT/wide_tests.log
