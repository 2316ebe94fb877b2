#!/bin/bash
# round 5: the draw's records written on the device (nsa_importance_mix_points): tests, then config 4 with and without
OUT=gpurun_out/r05p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_importance.py -m gpu -q -x > $OUT/imp.log 2>&1
tail -12 $OUT/imp.log
timeout 300 python scripts/bench_configs.py --c4 > $OUT/c4_points.jsonl 2> $OUT/c4.err; cut -c1-200 $OUT/c4_points.jsonl
NESSAI_AMD_DRAW_POINTS=0 timeout 300 python scripts/bench_configs.py --c4 > $OUT/c4_host.jsonl 2>> $OUT/c4.err; cut -c1-200 $OUT/c4_host.jsonl
timeout 200 python scripts/profile_c4_host.py > $OUT/c4_host.log 2>&1; head -40 $OUT/c4_host.log | cut -c1-150
