#!/bin/bash
# round 6, call b: chunk-pipeline ubench (go / no-go gate of the hand-interleaved headline kernel); config-2 A/B with
# TRAINED flows (the r04 / r05 profile files were measured on trained flows); the tests that call a failed in
OUT=gpurun_out/r06b
mkdir -p $OUT
timeout 600 scripts/ubench/chunk_pipeline 4096 > $OUT/chunk_pipeline.txt 2>&1; cat $OUT/chunk_pipeline.txt
timeout 600 python -m pytest tests/test_gpu_conditional.py tests/test_gpu_flows.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
AB_TRAINED=1 timeout 900 python scripts/bench_c2_ab.py --root . --root variants/r04 --rounds 2 > $OUT/c2_ab_trained.jsonl 2> $OUT/c2_ab.err; cut -c1-700 $OUT/c2_ab_trained.jsonl
