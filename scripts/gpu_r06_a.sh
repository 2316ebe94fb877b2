#!/bin/bash
# round 6, call a: the new tests of the advisor fixes + RCCL one-rank test; config-2 same-box A/B (r04 tree vs now)
OUT=gpurun_out/r06a
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_lean.py tests/test_gpu_sharded.py tests/test_gpu_conditional.py -x -q -m gpu > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
timeout 600 python -m pytest tests/test_gpu_flows.py -x -q -m gpu -k "tail_bound or k5 or k2 or k13" > $OUT/pytest_flows.log 2>&1; tail -3 $OUT/pytest_flows.log
timeout 900 python scripts/bench_c2_ab.py --root . --root variants/r04 --rounds 3 > $OUT/c2_ab.jsonl 2> $OUT/c2_ab.err; cat $OUT/c2_ab.jsonl | cut -c1-900
