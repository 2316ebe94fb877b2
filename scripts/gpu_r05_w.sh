#!/bin/bash
# round 5: tiles per workgroup by default - geometry tests, config 4's secondary numbers, then the whole suite and smoke
OUT=gpurun_out/r05w
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_team.py tests/test_gpu_importance.py -m gpu -q -x > $OUT/team.log 2>&1; tail -5 $OUT/team.log
timeout 300 python scripts/bench_configs.py --c4 > $OUT/c4.jsonl 2> $OUT/c4.err; cut -c1-200 $OUT/c4.jsonl
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
