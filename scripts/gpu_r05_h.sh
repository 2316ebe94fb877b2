#!/bin/bash
# round 5: batch norm within the residual blocks first, then the whole GPU suite (every failure, not the first)
OUT=gpurun_out/r05h
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_batch_norm_within.py tests/test_gpu_reference_bars.py -m gpu -q > $OUT/bnw.log 2>&1
tail -25 $OUT/bnw.log
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
grep -E "bar .* missed" $OUT/pytest.log | sort | uniq | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -6 $OUT/smoke.log
