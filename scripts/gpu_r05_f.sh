#!/bin/bash
# round 5: conditional flows (inference + training) first, then the whole GPU suite
OUT=gpurun_out/r05f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conditional.py tests/test_gpu_conditional_train.py -m gpu -q > $OUT/pytest_conditional.log 2>&1
tail -25 $OUT/pytest_conditional.log
timeout 1800 python -m pytest tests -m gpu -x -q --ignore tests/test_gpu_conditional.py --ignore tests/test_gpu_conditional_train.py > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
grep -E "bar .* missed" $OUT/pytest.log | head -20
