#!/bin/bash
# round 5: the driver's own round-end sequence on the committed state
OUT=gpurun_out/r05final
mkdir -p $OUT
timeout 3000 python -m pytest tests/ -x -q -m gpu > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json
