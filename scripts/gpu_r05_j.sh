#!/bin/bash
# round 5: the whole GPU suite on the library with the team geometry and the stage ring, then smoke()
OUT=gpurun_out/r05j
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
grep -E "bar .* missed" $OUT/pytest.log | sort | uniq | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -6 $OUT/smoke.log
