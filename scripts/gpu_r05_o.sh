#!/bin/bash
# round 5: the whole GPU suite on the final library (loader geometry, scalar stage table), smoke(), the round's profile set
OUT=gpurun_out/r05z
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
grep -E "bar .* missed" $OUT/pytest.log | sort | uniq | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -6 $OUT/smoke.log
bash scripts/profile_r05.sh > $OUT/profile.log 2>&1
tail -30 $OUT/profile.log
