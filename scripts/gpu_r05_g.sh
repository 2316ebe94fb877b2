#!/bin/bash
# round 5: the whole GPU suite on the final library, smoke(), then the round's profile set
OUT=gpurun_out/r05g
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
grep -E "bar .* missed" $OUT/pytest.log | sort | uniq | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -6 $OUT/smoke.log
bash scripts/profile_r05.sh > $OUT/profile.log 2>&1
tail -30 $OUT/profile.log
