#!/bin/bash
# round 5, final library: geometry / importance tests first, small-batch sweep per flow shape, then the whole suite,
# smoke() and the round's profile set
OUT=gpurun_out/r05u
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_team.py tests/test_gpu_importance.py -m gpu -q -x > $OUT/team.log 2>&1; tail -5 $OUT/team.log
run() { "$@" timeout 120 python scripts/bench_small_batch.py $ROWS $SHAPE 2>/dev/null | tail -1 >> $OUT/small.jsonl; }
for cfg in "5050 64 128 6 2" "2000 32 64 6 2" "4000 15 30 4 2" "5050 96 192 2 1" "2000 16 32 4 2"; do
  set -- $cfg; ROWS=$1; SHAPE="$2 $3 $4 $5"
  run env NESSAI_AMD_TEAM=0
  run env
  run env NESSAI_AMD_LOADER=1
  run env INVERSE=1 NESSAI_AMD_TEAM=0
  run env INVERSE=1
done
cat $OUT/small.jsonl
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -6 $OUT/smoke.log
bash scripts/profile_r05.sh > $OUT/profile.log 2>&1
tail -8 $OUT/profile.log | cut -c1-300
