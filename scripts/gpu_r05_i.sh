#!/bin/bash
# round 5: team geometry + batch norm within the blocks + the failed cases of the last full run, config 4 with and
# without the team geometry, the host profile of config 4's draw, the invertibility-bar case
OUT=gpurun_out/r05i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_team.py tests/test_gpu_batch_norm_within.py tests/test_gpu_resident.py -m gpu -q -x -k "team or within or nsf_k13 or ring" > $OUT/team.log 2>&1
tail -25 $OUT/team.log
timeout 200 python scripts/debug_bars.py nsf 2 2 > $OUT/bars.log 2>&1; tail -12 $OUT/bars.log
timeout 300 python scripts/bench_configs.py --c4 > $OUT/c4_team.jsonl 2> $OUT/c4_team.err; cut -c1-420 $OUT/c4_team.jsonl
NESSAI_AMD_TEAM=0 timeout 300 python scripts/bench_configs.py --c4 > $OUT/c4_noteam.jsonl 2> $OUT/c4_noteam.err; cut -c1-420 $OUT/c4_noteam.jsonl
timeout 200 python scripts/profile_c4_host.py > $OUT/c4_host.log 2>&1; head -60 $OUT/c4_host.log
