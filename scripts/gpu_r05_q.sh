#!/bin/bash
# round 5: randomised sweeps on the final library (the small-batch geometry is the default for most of their launches)
OUT=gpurun_out/r05q
mkdir -p $OUT
timeout 900 python scripts/fuzz_parity.py --round5 --iterations 160 --seed 0 > $OUT/fuzz_parity_r5_s0.log 2>&1; tail -4 $OUT/fuzz_parity_r5_s0.log
timeout 900 python scripts/fuzz_parity.py --round5 --iterations 160 --seed 5 > $OUT/fuzz_parity_r5_s5.log 2>&1; tail -4 $OUT/fuzz_parity_r5_s5.log
timeout 900 python scripts/fuzz_parity.py --iterations 160 --seed 2 > $OUT/fuzz_parity_s2.log 2>&1; tail -4 $OUT/fuzz_parity_s2.log
NESSAI_AMD_TEAM=4 timeout 900 python scripts/fuzz_parity.py --round5 --iterations 100 --seed 3 > $OUT/fuzz_parity_team4_s3.log 2>&1; tail -3 $OUT/fuzz_parity_team4_s3.log
timeout 900 python scripts/fuzz_populate.py --iterations 40 --seed 1 > $OUT/fuzz_populate.log 2>&1; tail -4 $OUT/fuzz_populate.log
timeout 900 python scripts/fuzz_train.py --iterations 40 --seed 1 > $OUT/fuzz_train.log 2>&1; tail -4 $OUT/fuzz_train.log
