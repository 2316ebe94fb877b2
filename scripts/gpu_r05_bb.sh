#!/bin/bash
# round 5: SQ / SQC counters of config 4's 5050-row launch in the small-batch geometry and in the staged one
OUT=gpurun_out/r05bb
mkdir -p $OUT
timeout 900 bash scripts/pmc_kernel.sh $OUT/loader "nsa_flow_kernel<4" -- python $(pwd)/scripts/bench_small_batch.py > $OUT/loader.log 2>&1
cp $OUT/loader/summary.txt $OUT/loader_summary.txt
NESSAI_AMD_TEAM=0 timeout 900 bash scripts/pmc_kernel.sh $OUT/staged "nsa_flow_kernel<4" -- python $(pwd)/scripts/bench_small_batch.py > $OUT/staged.log 2>&1
cp $OUT/staged/summary.txt $OUT/staged_summary.txt
paste $OUT/loader_summary.txt $OUT/staged_summary.txt | cut -c1-200
