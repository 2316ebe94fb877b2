#!/bin/bash
# Profile the headline benchmark on the GPU box (run through gpurun from the repo root).
#   1. rocprofv3 --kernel-trace --stats  -> per-kernel time
#   2. separate --pmc passes             -> MFMA busy, VALU, LDS conflicts, HBM bytes
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $group | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $group -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
find $OUT -name "*.csv" | head -50
