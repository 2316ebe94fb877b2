#!/bin/bash
# round 4, session B: SQ counters of the C2 draw kernel, resident 4x1 / 3x2 and staged
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4b
for g in 4x1 3x2 0; do
  NESSAI_AMD_RESIDENT=$g bash scripts/pmc_kernel.sh gpurun_out/r4b/pmc_$g nsa_flow_kernel -- python $(pwd)/scripts/bench_resident.py --geometry $g --net resnet 1000000 > gpurun_out/r4b/pmc_$g.txt 2>&1
done
tail -n 40 gpurun_out/r4b/pmc_4x1/summary.txt gpurun_out/r4b/pmc_3x2/summary.txt gpurun_out/r4b/pmc_0/summary.txt
tail -3 gpurun_out/r4b/pmc_4x1/g1.log
