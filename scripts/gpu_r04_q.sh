#!/bin/bash
# training fuzz on the final library (both generations of the fused kernels; TRAIN_V2=0 sends every fused case to the first)
mkdir -p gpurun_out/r4q2
for s in 3 4; do
  python scripts/fuzz_train.py --iterations 75 --seed $s 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-600
  NESSAI_AMD_TRAIN_V2=0 python scripts/fuzz_train.py --iterations 75 --seed $s 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-600
done | tee gpurun_out/r4q2/fuzz_train.txt
