#!/bin/bash
# more seeds of the randomised sweeps on the final library
mkdir -p gpurun_out/r4q3
for s in 5 6; do
  python scripts/fuzz_parity.py --iterations 350 --seed $s 2>&1 | grep -v amdgpu.ids | cut -c1-700 > gpurun_out/r4q3/parity_seed$s.txt
  tail -n 1 gpurun_out/r4q3/parity_seed$s.txt
done
python scripts/fuzz_populate.py --iterations 80 --seed 5 2>&1 | grep -v amdgpu.ids | tail -n 3 | cut -c1-500 | tee gpurun_out/r4q3/populate_seed5.txt
python scripts/fuzz_reparam.py --iterations 50 --seed 5 2>&1 | grep -v amdgpu.ids | tail -n 3 | cut -c1-500 | tee gpurun_out/r4q3/reparam_seed5.txt
