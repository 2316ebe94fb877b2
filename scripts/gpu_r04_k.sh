#!/bin/bash
# fuzz sweeps of the final round-4 library (flow kernels; populate / training / reparameterisations)
mkdir -p gpurun_out/r4k
for s in 1 2 11; do
  python scripts/fuzz_parity.py --iterations 350 --seed $s 2>&1 | grep -v amdgpu.ids > gpurun_out/r4k/parity_seed$s.txt
done
tail -n 8 gpurun_out/r4k/parity_seed*.txt | cut -c1-600
