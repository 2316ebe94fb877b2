#!/bin/bash
# the cases of the round-4 fuzz sweeps over the max-norm inverse bar, re-run verbosely (rms and conditioning rules)
mkdir -p gpurun_out/r4l
{
python scripts/fuzz_parity.py --iterations 350 --seed 1 --only 5 --only 54
python scripts/fuzz_parity.py --iterations 350 --seed 2 --only 55 --only 56 --only 262 --only 314 --only 333
python scripts/fuzz_parity.py --iterations 350 --seed 11 --only 6
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4l/over_bar_cases.txt
cut -c1-700 gpurun_out/r4l/over_bar_cases.txt
