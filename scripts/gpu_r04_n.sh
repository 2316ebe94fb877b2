#!/bin/bash
# config-4 training step: phase stamps of the first-generation kernels, then tests and step times
mkdir -p gpurun_out/r4n
NESSAI_AMD_TRAIN_TIMING=1 python scripts/train_phase_timing.py --c4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4n/phases_c4.txt
for c in c4 c2; do python scripts/bench_train_c2.py --config $c 2>&1 | grep -v amdgpu.ids | cut -c1-220; done
if [ "$1" != "quick" ]; then
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd.py tests/test_gpu_importance.py -x -q 2>&1 | tail -n 3
fi
