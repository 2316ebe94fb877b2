#!/bin/bash
# tanh / silu of the conditioner on the bare transcendental units: parity tests, kernel time by activation, a fuzz seed
mkdir -p gpurun_out/r4r
python -m pytest tests/test_gpu_flows.py tests/test_gpu_resident.py tests/test_gpu_fuzz.py tests/test_gpu_row_tiles.py tests/test_gpu_populate.py tests/test_gpu_golden.py -x -q 2>&1 | tail -n 3
for p in f16x3 f32; do python scripts/bench_activations.py $p 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4r/activations_after.jsonl
python scripts/fuzz_parity.py --iterations 350 --seed 2 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tail -n 8 | tee gpurun_out/r4r/parity_seed2.txt
