#!/bin/bash
# round 4, session A: the weights-resident geometry - parity, then time per geometry
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | tail -15 > gpurun_out/r4a/test_resident.txt
timeout 600 python scripts/bench_resident.py 1000000 > gpurun_out/r4a/resident_geometries.jsonl 2> gpurun_out/r4a/resident_err.txt
timeout 900 python -m pytest tests/test_gpu_flows.py tests/test_gpu_row_tiles.py tests/test_gpu_populate.py -x -q 2>&1 | tail -15 > gpurun_out/r4a/test_flows.txt
cat gpurun_out/r4a/test_resident.txt gpurun_out/r4a/resident_geometries.jsonl gpurun_out/r4a/test_flows.txt
tail -5 gpurun_out/r4a/resident_err.txt
