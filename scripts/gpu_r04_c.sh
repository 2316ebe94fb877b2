#!/bin/bash
# round 4, session C: VALU diet of the small-flow path (fma_mix split, bare-unit affine / Box-Muller, batch norm
# folded into LU) - parity, time per geometry, instruction rates, counters
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4c
o=gpurun_out/r4c
scripts/ubench/valu_rates > $o/valu_rates.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | tail -15 > $o/test_resident.txt
timeout 600 python scripts/bench_resident.py 1000000 > $o/resident_geometries.jsonl 2> $o/resident_err.txt
timeout 1500 python -m pytest tests/test_gpu_flows.py tests/test_gpu_row_tiles.py tests/test_gpu_populate.py tests/test_gpu_golden.py -q 2>&1 | tail -40 > $o/test_flows.txt
NESSAI_AMD_RESIDENT=4x1 bash scripts/pmc_kernel.sh $o/pmc_4x1 nsa_flow_kernel -- python $(pwd)/scripts/bench_resident.py --geometry 4x1 --net resnet 1000000 > $o/pmc_4x1.txt 2>&1
python bench.py --no-pmc --no-train-leg > $o/bench.json 2> $o/bench.err
cat $o/valu_rates.txt $o/test_resident.txt $o/resident_geometries.jsonl $o/test_flows.txt $o/pmc_4x1/summary.txt $o/bench.json
tail -5 $o/resident_err.txt $o/bench.err
