#!/bin/bash
# round 3, GPU session A: split-f16 facts + parity + timing, skewed-wave variant
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
./scripts/ubench/f16_facts > $O/f16_facts.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_flows.py tests/test_gpu_row_tiles.py -q --maxfail=20 -x -k "f16 or precision or auto" > $O/pytest_f16.txt 2>&1
tail -5 $O/pytest_f16.txt
timeout 600 python scripts/gpu_precision.py > $O/precision.txt 2>&1
timeout 300 python scripts/bench_kernel_variants.py --quick > $O/variants_main.jsonl 2>&1
for sk in 0 1 2; do
  NESSAI_AMD_SKEW=$sk NESSAI_AMD_LIBRARY=variants/libnessai_amd_skew.so timeout 300 python scripts/bench_kernel_variants.py --quick >> $O/variants_skew.jsonl 2>&1
done
timeout 600 python scripts/fuzz_precision.py --iterations 60 --modes bf16x3,f16x3,f32 > $O/fuzz.txt 2>&1
timeout 600 python scripts/fuzz_precision.py --iterations 60 --wide --seed 5 --modes f16x3,f32 > $O/fuzz_wide.txt 2>&1
cat $O/f16_facts.txt; cat $O/variants_main.jsonl | cut -c1-260; cat $O/variants_skew.jsonl | cut -c1-260; tail -3 $O/fuzz.txt; tail -3 $O/fuzz_wide.txt
