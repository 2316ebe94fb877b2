#!/bin/bash
# round 3, GPU session B: whole GPU suite, skew A/B, bench, fuzz, sharded host loop
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -40 $O/pytest_gpu.txt
for sk in 0 1; do
  NESSAI_AMD_SKEW=$sk timeout 300 python scripts/bench_kernel_variants.py --quick >> $O/variants.jsonl 2>&1
done
cut -c1-230 $O/variants.jsonl | grep -v amdgpu.ids
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-1500
timeout 300 python scripts/bench_sharded_host.py > $O/sharded_host.txt 2>&1; tail -2 $O/sharded_host.txt
timeout 600 python scripts/fuzz_precision.py --iterations 60 --modes f16x3,f32 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
timeout 600 python scripts/gpu_precision.py > $O/precision.txt 2>&1; tail -20 $O/precision.txt | cut -c1-300
