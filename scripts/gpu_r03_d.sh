#!/bin/bash
# round 3, GPU session D: device repack, fused small trains, validation beside training, resume fix
mkdir -p gpurun_out/r3d
O=gpurun_out/r3d
timeout 2400 python -m pytest tests/test_gpu_flows.py tests/test_gpu_populate.py tests/test_gpu_truncation_resume.py tests/test_gpu_train.py tests/test_gpu_sharded.py tests/test_gpu_end_to_end.py tests/test_gpu_importance.py tests/test_gpu_configs.py tests/test_gpu_golden.py -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -30 $O/pytest_gpu.txt
timeout 600 python scripts/profile_c2_small.py > $O/c2_small.txt 2>&1; head -3 $O/c2_small.txt | cut -c1-300
timeout 600 python scripts/bench_retrain_populate.py > $O/retrain.txt 2>&1; tail -1 $O/retrain.txt
timeout 300 python scripts/bench_train.py > $O/train.json 2>&1; tail -1 $O/train.json | cut -c1-400
NESSAI_AMD_VAL_OVERLAP=0 timeout 300 python scripts/bench_train.py > $O/train_inline.json 2>&1; tail -1 $O/train_inline.json | cut -c1-400
timeout 900 python bench.py --no-cpu-baseline --no-pmc > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
