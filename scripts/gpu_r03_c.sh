#!/bin/bash
# round 3, GPU session C: failed tests again + remaining suite, C2 small-batch profile, secondary configs
mkdir -p gpurun_out/r3c
O=gpurun_out/r3c
timeout 2400 python -m pytest tests/test_gpu_row_tiles.py tests/test_gpu_flows.py tests/test_gpu_importance.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py tests/test_gpu_populate.py tests/test_gpu_truncation_resume.py tests/test_gpu_train.py tests/test_gpu_autograd.py tests/test_gpu_end_to_end.py -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -30 $O/pytest_gpu.txt
timeout 600 python scripts/profile_c2_small.py > $O/c2_small.txt 2>&1; head -45 $O/c2_small.txt | cut -c1-200
timeout 1500 python scripts/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cut -c1-900 $O/configs.jsonl
timeout 300 python scripts/bench_train.py > $O/train.json 2>&1; tail -1 $O/train.json | cut -c1-600
