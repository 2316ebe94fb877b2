#!/bin/bash
# training kernels after the first-generation operand-load work: tests, step times, phase stamps and per-kernel
# statistics of config 4's flow
mkdir -p gpurun_out/r4m
if [ "$1" != "quick" ]; then
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd.py tests/test_gpu_importance.py tests/test_gpu_end_to_end.py -x -q > gpurun_out/r4m/tests.txt 2>&1
tail -n 3 gpurun_out/r4m/tests.txt
fi
rm -f gpurun_out/r4m/train_configs.jsonl
for c in c4 c2 c5; do python scripts/bench_train_c2.py --config $c 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4m/train_configs.jsonl; done
python scripts/bench_train_c2.py --plain 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4m/train_configs.jsonl
python scripts/bench_train.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4m/train_c3.json | cut -c1-200
NESSAI_AMD_TRAIN_TIMING=1 python scripts/train_phase_timing.py --c4 --fine 2>&1 | grep -v amdgpu.ids > gpurun_out/r4m/train_c4_phases.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 -- python scripts/bench_train_c2.py --config c4 > /tmp/prof_c4.log 2>&1
python - <<'PY' | tee gpurun_out/r4m/train_c4_kernels.txt
import sys, glob
sys.path.insert(0, 'scripts')
import summarise_prof as S
db = glob.glob('/tmp/prof_c4/**/*.db', recursive=True)
print(S.kernel_stats(db[0])[:2500])
PY
