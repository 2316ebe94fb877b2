#!/bin/bash
# second-generation training kernels: parity suite, step time with and without them, phase stamps, kernel stats
cd /root/repo
mkdir -p gpurun_out/h
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/h/tests.txt
python scripts/bench_train.py --epochs 40 --cpu-epochs 0 > gpurun_out/h/bench_v2.txt 2>&1
NESSAI_AMD_TRAIN_V2=0 python scripts/bench_train.py --epochs 40 --cpu-epochs 0 > gpurun_out/h/bench_v1.txt 2>&1
python scripts/train_phase_timing.py > gpurun_out/h/phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/h/prof -o train -- python /root/repo/scripts/bench_train.py --epochs 40 --cpu-epochs 0 > /dev/null 2>&1
cd /root/repo
python - <<'PY' > gpurun_out/h/prof_summary.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/h/prof/**/*.db', recursive=True):
    print(S.kernel_stats(db))
PY
rm -rf gpurun_out/h/prof
tail -5 gpurun_out/h/tests.txt; tail -3 gpurun_out/h/bench_v2.txt; tail -3 gpurun_out/h/bench_v1.txt; cat gpurun_out/h/phases.txt; head -8 gpurun_out/h/prof_summary.txt
