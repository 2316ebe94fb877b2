#!/bin/bash
# training kernels, second generation: parity suite, phase stamps, kernel stats; same without the dW atomics
cd /root/repo
mkdir -p gpurun_out/i
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/i/tests.txt
python scripts/bench_train.py --epochs 40 --cpu-epochs 1 > gpurun_out/i/bench_v2.txt 2>&1
python scripts/train_phase_timing.py > gpurun_out/i/phases.txt 2>&1
if [ -f nessai_amd/csrc/build_abl/libnessai_amd_abl.so ]; then
NESSAI_AMD_LIBRARY=/root/repo/nessai_amd/csrc/build_abl/libnessai_amd_abl.so python scripts/train_phase_timing.py > gpurun_out/i/phases_noatomics.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/i/prof -o train -- python /root/repo/scripts/bench_train.py --epochs 40 --cpu-epochs 1 > /dev/null 2>&1
cd /root/repo
python - <<'PY' > gpurun_out/i/prof_summary.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/i/prof/**/*.db', recursive=True):
    print(S.kernel_stats(db))
    print(S.gap_stats(db))
PY
rm -rf gpurun_out/i/prof
tail -3 gpurun_out/i/tests.txt; tail -2 gpurun_out/i/bench_v2.txt; cat gpurun_out/i/phases.txt; head -4 gpurun_out/i/prof_summary.txt; tail -3 gpurun_out/i/prof_summary.txt
