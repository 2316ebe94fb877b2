#!/bin/bash
# couplings per forward launch (NESSAI_AMD_TRAIN_RUN = 1, 2, 3): parity suite once, step time and kernel stats for each
cd /root/repo
mkdir -p gpurun_out/j
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/j/tests.txt
for run in 1 3 6; do
  for rep in 1 2; do
    NESSAI_AMD_TRAIN_RUN=$run python scripts/bench_train.py --epochs 40 --cpu-epochs 1 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/j/bench_run$run.txt
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/j/prof -o train -- python /root/repo/scripts/bench_train.py --epochs 40 --cpu-epochs 1 > /dev/null 2>&1
cd /root/repo
python - <<'PY' > gpurun_out/j/prof_summary.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/j/prof/**/*.db', recursive=True):
    print(S.kernel_stats(db))
    print(S.gap_stats(db))
PY
rm -rf gpurun_out/j/prof
python scripts/train_phase_timing.py > gpurun_out/j/phases.txt 2>&1
tail -2 gpurun_out/j/tests.txt; for run in 1 3 6; do echo run $run; cat gpurun_out/j/bench_run$run.txt; done; head -4 gpurun_out/j/prof_summary.txt | cut -c1-120; tail -3 gpurun_out/j/prof_summary.txt; grep -v bwd gpurun_out/j/phases.txt
