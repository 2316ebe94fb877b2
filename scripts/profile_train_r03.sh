#!/bin/bash
# Round-3 training profiles (second-generation kernels), run through gpurun from the repo root; text lands in
# gpurun_out/train_r03/ and is copied to profiles/ by hand:
#   r03_layer_latency_ubench.txt   scripts/ubench/layer_latency (the floor of one hidden layer)
#   r03_train_c3_v2.json / .txt    scripts/bench_train.py, plain and under rocprofv3 --kernel-trace --stats (+ gaps)
#   r03_train_phases_v2.txt        in-kernel phase stamps
#   r03_train_run_lengths.txt      step time with 1 / 2 / 6 couplings per forward launch, and with the first generation
set -u
cd /root/repo
O=gpurun_out/train_r03
mkdir -p $O
[ -x scripts/ubench/layer_latency ] && timeout 120 scripts/ubench/layer_latency > $O/r03_layer_latency_ubench.txt 2>&1
python scripts/bench_train.py --epochs 40 --cpu-epochs 3 2>/dev/null | tail -1 > $O/r03_train_c3_v2.json
: > $O/r03_train_run_lengths.txt
for run in 1 2 6; do
  echo "NESSAI_AMD_TRAIN_RUN=$run" >> $O/r03_train_run_lengths.txt
  for rep in 1 2; do
    NESSAI_AMD_TRAIN_RUN=$run python scripts/bench_train.py --epochs 40 --cpu-epochs 1 2>/dev/null | tail -1 | cut -c1-150 >> $O/r03_train_run_lengths.txt
  done
done
echo "NESSAI_AMD_TRAIN_V2=0 (first-generation kernels)" >> $O/r03_train_run_lengths.txt
NESSAI_AMD_TRAIN_V2=0 python scripts/bench_train.py --epochs 40 --cpu-epochs 1 2>/dev/null | tail -1 | cut -c1-150 >> $O/r03_train_run_lengths.txt
python scripts/train_phase_timing.py 2>/dev/null > $O/r03_train_phases_v2.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o train -- python scripts/bench_train.py --epochs 40 --cpu-epochs 1 > /dev/null 2>&1
python - <<'PY' > gpurun_out/train_r03/r03_train_c3_v2.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/train_r03/trace/**/*.db', recursive=True):
    print(S.kernel_stats(db))
    print(S.gap_stats(db))
PY
rm -rf $O/trace
ls -la $O
