#!/bin/bash
# Round-3 profiles (run through gpurun from the repo root): everything lands in gpurun_out/r3prof/ as text.
#   bench trace + PMC passes, SQ counters of the draw kernel with and without skewed waves, kernel traces of the
#   secondary configs and of the training benchmark, bench.py with the full-batch CPU baseline.
set -u
export TMPDIR=/tmp
O=gpurun_out/r3prof
mkdir -p $O
bash scripts/profile_bench.sh r03 > $O/profile_bench.log 2>&1
python scripts/summarise_prof.py gpurun_out/prof_r03 > $O/r03_bench_c3.txt 2>&1
python bench.py --steps 10 --warmup 2 > $O/r03_bench_c3.json 2> $O/bench.err
find gpurun_out/prof_r03 -name "*.db" -delete; find gpurun_out/prof_r03 -name "*.csv" -delete
for sk in 1 0; do
  NESSAI_AMD_SKEW=$sk bash scripts/pmc_sq.sh gpurun_out/pmc_sq_skew$sk > $O/pmc_sq_skew$sk.log 2>&1
  cp gpurun_out/pmc_sq_skew$sk/summary.txt $O/r03_draw_kernel_sq_counters_skew$sk.txt
  rm -rf gpurun_out/pmc_sq_skew$sk
done
rocprofv3 --kernel-trace --stats -d $O/cfg_trace -o trace -- python scripts/bench_configs.py > $O/r03_secondary_configs.jsonl 2> $O/cfg.err
python - <<'PY' > gpurun_out/r3prof/r03_secondary_configs_kernels.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/r3prof/cfg_trace/**/*.db', recursive=True):
    print(S.kernel_stats(db))
PY
rm -rf $O/cfg_trace
rocprofv3 --kernel-trace --stats -d $O/train_trace -o trace -- python scripts/bench_train.py --epochs 40 --cpu-epochs 1 > $O/r03_train_c3.json 2> $O/train.err
python - <<'PY' > gpurun_out/r3prof/r03_train_c3.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/r3prof/train_trace/**/*.db', recursive=True):
    print(S.kernel_stats(db))
PY
rm -rf $O/train_trace
python bench.py --steps 10 --warmup 2 --no-pmc --cpu-full > $O/r03_bench_c3_cpu_full.json 2>> $O/bench.err
ls -la $O
