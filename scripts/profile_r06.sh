#!/bin/bash
# Round-6 profiles (run through gpurun from the repo root): everything lands in gpurun_out/r6prof/ as text / JSON
# lines, ready to be copied into profiles/.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
O=gpurun_out/r6prof
mkdir -p $O
# 1. headline: trace + stats of the bench command, the bench line itself (PMC traffic, train leg, full CPU pass), the RCCL code path
timeout 900 bash scripts/profile_bench.sh r06 > $O/profile_bench.log 2>&1
python scripts/summarise_prof.py gpurun_out/prof_r06 > $O/r06_bench_c3.txt 2>&1
find gpurun_out/prof_r06 -name "*.db" -delete; find gpurun_out/prof_r06 -name "*.csv" -delete
timeout 600 python bench.py --steps 20 --warmup 3 > $O/r06_bench_c3.json 2> $O/bench.err
NESSAI_AMD_FORCE_SHARD=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-train-leg > $O/r06_bench_force_shard.json 2> $O/bench_shard.err
NESSAI_AMD_LEAN=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-leg > $O/r06_bench_c3_full_passes.json 2> $O/bench_full.err
# 2. SQ counters of the headline kernel (the head of a lean pass)
timeout 900 bash scripts/pmc_kernel.sh $O/pmc_c3 "nsa_flow_kernel<2, NsaRQ<8, false>, true, 2, 2, 2" -- python $(pwd)/bench.py --traffic-probe > /dev/null 2>&1
cp $O/pmc_c3/summary.txt $O/r06_draw_kernel_sq_counters_c3.txt
# 3. secondary configs (with the kernel table of the same run), training
timeout 900 rocprofv3 --kernel-trace --stats -d $O/cfg_trace -o trace -- python scripts/bench_configs.py > $O/r06_secondary_configs.jsonl 2> $O/cfg.err
python - <<'PY' > gpurun_out/r6prof/r06_secondary_configs_kernels.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/r6prof/cfg_trace/**/*.db', recursive=True):
    print(S.kernel_stats(db))
PY
rm -rf $O/cfg_trace $O/pmc_c3
for c in c2 c4 c5; do timeout 300 python scripts/bench_train_c2.py --config $c >> $O/r06_train_configs.jsonl 2>> $O/train.err; done
NESSAI_AMD_TIMING=1 timeout 600 python scripts/run_ns_c3.py --iterations 12000 > $O/r06_ns_run_c3.json 2> $O/ns.err
ls -la $O | head -40
head -c 700 $O/r06_bench_c3.json; echo; head -c 400 $O/r06_bench_force_shard.json; echo
