#!/bin/bash
# Round-4 profiles (run through gpurun from the repo root): everything lands in gpurun_out/r4prof/ as text / JSON
# lines, ready to be copied into profiles/.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
O=gpurun_out/r4prof
mkdir -p $O
# 1. headline: trace + stats of the bench command, the bench line itself (PMC traffic + train leg), the RCCL code path
bash scripts/profile_bench.sh r04 > $O/profile_bench.log 2>&1
python scripts/summarise_prof.py gpurun_out/prof_r04 > $O/r04_bench_c3.txt 2>&1
find gpurun_out/prof_r04 -name "*.db" -delete; find gpurun_out/prof_r04 -name "*.csv" -delete
python bench.py --steps 10 --warmup 2 > $O/r04_bench_c3.json 2> $O/bench.err
NESSAI_AMD_FORCE_SHARD=1 python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-train-leg > $O/r04_bench_force_shard.json 2> $O/bench_shard.err
# 2. SQ counters: the headline kernel, the small-flow kernel resident / staged
bash scripts/pmc_kernel.sh $O/pmc_c3 "nsa_flow_kernel<2, NsaRQ<8>, true, 2, 2, 2" -- python $(pwd)/bench.py --traffic-probe > /dev/null 2>&1
cp $O/pmc_c3/summary.txt $O/r04_draw_kernel_sq_counters_c3.txt
NESSAI_AMD_RESIDENT=4x1 bash scripts/pmc_kernel.sh $O/pmc_c2 nsa_flow_kernel -- python $(pwd)/scripts/bench_resident.py --geometry 4x1 --net resnet 1000000 > /dev/null 2>&1
cp $O/pmc_c2/summary.txt $O/r04_c2_resident_sq_counters.txt
NESSAI_AMD_RESIDENT=0 bash scripts/pmc_kernel.sh $O/pmc_c2s nsa_flow_kernel -- python $(pwd)/scripts/bench_resident.py --geometry 0 --net resnet 1000000 > /dev/null 2>&1
cp $O/pmc_c2s/summary.txt $O/r04_c2_staged_sq_counters.txt
# 3. geometries, secondary configs (with the kernel table of the same run), small populate, NS run, training
python scripts/bench_resident.py --geometry auto --geometry 0 --geometry 4x1 --geometry 3x1 --geometry 2x2 --geometry 2x1 --geometry auto 1000000 > $O/r04_resident_geometries.jsonl 2> $O/resident.err
rocprofv3 --kernel-trace --stats -d $O/cfg_trace -o trace -- python scripts/bench_configs.py > $O/r04_secondary_configs.jsonl 2> $O/cfg.err
python - <<'PY' > gpurun_out/r4prof/r04_secondary_configs_kernels.txt 2>&1
import sys, glob; sys.path.insert(0, 'scripts')
import summarise_prof as S
for db in glob.glob('gpurun_out/r4prof/cfg_trace/**/*.db', recursive=True):
    print(S.kernel_stats(db))
PY
rm -rf $O/cfg_trace
python scripts/profile_c2_small.py > $O/r04_c2_small_populate.txt 2>&1
NESSAI_AMD_TIMING=1 python scripts/run_ns_c3.py --iterations 12000 > $O/r04_ns_run_c3.json 2> $O/ns.err
python scripts/run_ns_c3.py --iterations 12000 > $O/r04_ns_run_c3_untimed.json 2>> $O/ns.err
python scripts/bench_train.py --epochs 40 --cpu-epochs 1 > $O/r04_train_c3.json 2> $O/train.err
for c in c2 c4 c5; do python scripts/bench_train_c2.py --config $c >> $O/r04_train_configs.jsonl 2>> $O/train.err; done
python scripts/bench_train_c2.py --plain >> $O/r04_train_configs.jsonl 2>> $O/train.err
python scripts/bench_c3_coupling_resident.py > $O/r04_c3_coupling_resident.jsonl 2>> $O/resident.err
scripts/ubench/valu_rates > $O/r04_valu_rates_ubench.txt 2>&1
scripts/ubench/tile_handoff > $O/r04_tile_handoff_ubench.txt 2>&1
ls -la $O | head -50
head -c 600 $O/r04_bench_c3.json; echo; head -c 400 $O/r04_bench_force_shard.json; echo
